#!/bin/bash
# default-bench and configs[4]-bf16 throughput of several library builds on ONE box.  usage: tools/ab_lib.sh lib1.so lib2.so ...   ("-" = the in-tree build)
cd $GRAFT_REPO_ROOT
C4="--hidden 768 --word-heads 8 --window 5 --gsl-rate 0.8 --no-cpu-baseline --no-series --steps 10 --warmup 3 --batch 32 --gemm-mode bf16 --no-profile"
for lib in "$@"; do
  if [ "$lib" = "-" ]; then unset GET_AMD_LIB; else export GET_AMD_LIB=$GRAFT_REPO_ROOT/$lib; fi
  for rep in 1 2; do
    python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-series --no-profile 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib', 'cfg1 pairs/s %.0f  ms/step %.4f' % (d['value'], d['ms_per_step']))"
  done
  python bench.py $C4 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib', 'cfg4-bf16 pairs/s %.0f  ms/step %.4f' % (d['value'], d['ms_per_step']))"
done
