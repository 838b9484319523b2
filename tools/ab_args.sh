#!/bin/bash
# pairs/s of bench.py under different ARGUMENT sets (product build), alternating on ONE box.  usage: [REPS=n] tools/ab_args.sh "args1" "args2" ...   ("-" = none)
REPS=${REPS:-2}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4
for rep in $(seq 1 $REPS); do
  for a in "$@"; do
    if [ "$a" = "-" ]; then aa=""; else aa="$a"; fi
    python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-series --no-side-modes --no-strong --no-profile $aa 2> gpurun_out/r4/ab_err.log | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-40s pairs/s %.0f  ms/step %.4f' % ('$a', d['value'], d['ms_per_step']))
" || tail -5 gpurun_out/r4/ab_err.log
  done
done
