#!/bin/bash
# bench.py under several environment settings on ONE box, interleaved twice.  usage: tools/ab_bench.sh "ENV1" "ENV2" ...  ("-" = none)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for e in "$@"; do
    if [ "$e" = "-" ]; then timeout 200 python bench.py --no-cpu-baseline --no-series > /tmp/ab.json 2>/dev/null; else env $e timeout 200 python bench.py --no-cpu-baseline --no-series > /tmp/ab.json 2>/dev/null; fi
    python - "$e" <<'P'
import json, sys
d = json.load(open("/tmp/ab.json"))
k = d["kernels"]
print("%-28s %8.0f pairs/s  %.3f ms  big %.3f  tn %.3f  small %.3f" % (sys.argv[1], d["value"], d["ms_per_step"], k["gemm_big"]["ms_per_step"], k["gemm_big_tn"]["ms_per_step"], k["gemm_small"]["ms_per_step"]))
P
  done
done
