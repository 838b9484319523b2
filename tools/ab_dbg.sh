#!/bin/bash
# headline bench under several settings of the tool build's switches, alternating.  usage: tools/ab_dbg.sh "GH_DBG=0" "GH_DBG=16" ...
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for cfg in "$@"; do
  env $cfg python bench.py --measure-build --steps 20 --warmup 6 --no-cpu-baseline --no-series --no-side-modes 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['kernels']
print('%-28s pairs/s %.0f  ms/step %.4f  gemm_big %.4f ms (%.3f)  tn %.4f  small %.4f' % ('$cfg', d['value'], d['ms_per_step'], k['gemm_big']['ms_per_step'], k['gemm_big']['frac'], k['gemm_big_tn']['ms_per_step'], k['gemm_small']['ms_per_step']))"
done
done
