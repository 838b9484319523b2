#!/bin/bash
# A/B of aggregation-kernel switches on ONE box (tool build).  usage: tools/ab_spmm.sh "GH_SPMM_XCD=0" "GH_SPMM_XCD=1" ...
cd $GRAFT_REPO_ROOT
export GET_AMD_LIB=$GRAFT_REPO_ROOT/get_amd/lib/libget_hip_measure.so
for cfg in "$@"; do
  echo "== $cfg"
  env $cfg python tools/spmm_bench.py 2>&1 | grep -v amdgpu.ids
  env $cfg python bench.py --measure-build --steps 20 --warmup 6 --no-cpu-baseline --no-series --no-side-modes 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('   pairs/s %.0f  ms/step %.4f' % (d['value'], d['ms_per_step']))
for k in ('spmm', 'att_softmax_fwd', 'att_softmax_bwd', 'att_dpre', 'few_row_streams'):
    v = d['kernels'].get(k)
    if v: print('   %-16s %7.4f ms/step %5.1f launches %8.1f GB/s frac %.3f' % (k, v['ms_per_step'], v['launches_per_step'], v['achieved_gbps'], v['frac']))"
done
