#!/bin/bash
# A/B of the straight-line fp32 epilogue (variant build, GH_DBG=128 = the plain passes) on the headline workload
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6
V=${VARIANT:-e32}
for rep in 1 2; do
 for e in "$@"; do
  env $e python bench.py --measure-lib $V --steps 30 --warmup 8 --no-cpu-baseline --no-series --no-side-modes --no-strong --no-other-configs $AB_ARGS 2> gpurun_out/r6/ab_err.log | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d.get('kernels', {})
print('%-24s pairs/s %.0f  ms/step %.4f  roof %.3f  parity %s ' % ('$e', d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('parity', {}).get('max_abs_logit_diff_vs_cpu_oracle')) + '  '.join('%s %.3f' % (n, k[n]['ms_per_step']) for n in ('gemm_big', 'gemm_big_tn', 'gemm_small', 'spmm') if n in k))
" || tail -5 gpurun_out/r6/ab_err.log
 done
done
