#!/bin/bash
# A/B of the fp32 128 x 160 ping-pong tile (tool build, GH_PP32_ROWS = row threshold) on the headline workload + the big-tile parity tests on it
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6
if [ -z "$NO_TESTS" ]; then
echo "== tests with the fp32 ping-pong tile"
GET_AMD_LIB=$GRAFT_REPO_ROOT/get_amd/lib/libget_hip_measure.so GH_PP32_ROWS=8192 timeout 1500 python -m pytest -x -q -m gpu tests/test_gpu_fullsize_grads.py tests/test_gpu_ops.py tests/test_gpu_model.py -k "not bf16" 2>&1 | tail -4
fi
for rep in 1 2; do
 for e in "$@"; do
  env $e python bench.py --measure-build --steps 30 --warmup 8 --no-cpu-baseline --no-series --no-side-modes --no-strong --no-other-configs $AB_ARGS 2> gpurun_out/r6/ab_err.log | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d.get('kernels', {})
print('%-24s pairs/s %.0f  ms/step %.4f  roof %.3f  parity %s ' % ('$e', d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('parity', {}).get('max_abs_logit_diff_vs_cpu_oracle')) + '  '.join('%s %.3f' % (n, k[n]['ms_per_step']) for n in ('gemm_big', 'gemm_big_tn', 'gemm_small', 'spmm') if n in k))
" || tail -5 gpurun_out/r6/ab_err.log
 done
done
