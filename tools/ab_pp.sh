#!/bin/bash
# A/B of the 256 x 256 ping-pong bf16 tile (variant build `pp`, GH_BF16_TILE=256) against the 128 x 256 tile on configs[4] bf16,
# plus the bf16 parity tests on the new tile.  usage: tools/ab_pp.sh "ENV1" "ENV2" ...
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6
if [ -z "$NO_TESTS" ]; then
for te in ${TEST_ENVS:-"GH_BF16_TILE=256"}; do
echo "== tests on the 256 tile ($te)"
env GET_AMD_LIB=$GRAFT_REPO_ROOT/get_amd/lib/libget_hip_pp.so $te GH_BF16_TILE256_ROWS=0 timeout 900 python -m pytest -x -q tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_wide_composite.py -k "bf16" 2>&1 | tail -3
done
fi
C4="--hidden 768 --word-heads 8 --window 5 --gsl-rate 0.8 --gemm-mode bf16"
for rep in 1 2; do
 for e in "$@"; do
  env $e python bench.py --measure-lib pp --steps 20 --warmup 5 --no-cpu-baseline --no-series --no-side-modes --no-strong --no-other-configs $C4 2> gpurun_out/r6/ab_err.log | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d.get('kernels', {})
print('%-28s pairs/s %.0f  ms/step %.4f  roof %.3f ' % ('$e', d['value'], d['ms_per_step'], d['roofline']['frac']) + '  '.join('%s %.3f' % (n, k[n]['ms_per_step']) for n in ('gemm_big', 'gemm_big_tn', 'spmm') if n in k))
" || tail -5 gpurun_out/r6/ab_err.log
 done
done
