#!/bin/bash
# configs[4] bf16 step: product library vs the tool build (libget_hip_measure.so) alternating on one box + the bf16 parity tests on the tool build
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6
if [ -z "$NO_TESTS" ]; then
GET_AMD_LIB=$GRAFT_REPO_ROOT/get_amd/lib/libget_hip_measure.so timeout 900 python -m pytest -x -q -m gpu tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_wide_composite.py tests/test_gpu_fullsize_grads.py -k "bf16 or 256_tile" 2>&1 | tail -3
fi
C4="--hidden 768 --word-heads 8 --window 5 --gsl-rate 0.8 --gemm-mode bf16 --no-strong --no-other-configs --no-series --no-side-modes --no-cpu-baseline"
for rep in 1 2 3; do
for v in product measure; do
if [ $v = product ]; then A=""; else A="--measure-build"; fi
python bench.py $A $C4 $AB_ARGS 2>/dev/null | tail -1 | python -c "
import sys, json
d=json.loads(sys.stdin.read()); k=d['kernels']; print('$v', round(d['value']), round(d['roofline']['frac'],4), ' '.join('%s %.3f' % (n, k[n]['ms_per_step']) for n in ('gemm_big','gemm_big_tn','spmm')), 'tn frac %.3f' % k['gemm_big_tn'].get('frac_of_peak', k['gemm_big_tn'].get('frac', 0)))"
done
done
