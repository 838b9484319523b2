#!/bin/bash
# configs[4] bf16 step on the tool build: weight gradients on the 128 x 320 kernel (GH_TN_PP_ROWS huge) vs the 256 x 256 ping-pong kernel, alternating on one box
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6
export GET_AMD_LIB=$GRAFT_REPO_ROOT/get_amd/lib/libget_hip_measure.so
if [ -z "$NO_TESTS" ]; then
timeout 1200 python -m pytest -x -q -m gpu tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_wide_composite.py tests/test_gpu_fullsize_grads.py -k "bf16 or 256_tile" 2>&1 | tail -5
fi
C4="--measure-build --hidden 768 --word-heads 8 --window 5 --gsl-rate 0.8 --gemm-mode bf16 --no-strong --no-other-configs --no-series --no-side-modes --no-cpu-baseline"
for rep in 1 2 3; do
for v in old pp ${AB_EXTRA}; do
case $v in
old) export GH_TN_PP_ROWS=1000000000; unset GH_TN_PP_KS;;
pp) unset GH_TN_PP_ROWS; unset GH_TN_PP_KS;;
ks*) unset GH_TN_PP_ROWS; export GH_TN_PP_KS=${v#ks};;
esac
python bench.py $C4 $AB_ARGS 2>/dev/null | tail -1 | python -c "
import sys, json
d=json.loads(sys.stdin.read()); k=d['kernels']; print('$v', round(d['value']), round(d['roofline']['frac'],4), ' '.join('%s %.3f' % (n, k[n]['ms_per_step']) for n in ('gemm_big','gemm_big_tn','spmm')), 'tn frac %.3f' % k['gemm_big_tn'].get('frac_of_peak', k['gemm_big_tn'].get('frac', 0)), 'parity', d.get('parity', {}).get('max_abs_logit_diff_vs_cpu_oracle'))"
done
done
