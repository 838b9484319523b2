C4="--hidden 768 --word-heads 8 --window 5 --gsl-rate 0.8 --gemm-mode bf16 --no-strong --no-other-configs --measure-lib pp"
GH_BF16_TILE=256 tools/trace_r6.sh pp256 $C4
GH_BF16_TILE=256 GH_DBG=1 tools/trace_r6.sh pp256_noepi $C4
grep "gemm_nt_kernel<2, 4, 4, 8" gpurun_out/r6/trace_pp256.txt
echo ---; grep "gemm_nt_kernel<2, 4, 4, 8" gpurun_out/r6/trace_pp256_noepi.txt
