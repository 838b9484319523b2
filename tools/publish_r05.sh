#!/bin/bash
# copy the evidence of gpurun_out/r05 (tools/collect_r05.sh) into profiles/ under the round's names
cd "$(dirname "$0")/.."
O=gpurun_out/r05
cp $O/bench_line.json profiles/r05_bench_line.json
cp $O/bench_under_rocprof.json profiles/r05_bench_line_under_rocprof.json
cp $O/bench_kernel_stats.csv profiles/r05_bench_kernel_stats.csv
cp $O/pmc_traffic.json profiles/pmc_traffic.json
cat $O/pmc_SQ_BUSY_CYCLES_SQ_WAVES_.txt $O/pmc_FETCH_SIZE.txt $O/pmc_WRITE_SIZE.txt $O/pmc_TCC_HIT_sum_TCC_MISS_sum.txt > profiles/r05_pmc_bench_step.txt
cp $O/step_timeline.txt profiles/r05_step_timeline.txt
cp $O/step_timeline_snopes.txt profiles/r05_step_timeline_snopes.txt
cp $O/step_timeline_cfg4_bf16.txt profiles/r05_step_timeline_cfg4_bf16.txt
cp $O/cfg2_bench_line.json profiles/r05_cfg2_bench_line.json
cat $O/cfg4_fp32_bench_line.json $O/cfg4_bf16_bench_line.json > profiles/r05_cfg4_bench_lines.json
cp $O/cfg4_pmc_mfma.txt profiles/r05_cfg4_pmc_mfma.txt
cp $O/cfg4_bf16_kernel_stats.csv profiles/r05_cfg4_bf16_kernel_stats.csv
cp $O/blas_ref.txt profiles/r05_blas_ref.txt
cp $O/bench_fp32x3p.json profiles/r05_bench_fp32x3p.json
cp $O/bench_collective_library.json profiles/r05_bench_collective_library.json
cp $O/spmm_bench.txt profiles/r05_spmm_bench.txt
cp $O/bench_2rank_gloo.json profiles/r05_bench_2rank_gloo_selfspawn.json
cp $O/bench_8rank_gloo_gb256_snopes.json profiles/r05_bench_8rank_gloo_gb256_snopes.json
cp $O/bench_8rank_gloo_weak.json profiles/r05_bench_8rank_gloo_weak.json
cp $O/bench_gpus2_on_1gpu_box.out profiles/r05_bench_gpus2_on_a_1gpu_box.txt
cp $O/batch_sweep.txt profiles/r05_batch_sweep.txt
[ -f $O/soak.json ] && cp $O/soak.json profiles/r05_soak.json
echo "published build $(cat $O/commit.txt)"
