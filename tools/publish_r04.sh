#!/bin/bash
# copy the evidence of gpurun_out/r04 (tools/collect_r04.sh) into profiles/ under the round's names
cd "$(dirname "$0")/.."
O=gpurun_out/r04
cp $O/bench_line.json profiles/r04_bench_line.json
cp $O/bench_under_rocprof.json profiles/r04_bench_line_under_rocprof.json
cp $O/bench_kernel_stats.csv profiles/r04_bench_kernel_stats.csv
cp $O/pmc_traffic.json profiles/pmc_traffic.json
sed -i 's#tools/collect_r03.sh#tools/collect_r04.sh#' profiles/pmc_traffic.json
cat $O/pmc_SQ_BUSY_CYCLES_SQ_WAVES_.txt $O/pmc_FETCH_SIZE.txt $O/pmc_WRITE_SIZE.txt $O/pmc_TCC_HIT_sum_TCC_MISS_sum.txt > profiles/r04_pmc_bench_step.txt
cp $O/step_timeline.txt profiles/r04_step_timeline.txt
cp $O/step_timeline_snopes.txt profiles/r04_step_timeline_snopes.txt
cp $O/cfg2_bench_line.json profiles/r04_cfg2_bench_line.json
cat $O/cfg4_fp32_bench_line.json $O/cfg4_bf16_bench_line.json > profiles/r04_cfg4_bench_lines.json
cp $O/bench_fp32x3p.json profiles/r04_bench_fp32x3p.json
cp $O/bench_collective_library.json profiles/r04_bench_collective_library.json
cp $O/gemm_bench_modes.txt profiles/r04_gemm_bench_modes.txt
cp $O/spmm_bench.txt profiles/r04_spmm_bench.txt
cp $O/bench_2rank_gloo.json profiles/r04_bench_2rank_gloo_selfspawn.json
cp $O/bench_2rank_gloo_gb64_snopes.json profiles/r04_bench_2rank_gloo_gb64_snopes_striped.json
cp $O/bench_gpus2_on_1gpu_box.out profiles/r04_bench_gpus2_on_a_1gpu_box.txt
cp $O/batch_sweep.txt profiles/r04_batch_sweep.txt
[ -f $O/soak.json ] && cp $O/soak.json profiles/r04_soak.json
echo "published build $(cat $O/commit.txt)"
