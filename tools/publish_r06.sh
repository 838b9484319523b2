#!/bin/bash
# copy the evidence of gpurun_out/r06 (tools/collect_r06.sh) into profiles/ under the round's names
cd "$(dirname "$0")/.."
O=gpurun_out/r06
cp $O/bench_line.json profiles/r06_bench_line.json
cp $O/bench_under_rocprof.json profiles/r06_bench_line_under_rocprof.json
cp $O/bench_kernel_stats.csv profiles/r06_bench_kernel_stats.csv
cp $O/pmc_traffic.json profiles/pmc_traffic.json
cat $O/pmc_SQ_BUSY_CYCLES_SQ_WAVES_.txt $O/pmc_FETCH_SIZE.txt $O/pmc_WRITE_SIZE.txt $O/pmc_TCC_HIT_sum_TCC_MISS_sum.txt > profiles/r06_pmc_bench_step.txt
cp $O/step_timeline.txt profiles/r06_step_timeline.txt
cp $O/step_timeline_snopes.txt profiles/r06_step_timeline_snopes.txt
cp $O/step_timeline_cfg4_bf16.txt profiles/r06_step_timeline_cfg4_bf16.txt
cp $O/cfg2_bench_line.json profiles/r06_cfg2_bench_line.json
cat $O/cfg4_fp32_bench_line.json $O/cfg4_bf16_bench_line.json > profiles/r06_cfg4_bench_lines.json
cp $O/cfg4_pmc_mfma.txt profiles/r06_cfg4_pmc_mfma.txt
cp $O/cfg4_bf16_kernel_stats.csv profiles/r06_cfg4_bf16_kernel_stats.csv
cp $O/blas_ref.txt profiles/r06_blas_ref.txt
cp $O/bench_fp32x3p.json profiles/r06_bench_fp32x3p.json
cp $O/bench_collective_library.json profiles/r06_bench_collective_library.json
cp $O/spmm_bench.txt profiles/r06_spmm_bench.txt
cp $O/bench_2rank_gloo.json profiles/r06_bench_2rank_gloo_selfspawn.json
cp $O/bench_8rank_gloo_gb256_snopes.json profiles/r06_bench_8rank_gloo_gb256_snopes.json
cp $O/bench_8rank_gloo_weak.json profiles/r06_bench_8rank_gloo_weak.json
cp $O/bench_gpus2_on_1gpu_box.out profiles/r06_bench_gpus2_on_a_1gpu_box.txt
cp $O/batch_sweep.txt profiles/r06_batch_sweep.txt
[ -f $O/soak.json ] && cp $O/soak.json profiles/r06_soak.json
[ -s $O/soak_cfg4_bf16.json ] && cp $O/soak_cfg4_bf16.json profiles/r06_soak_cfg4_bf16.json
[ -f $O/nt256_proto.txt ] && cp $O/nt256_proto.txt profiles/r06_nt256_proto.txt
[ -f $O/tn256_proto.txt ] && cp $O/tn256_proto.txt profiles/r06_tn256_proto.txt
echo "published build $(cat $O/commit.txt)"
