#!/bin/bash
# Evidence run behind profiles/r06_*: bench line, rocprofv3 kernel stats, the separate --pmc passes, the other BASELINE configs with
# their kernel tables and MFMA-busy counters, step timelines, vendor-BLAS ceilings, the 2- and 8-rank rehearsals (GPU box, repo root).
# usage: tools/collect_r06.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; rm -rf $O; mkdir -p $O
[ -f tools/_commit.txt ] && cp tools/_commit.txt $O/commit.txt
timeout 1200 python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_err.log
tail -c 300 $O/bench_line.json
rm -rf /tmp/stats
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/stats -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-series --no-side-modes --no-other-configs > $GRAFT_REPO_ROOT/$O/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$O/rocprof_err.log )
f=$(find /tmp/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/bench_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE "SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES" "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | tr ' ' '_' | cut -c1-24)
  d=/tmp/pmc_$n; rm -rf $d
  ( cd /tmp && timeout 600 rocprofv3 --pmc $c --output-format csv -d $d -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-series --no-side-modes --no-other-configs --no-strong > $GRAFT_REPO_ROOT/$O/pmc_$n.log 2>&1 )
  python tools/pmc_sum.py $d > $O/pmc_$n.txt 2>&1
done
python tools/pmc_traffic.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE "$(cat $O/commit.txt 2>/dev/null)" > $O/pmc_traffic.json 2>> $O/bench_err.log
bash tools/trace_r6.sh r06 > /dev/null 2>&1; cp gpurun_out/r6/trace_r06.txt $O/step_timeline.txt
bash tools/trace_r6.sh r06_snopes --evd-dist snopes > /dev/null 2>&1; cp gpurun_out/r6/trace_r06_snopes.txt $O/step_timeline_snopes.txt
# BASELINE configs[2] (PolitiFact-shaped) and configs[4] (h = 768, fp32 and bf16 storage): full lines with their kernel tables
timeout 600 python bench.py --len-right 200 --n-evd 10 --batch 64 --no-cpu-baseline --no-series > $O/cfg2_bench_line.json 2>> $O/bench_err.log
C4="--hidden 768 --word-heads 8 --window 5 --gsl-rate 0.8 --batch 32 --no-cpu-baseline --no-series"
timeout 600 python bench.py $C4 --gemm-mode fp32 > $O/cfg4_fp32_bench_line.json 2>> $O/bench_err.log
timeout 600 python bench.py $C4 --gemm-mode bf16 > $O/cfg4_bf16_bench_line.json 2>> $O/bench_err.log
bash tools/trace_r6.sh r06_cfg4bf16 $C4 --gemm-mode bf16 > /dev/null 2>&1; cp gpurun_out/r6/trace_r06_cfg4bf16.txt $O/step_timeline_cfg4_bf16.txt
# configs[4] bf16: MFMA pipe busy of the round-5 kernels (own --pmc pass, no tracing beside it) and the kernel stats
d=/tmp/pmc_cfg4; rm -rf $d
( cd /tmp && timeout 600 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $d -- python $GRAFT_REPO_ROOT/bench.py $C4 --gemm-mode bf16 --steps 3 --warmup 1 --no-profile --no-side-modes > $GRAFT_REPO_ROOT/$O/pmc_cfg4.log 2>&1 )
python tools/pmc_sum.py $d > $O/cfg4_pmc_mfma.txt 2>&1
d=/tmp/pmc_cfg4b; rm -rf $d
( cd /tmp && timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $d -- python $GRAFT_REPO_ROOT/bench.py $C4 --gemm-mode bf16 --steps 3 --warmup 1 --no-profile --no-side-modes > $GRAFT_REPO_ROOT/$O/pmc_cfg4b.log 2>&1 )
python tools/pmc_sum.py $d >> $O/cfg4_pmc_mfma.txt 2>&1
rm -rf /tmp/stats4
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/stats4 -o bench -- python $GRAFT_REPO_ROOT/bench.py $C4 --gemm-mode bf16 --steps 10 --warmup 3 --no-side-modes > /dev/null 2>> $GRAFT_REPO_ROOT/$O/rocprof_err.log )
f=$(find /tmp/stats4 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/cfg4_bf16_kernel_stats.csv
python tools/blas_ref.py 2>&1 | grep -v amdgpu.ids > $O/blas_ref.txt
# the standalone K-loop prototype of the 256 x 256 ping-pong tile at the cell's shapes (tools/nt256_proto.hip, built into tools/_bin)
[ -x tools/_bin/nt256_base ] && NT256_VARIANTS="base nostore" tools/nt256_run.sh > $O/nt256_proto.txt 2>&1
# ... and of the weight-gradient kernel on the same loop (tools/tn256_proto.hip)
[ -x tools/_bin/tn256_base ] && TN256_VARIANTS="base nostore" tools/tn256_run.sh > $O/tn256_proto.txt 2>&1
# opt-in fp32x3 with pre-split weights, the library-owned communicator, the aggregation micro-benchmark (tool build)
timeout 600 python bench.py --gemm-mode fp32x3p --no-cpu-baseline --no-series --no-side-modes > $O/bench_fp32x3p.json 2>> $O/bench_err.log
timeout 600 python bench.py --collective library --no-cpu-baseline --no-series --no-side-modes 2>> $O/bench_err.log | grep '^{' > $O/bench_collective_library.json
[ -f get_amd/lib/libget_hip_measure.so ] && GET_AMD_LIB=$GRAFT_REPO_ROOT/get_amd/lib/libget_hip_measure.so timeout 300 python tools/spmm_bench.py 2>&1 | grep -v amdgpu.ids > $O/spmm_bench.txt
# N > 1 code path on the 1-GPU box: bench.py spawns its own ranks (gloo, ranks share the device): 2 ranks, and the 8-rank rehearsal of
# configs[3] (global batch 256, Snopes counts, striped shards) plus its weak variant
GET_AMD_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 10 --warmup 3 > $O/bench_2rank_gloo.json 2> $O/bench_2rank.err
GET_AMD_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 8 --steps 6 --warmup 2 --global-batch 256 --evd-dist snopes --no-cpu-baseline > $O/bench_8rank_gloo_gb256_snopes.json 2> $O/bench_8rank.err
GET_AMD_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 8 --steps 6 --warmup 2 --no-cpu-baseline --no-strong > $O/bench_8rank_gloo_weak.json 2>> $O/bench_8rank.err
python bench.py --gpus 2 > $O/bench_gpus2_on_1gpu_box.out 2>&1; echo "rc=$?" >> $O/bench_gpus2_on_1gpu_box.out
bash tools/batch_sweep.sh > /dev/null 2>&1
timeout 900 python tools/soak.py 30000 2>/dev/null | grep '^{' > $O/soak.json
timeout 900 python tools/soak.py 10000 cfg4bf16 2>/dev/null | grep '^{' > $O/soak_cfg4_bf16.json
ls -la $O
