#!/bin/bash
# Evidence run behind profiles/r02_*: bench line, rocprofv3 kernel stats and the separate --pmc passes (GPU box, repo root).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02; mkdir -p $O
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_err.log
tail -c 300 $O/bench_line.json
rm -rf /tmp/stats
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/stats -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-series > $GRAFT_REPO_ROOT/$O/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$O/rocprof_err.log )
f=$(find /tmp/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/bench_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE "SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES" "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | tr ' ' '_' | cut -c1-24)
  d=/tmp/pmc_$n; rm -rf $d
  ( cd /tmp && timeout 600 rocprofv3 --pmc $c --output-format csv -d $d -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-series > $GRAFT_REPO_ROOT/$O/pmc_$n.log 2>&1 )
  python tools/pmc_sum.py $d > $O/pmc_$n.txt 2>&1
done
# BASELINE configs[4] (h=768): MFMA pipe busy, fp32 vs bf16 storage
C4="--hidden 768 --word-heads 8 --window 5 --gsl-rate 0.8 --batch 32 --no-cpu-baseline --no-series --no-profile --steps 3 --warmup 1"
for M in fp32 bf16; do
  d=/tmp/pmc_cfg4_$M; rm -rf $d
  ( cd /tmp && timeout 600 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $d -- python $GRAFT_REPO_ROOT/bench.py $C4 --gemm-mode $M > $GRAFT_REPO_ROOT/$O/pmc_cfg4_$M.log 2>&1 )
  python tools/pmc_sum.py $d > $O/pmc_cfg4_$M.txt 2>&1
done
ls -la $O
