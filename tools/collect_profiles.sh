#!/bin/bash
# Evidence run behind profiles/r01_*: bench line, rocprofv3 kernel stats and the separate --pmc passes (run on the GPU box from the repo root).
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r01b
python bench.py --steps 20 --warmup 5 > gpurun_out/r01b/bench_line.json 2> gpurun_out/r01b/bench_err.log
tail -c 600 gpurun_out/r01b/bench_line.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r01b/stats -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r01b/bench_under_rocprof.json 2> gpurun_out/r01b/rocprof_err.log
for c in FETCH_SIZE WRITE_SIZE "SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES" "TCC_HIT_sum TCC_MISS_sum"; do
  d=gpurun_out/r01b/pmc_$(echo $c | tr ' ' '_' | cut -c1-24)
  timeout 600 rocprofv3 --pmc $c --output-format csv -d $d -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile > $d.log 2>&1
  python tools/pmc_sum.py $d > $d.txt 2>&1
  rm -rf $d
done
ls -la gpurun_out/r01b
