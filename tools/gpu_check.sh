#!/bin/bash
# One GPU-box round trip: GPU tests, GEMM microbench, the bench line + a per-kernel summary.  usage: tools/gpu_check.sh TAG [pytest args]
TAG=${1:-x}; shift
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2; mkdir -p $O
python -m pytest tests -m gpu -q -x "$@" 2>&1 | tail -15 > $O/pytest_$TAG.log
python tools/gemm_bench.py > $O/gemm_bench_$TAG.log 2>&1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$TAG.json 2> $O/bench_$TAG.err
tail -6 $O/pytest_$TAG.log; grep -v amdgpu.ids $O/gemm_bench_$TAG.log; tail -3 $O/bench_$TAG.err
python - <<P
import json
d=json.load(open("$O/bench_$TAG.json"))
print("pairs/s %.0f  ms/step %.3f  dominant %.1f TF" % (d["value"], d["ms_per_step"], d["roofline"]["achieved"]))
for k,v in d["kernels"].items(): print("  %-16s %7.4f ms/step %5.1f launches  %8.1f %s" % (k, v["ms_per_step"], v["launches_per_step"], v.get("achieved_tflops",v.get("achieved_gbps",0)), "TF" if "achieved_tflops" in v else "GB/s"))
P
