#!/bin/bash
# fp32x3 modes: tests + bench lines.  usage: tools/x3p_r3.sh
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -k "fp32x3" 2>&1 | tail -5
for m in fp32 fp32x3p; do
  timeout 600 python bench.py --steps 20 --warmup 6 --no-cpu-baseline --no-series --no-side-modes --gemm-mode $m > $O/x3p_$m.json 2> $O/x3p_$m.err; tail -2 $O/x3p_$m.err
  python - <<P
import json
d = json.loads([l for l in open("$O/x3p_$m.json").read().splitlines() if l.startswith("{")][-1])
print("$m pairs/s %.0f ms/step %.3f" % (d["value"], d["ms_per_step"]))
for k, v in d.get("kernels", {}).items():
    if k.startswith("gemm"):
        print("   %-16s %7.4f ms/step  %5.1f launches  %8.1f TF frac %.3f" % (k, v["ms_per_step"], v["launches_per_step"], v.get("achieved_tflops", 0), v["frac"]))
P
done
