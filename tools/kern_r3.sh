#!/bin/bash
# quick check: composite tests + per-kernel table of the bench step.  usage: tools/kern_r3.sh TAG
TAG=${1:-x}
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -q -x -k "composite or golden or concat_att or compact" 2>&1 | tail -3
timeout 600 python bench.py --steps 20 --warmup 6 --no-cpu-baseline --no-series --no-side-modes > $O/kern_$TAG.json 2> $O/kern_$TAG.err; tail -2 $O/kern_$TAG.err
python - <<P
import json
d = json.loads([l for l in open("$O/kern_$TAG.json").read().splitlines() if l.startswith("{")][-1])
print("pairs/s %.0f ms/step %.3f" % (d["value"], d["ms_per_step"]), json.dumps(d.get("step_split_ms"))[:200])
for k, v in d.get("kernels", {}).items():
    r = v.get("achieved_gbps", v.get("achieved_tflops", 0))
    print("   %-16s %7.4f ms/step  %5.1f launches  %8.1f %s  frac %.3f" % (k, v["ms_per_step"], v["launches_per_step"], r, "GB/s" if "achieved_gbps" in v else "TF", v["frac"]))
P
