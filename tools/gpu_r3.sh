#!/bin/bash
# Round-3 evidence pass on one GPU box: GPU tests, the default bench line.  usage: tools/gpu_r3.sh TAG [pytest -k expr]
TAG=${1:-x}; KEXPR=${2:-}
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3; mkdir -p $O
if [ -n "$KEXPR" ]; then
  timeout 1500 python -m pytest tests -m gpu -q -x -k "$KEXPR" 2>&1 | tail -25 > $O/pytest_$TAG.log
else
  timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -25 > $O/pytest_$TAG.log
fi
tail -6 $O/pytest_$TAG.log
timeout 900 python bench.py > $O/bench_$TAG.json 2> $O/bench_$TAG.err; tail -3 $O/bench_$TAG.err
python - <<P
import json
try:
    d = json.loads([l for l in open("$O/bench_$TAG.json").read().splitlines() if l.startswith("{")][-1])
    print("pairs/s %.0f ms/step %.3f n_gpus %d" % (d["value"], d["ms_per_step"], d["n_gpus"]))
    for k in ("timed", "roofline", "step_split_ms", "parity", "realistic_series", "other_regimes", "collective"):
        if k in d: print("  ", k, json.dumps(d[k])[:600])
    for k, v in d.get("kernels", {}).items():
        r = v.get("achieved_gbps", v.get("achieved_tflops", 0))
        print("   %-16s %7.4f ms/step  %5.1f launches  %8.1f %s  frac %.3f" % (k, v["ms_per_step"], v["launches_per_step"], r, "GB/s" if "achieved_gbps" in v else "TF", v["frac"]))
    cb = d.get("cpu_baseline"); print("  cpu", json.dumps(cb)[:300])
except Exception as e:
    print("bench FAILED", e)
P
