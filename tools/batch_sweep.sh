#!/bin/bash
# configs[3]'s global batch on ONE GPU (the N = 1 point of the strong-scaling curve) and a per-GPU batch sweep.
# (12 warm-up steps: at these sizes the first steps of a fresh process include allocator growth; with the default 5 the first timed block was up to 20x slow)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03; mkdir -p $O
for args in "--global-batch 256" "--global-batch 256 --evd-dist snopes" "--batch 64" "--batch 128" "--batch 64 --evd-dist snopes" "--batch 256 --evd-dist snopes"; do
  n=$(echo $args | tr -d ' -' )
  timeout 900 python bench.py $args --steps 10 --warmup 12 --no-cpu-baseline --no-series --no-side-modes --no-profile > $O/sweep_$n.json 2> $O/sweep_$n.err
  python - <<P
import json
try:
    d = json.loads([l for l in open("$O/sweep_$n.json").read().splitlines() if l.startswith("{")][-1])
    print("$args: %.0f pairs/s %.3f ms/step pairs/gpu %.0f claims/s %.0f scaling %s" % (d["value"], d["ms_per_step"], d["config"]["pairs_per_gpu"], d["claims_per_s"], d["scaling"]))
except Exception as e:
    print("$args FAILED", e, open("$O/sweep_$n.err").read()[-600:])
P
done
