#!/bin/bash
# configs[3]'s global batch on ONE GPU (the N = 1 point of the strong-scaling curve) and a per-GPU batch sweep -> gpurun_out/r06/batch_sweep.txt
# (12 warm-up steps: at these sizes the first steps of a fresh process include allocator growth)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
[ -f tools/_commit.txt ] && echo "build $(cat tools/_commit.txt)" > $O/batch_sweep.txt
for args in "--batch 32" "--batch 32 --evd-dist snopes" "--batch 64" "--batch 128" "--global-batch 256" "--batch 64 --evd-dist snopes" "--batch 256 --evd-dist snopes" "--global-batch 256 --evd-dist snopes"; do
  n=$(echo $args | tr -d ' -' )
  timeout 900 python bench.py $args --steps 10 --warmup 12 --no-cpu-baseline --no-series --no-side-modes --no-profile --no-strong > $O/sweep_$n.json 2> $O/sweep_$n.err
  python - <<P | tee -a $O/batch_sweep.txt
import json
try:
    d = json.loads([l for l in open("$O/sweep_$n.json").read().splitlines() if l.startswith("{")][-1])
    t = d["timed"]
    print("$args: %.0f pairs/s %.3f ms/step pairs/gpu %.0f claims/s %.0f scaling %s blocks %d spread %.3f settle %s%s" % (d["value"], d["ms_per_step"], d["config"]["pairs_per_gpu"], d["claims_per_s"], d["scaling"], t["blocks"], t["spread_rel"], t.get("settle_block_ms_per_step"), "  UNSETTLED" if t.get("unsettled") else ""))
except Exception as e:
    print("$args FAILED", e, open("$O/sweep_$n.err").read()[-600:])
P
done
