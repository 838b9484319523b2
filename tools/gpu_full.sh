#!/bin/bash
# Full evidence pass on one GPU box: GPU tests, the default bench line, a 2-rank rehearsal of the N>1 path (gloo, ranks share the GPU).
TAG=${1:-x}
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -k "not bf16" 2>&1 | tail -15 > $O/pytest_$TAG.log; tail -4 $O/pytest_$TAG.log
timeout 600 python bench.py > $O/bench_full_$TAG.json 2> $O/bench_full_$TAG.err; tail -3 $O/bench_full_$TAG.err
GET_AMD_BENCH_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 10 --warmup 3 > $O/bench_2rank_gloo_$TAG.json 2> $O/bench_2rank_$TAG.err; tail -3 $O/bench_2rank_$TAG.err
GET_AMD_BENCH_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 2 --steps 10 --warmup 3 --global-batch 64 > $O/bench_2rank_gloo_gb64_$TAG.json 2> $O/bench_2rank_gb_$TAG.err; tail -3 $O/bench_2rank_gb_$TAG.err
python - <<P
import json
for f in ("bench_full_$TAG", "bench_2rank_gloo_$TAG", "bench_2rank_gloo_gb64_$TAG"):
    try:
        d = json.loads([l for l in open("$O/" + f + ".json").read().splitlines() if l.startswith("{")][-1])   # (gloo prints its rank banner on stdout)
    except Exception as e:
        print(f, "FAILED", e); continue
    print(f, "pairs/s %.0f ms/step %.3f n_gpus %d scaling %s claims/s %.0f" % (d["value"], d["ms_per_step"], d["n_gpus"], d["scaling"], d["claims_per_s"]))
    for k in ("step_split_ms", "parity", "cpu_baseline", "cpu_baseline_probe", "realistic_series"):
        if k in d: print("   ", k, json.dumps(d[k])[:400])
P
