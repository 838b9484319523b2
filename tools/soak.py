#!/usr/bin/env python
"""Soak run of the training step (default bench workload, composite path): N steps over rotating batches, then checks that
nothing drifted -- loss finite and decreasing on the (memorisable) resident batches, device memory flat after warm-up, parity
against the CPU oracle still at the 1e-4 bar with the TRAINED weights.  usage: python tools/soak.py [steps] [cfg4bf16]
(cfg4bf16: BASELINE configs[4] in the bf16 storage mode -- finite, decreasing, memory flat; its parity bound is the bf16 one)"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from get_amd import _lib, ops  # noqa: E402
from get_amd.dist import FlatTrainer  # noqa: E402
from get_amd.synth import SynthConfig  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    bf16 = len(sys.argv) > 2 and sys.argv[2] == "cfg4bf16"
    _lib.load()
    cfg = SynthConfig(batch=32, n_evd=30)
    if bf16:
        cfg = SynthConfig(batch=32, n_evd=30, hidden=768, emb_dim=768, word_heads=8, window=5, gsl_rate=0.8)
        _lib.set_gemm_mode("bf16")
        ops.bump_weight_epoch()
    wl = bench.build_workload(seed=20240229, device="cuda:0", cfg=cfg, n_batches=4)
    model = wl["model"].train(True)
    tr = FlatTrainer(model, lr=1e-4, weight_decay=1e-3)
    ops.bump_weight_epoch()
    losses, mem = [], []
    t0 = time.time()
    for i in range(steps):
        b = wl["batches"][i % 4]
        tr.zero_grad()
        q, d, k = b.inputs()
        loss = ops.cross_entropy(model(q, d, **k), b.labels)
        ops.backward(loss)
        tr.step()
        if i % 500 == 0 or i == steps - 1:
            losses.append((i, float(loss.item())))
            mem.append(torch.cuda.memory_allocated() / 2 ** 20)
    torch.cuda.synchronize()
    dt = time.time() - t0
    par = bench.parity_check(wl)
    out = {"steps": steps, "seconds": dt, "pairs_per_s_incl_loss_readbacks": 960 * steps / dt, "loss_first": losses[0], "loss_last": losses[-1],
           "losses": losses[::4], "memory_allocated_mib_first_last_max": [mem[1] if len(mem) > 1 else mem[0], mem[-1], max(mem)],
           "parity_after_training": par, "all_finite": bool(torch.isfinite(tr.flat_p).all())}
    out["mode"] = "configs[4] bf16 storage" if bf16 else "configs[1] fp32"
    print(json.dumps(out))
    assert out["all_finite"] and losses[-1][1] < losses[0][1]
    assert bf16 or par["max_abs_logit_diff_vs_cpu_oracle"] <= 1e-4
    assert abs(mem[-1] - mem[1]) < 64, "device memory grew during the run"


if __name__ == "__main__":
    main()
