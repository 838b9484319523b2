#!/usr/bin/env python
"""Is the training step limited by the host's launch rate?  Times N steps two ways: host time to ISSUE them (no sync) and
wall time until the device has finished; plus the host time of forward / backward / optimiser issue separately with the
device idle-synced before each phase (pure Python + ctypes + autograd cost).   python tools/host_bound.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from get_amd.dist import FlatTrainer  # noqa: E402
from get_amd import ops  # noqa: E402

import sys as _s
wl = bench.build_workload(device="cuda:0", n_batches=4, evd_dist=("snopes" if "snopes" in _s.argv else "fixed"))
model = wl["model"]
trainer = FlatTrainer(model, lr=1e-4, weight_decay=1e-3)
ops.bump_weight_epoch()
model.train(True)
batches = wl["batches"]


def step(i):
    b = batches[i % 4]
    trainer.zero_grad()
    q, d, k = b.inputs()
    loss = torch.nn.functional.cross_entropy(model(q, d, **k), b.labels)
    loss.backward()
    trainer.step()


for i in range(8):
    step(i)
torch.cuda.synchronize()
N = 30
t0 = time.perf_counter()
for i in range(N):
    step(i)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"issue {1e3*(t1-t0)/N:.3f} ms/step, complete {1e3*(t2-t0)/N:.3f} ms/step")
# host-only cost per phase (device drained before each phase, so nothing blocks on queue depth)
acc = [0.0, 0.0, 0.0, 0.0]
for i in range(10):
    b = batches[i % 4]
    torch.cuda.synchronize(); a = time.perf_counter()
    trainer.zero_grad(); q, d, k = b.inputs()
    torch.cuda.synchronize(); c0 = time.perf_counter()
    loss = torch.nn.functional.cross_entropy(model(q, d, **k), b.labels)
    c1 = time.perf_counter(); torch.cuda.synchronize(); c2 = time.perf_counter()
    loss.backward()
    c3 = time.perf_counter(); torch.cuda.synchronize(); c4 = time.perf_counter()
    trainer.step()
    c5 = time.perf_counter()
    acc[0] += c0 - a; acc[1] += c1 - c0; acc[2] += c3 - c2; acc[3] += c5 - c4
print("host ms: inputs+zero_grad %.3f, forward issue %.3f, backward issue %.3f, optimiser issue %.3f" % tuple(1e2 * x for x in acc))
