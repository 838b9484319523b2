#!/usr/bin/env python
"""cProfile of the host side of the training step (realistic evidence counts, where the step is close to host-bound).
    python tools/host_profile.py [steps]"""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from get_amd.dist import FlatTrainer  # noqa: E402
from get_amd import ops  # noqa: E402

wl = bench.build_workload(device="cuda:0", n_batches=4, evd_dist="snopes")
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
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
pr = cProfile.Profile()
pr.enable()
for i in range(N):
    step(i)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
