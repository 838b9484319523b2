#!/usr/bin/env python
"""Aggregation (spmm) micro-benchmark: Zipf word graphs vs near-diagonal graphs, both kernel variants,
and a plain device copy of the same bytes for reference."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from get_amd import _lib, ops  # noqa: E402
from get_amd.synth import make_tokens  # noqa: E402

dev = "cuda:0"
n, r, h = 960, 100, 300
rng = np.random.default_rng(0)


def timeit(fn, reps=20):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


x = torch.randn(n, r, h, device=dev)
y = torch.empty_like(x)
for name, toks in (("zipf", make_tokens(rng, n, r, 20000, r, r)[0]),
                   ("distinct", (np.arange(n * r).reshape(n, r) % 19000 + 2).astype(np.int32))):
    lens = np.full((n,), r, np.int32)
    adj, _, _ = ops.graph_build(torch.from_numpy(toks).to(dev), torch.from_numpy(lens).to(dev), 3)
    nnz = float(torch.count_nonzero(adj.to_dense())) / n
    deg = adj.to_dense().ne(0).sum(-1).max().item()
    ms = timeit(lambda: _lib.call("gh_spmm", *adj._args(), None, 0, x.data_ptr(), y.data_ptr(), n, r, h, 0, 0, _lib.stream()))
    print(f"{name:9s} nnz/graph {nnz:7.1f} max degree {deg:3d}: {ms*1e3:7.1f} us  {2*n*r*h*4/ms/1e6:7.1f} GB/s")
ms = timeit(lambda: y.copy_(x))
print(f"copy      {ms*1e3:7.1f} us  {2*n*r*h*4/ms/1e6:7.1f} GB/s")
