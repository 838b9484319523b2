#!/usr/bin/env python
"""Debug aid: one gh_ggnn_cell_fwd on random inputs, every intermediate dumped to an .npz (run under two GET_AMD_LIB builds, diff)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from get_amd import _lib, ops
from get_amd._lib import call, ptr, stream
_lib.load(); _lib.ensure_workspace("cuda:0")
n, r, din, h = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
rng = np.random.default_rng(5)
D = "cuda:0"
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(D)
toks = rng.integers(2, 50, size=(n, r)).astype(np.int64)
lens = np.full((n,), r, np.int64)
adj, ids, nn = ops.graph_build(T(toks), T(lens), 3)
x = T(rng.standard_normal((n * r, din)).astype(np.float32))
W = lambda a, b: T((rng.standard_normal((a, b)) / np.sqrt(b)).astype(np.float32))
wp, wz0, wz1, wr0, wr1, wh0, wh1 = W(h, din), W(h, h), W(h, h), W(h, h), W(h, h), W(h, h), W(h, h)
B = lambda: T(rng.standard_normal((h,)).astype(np.float32) * 0.1)
bz0, bz1, br0, br1, bh0, bh1 = B(), B(), B(), B(), B(), B()
bufs = {k: torch.full((n * r, h), float("nan"), device=D) for k in ("xp", "a", "z", "rr", "rx", "hh", "out")}
call("gh_ggnn_cell_fwd", ptr(adj.bits), ptr(adj.dinv), None, None, None, n * r, n * r, ptr(x), None, n, r, din, h,
     ptr(wp), ptr(wz0), ptr(wz1), ptr(wr0), ptr(wr1), ptr(wh0), ptr(wh1), ptr(bz0), ptr(bz1), ptr(br0), ptr(br1), ptr(bh0), ptr(bh1),
     *[ptr(bufs[k]) for k in ("xp", "a", "z", "rr", "rx", "hh", "out")], 0.0, 0, None, None, 0.0, 0, stream())
torch.cuda.synchronize()
np.savez(sys.argv[1], **{k: v.cpu().numpy() for k, v in bufs.items()})
