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
    adj, _, n_nodes = ops.graph_build(torch.from_numpy(toks).to(dev), torch.from_numpy(lens).to(dev), 3)
    nnz = float(torch.count_nonzero(adj.to_dense())) / n
    deg = adj.to_dense().ne(0).sum(-1).max().item()
    ms = timeit(lambda: _lib.call("gh_spmm", *adj._args(), None, 0, x.data_ptr(), y.data_ptr(), n, r, h, 0, 0, _lib.stream()))
    print(f"{name:9s} nnz/graph {nnz:7.1f} max degree {deg:3d}: {ms*1e3:7.1f} us  {2*n*r*h*4/ms/1e6:7.1f} GB/s")
    # node-compact layout (what the training step runs): the real nodes of all graphs back to back
    goff = torch.zeros(n + 1, device=dev, dtype=torch.int32)
    goff[1:] = torch.cumsum(n_nodes, 0).to(torch.int32)
    m_real = int(goff[-1])
    real = (torch.arange(r, device=dev)[None, :] < n_nodes[:, None])
    xc = x[real].contiguous()
    yc = torch.empty_like(xc)
    msc = timeit(lambda: _lib.call("gh_spmm", *adj._args(), goff.data_ptr(), m_real, xc.data_ptr(), yc.data_ptr(), n, r, h, 0, 0, _lib.stream()))
    # reference: the padded launch on an input whose padding rows are zero
    xz = torch.where(real[..., None], x, torch.zeros_like(x))
    _lib.call("gh_spmm", *adj._args(), None, 0, xz.data_ptr(), y.data_ptr(), n, r, h, 0, 0, _lib.stream())
    err = float((y[real] - yc).abs().max())
    dense = (adj.to_dense().double() @ xz.double())[real]
    err64 = float((dense - yc.double()).abs().max())
    print(f"{name:9s} compact m_real {m_real} ({m_real / n:.1f} rows/graph): {msc*1e3:7.1f} us  {2*m_real*h*4/msc/1e6:7.1f} GB/s  "
          f"|compact - padded| {err:.2e}  |compact - dense f64| {err64:.2e}")
try:      # tool build: per-phase ticks of the list kernel (thread 0 of every workgroup)
    import ctypes
    L = _lib.load()
    L.gh_debug_spmm_phases.argtypes = [ctypes.c_void_p, ctypes.c_int]
    buf = (ctypes.c_uint * (8192 * 8))()
    L.gh_debug_spmm_phases(None, 1)
    _lib.call("gh_spmm", *adj._args(), goff.data_ptr(), m_real, xc.data_ptr(), yc.data_ptr(), n, r, h, 0, 0, _lib.stream())
    torch.cuda.synchronize()
    L.gh_debug_spmm_phases(buf, 1)
    a = np.frombuffer(buf, dtype=np.uint32).reshape(8192, 8)[:, :7].astype(np.float64)
    a = a[a.sum(1) > 0]                     # workgroups that ran (one per graph, or one per slab with GH_SPMM_SPW=1)
    names = ["issue", "rowwords+scan", "barrier1", "listbuild", "slabwait", "barrier2", "aggregate"]
    print(f"phases, thread 0 of each of {a.shape[0]} workgroups, s_memtime ticks (~0.5 ns; with several slabs per workgroup the slab phases hold the LAST slab), mean / p90: " + ", ".join(f"{nm} {a[:, i].mean():.2f}/{np.percentile(a[:, i], 90):.2f}" for i, nm in enumerate(names)) + f"  total {a.sum(1).mean():.2f}")
except Exception as e:
    print("no phase instrumentation:", e)
# the bf16 storage pipeline's aggregation at the configs[4] shape (h = 768, window 5, node-compact), with its phase ticks
try:
    import ctypes
    h16 = 768
    toks16 = make_tokens(rng, n, r, 20000, r, r)[0]
    adj16, _, nn16 = ops.graph_build(torch.from_numpy(toks16).to(dev), torch.from_numpy(np.full((n,), r, np.int32)).to(dev), 5)
    goff16 = torch.zeros(n + 1, device=dev, dtype=torch.int32)
    goff16[1:] = torch.cumsum(nn16, 0).to(torch.int32)
    m16 = int(goff16[-1])
    x16 = torch.randn(m16, h16, device=dev).to(torch.bfloat16)
    y16 = torch.empty_like(x16)
    nnz16 = float(torch.count_nonzero(adj16.to_dense())) / n
    for acc in (0, 1):
        ms16 = timeit(lambda: _lib.call("gh_spmm_bf16", *adj16._args(), goff16.data_ptr(), m16, x16.data_ptr(), y16.data_ptr(), n, r, h16, 0, acc, _lib.stream()))
        print(f"bf16 h=768 window 5 compact m_real {m16} nnz/graph {nnz16:.1f} accumulate {acc}: {ms16*1e3:7.1f} us  {(2 + acc)*m16*h16*2/ms16/1e6:7.1f} GB/s")
    L = _lib.load()
    L.gh_debug_spmm_phases.argtypes = [ctypes.c_void_p, ctypes.c_int]
    buf = (ctypes.c_uint * (8192 * 8))()
    L.gh_debug_spmm_phases(None, 1)
    _lib.call("gh_spmm_bf16", *adj16._args(), goff16.data_ptr(), m16, x16.data_ptr(), y16.data_ptr(), n, r, h16, 0, 0, _lib.stream())
    torch.cuda.synchronize()
    L.gh_debug_spmm_phases(buf, 1)
    a = np.frombuffer(buf, dtype=np.uint32).reshape(8192, 8)[:, :7].astype(np.float64)
    a = a[a.sum(1) > 0]
    names = ["issue", "rowwords+scan", "barrier1", "listbuild", "slabwait", "barrier2", "aggregate"]
    if a.shape[0] == 0:
        raise RuntimeError("no phase ticks: the matrix-pipe kernel ran (GH_SPMM_MFMA=0 selects the instrumented edge-list kernel)")
    print(f"bf16 phases, thread 0 of each of {a.shape[0]} workgroups (two slabs each: the slab phases hold the LAST slab), mean / p90: " + ", ".join(f"{nm} {a[:, i].mean():.2f}/{np.percentile(a[:, i], 90):.2f}" for i, nm in enumerate(names)) + f"  total {a.sum(1).mean():.2f}")
except Exception as e:
    print("bf16 section failed:", e)
ms = timeit(lambda: y.copy_(x))
print(f"copy      {ms*1e3:7.1f} us  {2*n*r*h*4/ms/1e6:7.1f} GB/s")
