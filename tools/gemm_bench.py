#!/usr/bin/env python
"""Micro-benchmark of the library GEMM through gh_linear_fwd / gh_linear_bwd (C-ABI), HIP-event timed.
Used to iterate on the MFMA kernel in isolation:  python tools/gemm_bench.py [M K N reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from get_amd import _lib  # noqa: E402
from get_amd._lib import call, ptr, stream  # noqa: E402


def bench(m, k, n, reps=20, mode="fwd"):
    dev = "cuda:0"
    if os.environ.get("GEMM_MODE") in ("bf16", "fp32x3", "fp32x3p"):
        _lib.set_gemm_mode(os.environ["GEMM_MODE"])
    if os.environ.get("GEMM_MODE") == "fp32x3p":      # fresh operands below, possibly at recycled addresses: no pre-split image survives
        call("gh_weights_changed")
    x = torch.randn(m, k, device=dev)
    w = torch.randn(n, k, device=dev) / k ** 0.5
    wt = w.t().contiguous()
    b = torch.randn(n, device=dev)
    y = torch.empty(m, n, device=dev)
    g = torch.randn(m, n, device=dev)
    dx = torch.empty(m, k, device=dev)
    dw = torch.zeros(n, k, device=dev)

    def run():
        if mode == "fwd":
            call("gh_linear_fwd", ptr(x), ptr(w), ptr(b), ptr(y), m, k, n, stream())
        elif mode == "dx":
            call("gh_linear_bwd", ptr(x), ptr(wt), ptr(w), ptr(g), m, k, n, ptr(dx), None, None, stream())
        else:
            call("gh_linear_bwd", ptr(x), ptr(wt), ptr(w), ptr(g), m, k, n, None, ptr(dw), None, stream())

    for _ in range(40):          # long warm-up: the clock ramps for several ms after idle
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    tf = 2.0 * m * k * n / ms / 1e9
    if mode == "fwd":
        if os.environ.get("GEMM_MODE") == "bf16":       # compare against the same operand rounding
            ref = x.bfloat16().float() @ w.bfloat16().float().t() + b
        else:
            ref = (x.double() @ w.double().t() + b.double())
        err = float((y.double() - ref).abs().max())
    else:
        err = float("nan")
    print(f"{mode:4s} M={m:6d} K={k:5d} N={n:5d}: {ms:8.4f} ms  {tf:7.2f} TFLOP/s  ({100*tf/157.3:5.1f}% of f32 MFMA peak)  maxerr {err:.2e}")
    return tf


def floor_probe():
    """Fixed per-launch cost: tiny K (store/launch bound) vs a plain device copy of the same output."""
    for k in (16, 32, 64, 128, 304, 608, 1216, 2432):
        bench(96000, k, 300, reps=20)
    for k in (304, 608, 1216):
        bench(96000, k, 600, reps=20)
    x = torch.empty(96000, 300, device="cuda:0")
    y = torch.empty_like(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        y.copy_(x)
    e1.record()
    torch.cuda.synchronize()
    print(f"copy 115MB: {e0.elapsed_time(e1)/20:.4f} ms")


if __name__ == "__main__":
    _lib.load()
    _lib.ensure_workspace("cuda:0")
    if len(sys.argv) == 2 and sys.argv[1] == "floor":
        floor_probe()
    elif len(sys.argv) >= 4:
        m, k, n = map(int, sys.argv[1:4])
        reps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
        mode = sys.argv[5] if len(sys.argv) > 5 else "fwd"
        bench(m, k, n, reps, mode)
    else:
        for shape in [(96000, 300, 300), (96000, 600, 300), (98304, 304, 304), (96000, 1200, 300), (96000, 300, 600)]:
            bench(*shape)
        bench(96000, 300, 300)          # repeated: clock / cache state vs the first line
        bench(96000, 300, 300, mode="dx")
        bench(96000, 300, 300)
        bench(96000, 300, 300, mode="dw")
        bench(960, 1628, 300)
        bench(32, 3556, 300)

