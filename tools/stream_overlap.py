#!/usr/bin/env python
"""Do launches on two HIP streams fill each other's tails?  Times {NT GEMM, TN GEMM, streaming kernel} pairs issued
back to back on one stream vs concurrently on two streams (library kernels through the C-ABI, HIP events).
    python tools/stream_overlap.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from get_amd import _lib  # noqa: E402
from get_amd._lib import call, ptr  # noqa: E402

dev = "cuda:0"
M = 62128


def main():
    _lib.load()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for s in (s1, s2):
        with torch.cuda.stream(s):
            _lib.ensure_workspace(dev)
    x = torch.randn(M, 600, device=dev)
    w = torch.randn(600, 600, device=dev) / 24
    wt = w.t().contiguous()
    y = torch.empty(M, 600, device=dev)
    g = torch.randn(M, 300, device=dev)
    x3 = torch.randn(M, 300, device=dev)
    w3 = torch.randn(300, 300, device=dev)
    dws = [torch.zeros(300, 300, device=dev) for _ in range(7)]
    c0 = torch.randn(M * 300, device=dev)
    c1 = torch.empty_like(c0)
    torch.cuda.synchronize()

    def nt(s):
        call("gh_linear_fwd", ptr(x), ptr(w), None, ptr(y), M, 600, 600, s.cuda_stream)

    def tn(s):
        for d in dws:
            call("gh_linear_bwd", ptr(x3), ptr(w3), ptr(w3), ptr(g), M, 300, 300, None, ptr(d), None, s.cuda_stream)

    def cp(s):
        with torch.cuda.stream(s):
            for _ in range(4):
                c1.copy_(c0)

    ops = {"nt": nt, "tn": tn, "copy": cp}

    def timed(fa, fb, conc, reps=10):
        for _ in range(3):
            fa(s1); fb(s1)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s1)
        if conc:
            s2.wait_stream(s1)
        for _ in range(reps):
            fa(s1)
            fb(s2 if conc else s1)
        if conc:
            s1.wait_stream(s2)
        e1.record(s1)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    def alone(f, reps=10):
        for _ in range(3):
            f(s1)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s1)
        for _ in range(reps):
            f(s1)
        e1.record(s1)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    for _ in range(20):
        nt(s1)
    t = {k: alone(f) for k, f in ops.items()}
    print("alone ms:", {k: round(v, 4) for k, v in t.items()})
    for a, b in (("nt", "tn"), ("nt", "copy"), ("tn", "copy"), ("nt", "nt")):
        ser = timed(ops[a], ops[b], False)
        con = timed(ops[a], ops[b], True)
        print(f"{a}+{b}: serial {ser:.4f} ms, two streams {con:.4f} ms  (sum alone {t[a]+t[b]:.4f}, max alone {max(t[a], t[b]):.4f})")


if __name__ == "__main__":
    main()
