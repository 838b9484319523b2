#!/usr/bin/env python
"""Weight-gradient GEMM of the bf16 storage pipeline on its own (gh_linear_wgrad_bf16): dw += g16^T x16 with and without the fused bias
gradient, at the configs[4] cell shapes.  Shapes whose widths are multiples of 256 over >= 16 384 rows run on gemm_tn_pp_kernel."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from get_amd import ops  # noqa: E402

dev = "cuda:0"


def timeit(fn, reps=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best


for m, n, k in ((62208, 768, 768), (62208, 1536, 768), (62208, 2304, 768), (62208, 768, 1536), (12000, 768, 768), (62208, 320, 768)):
    g = torch.randn(m, n, device=dev).to(torch.bfloat16)
    x = torch.randn(m, k, device=dev).to(torch.bfloat16)
    dw = torch.zeros(n, k, device=dev)
    db = torch.zeros(n, device=dev)
    t0 = timeit(lambda: ops.linear_wgrad_bf16(g, x, dw))
    t1 = timeit(lambda: ops.linear_wgrad_bf16(g, x, dw, db))
    fl = 2.0 * m * n * k
    print(f"m={m} n={n} k={k}: dw {t0*1e3:7.1f} us ({fl/t0/1e9:6.0f} TF)   dw + db {t1*1e3:7.1f} us ({fl/t1/1e9:6.0f} TF)   [kernel + reduce_partials]")
