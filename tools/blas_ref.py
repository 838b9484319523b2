#!/usr/bin/env python
"""What the vendor BLAS (torch.mm -> hipBLASLt/rocBLAS) reaches on the path's GEMM shapes, fp32, for reference."""
import torch

def t(fn, flops, reps=30):
    for _ in range(40):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    return ms, flops / ms / 1e9

torch.backends.cuda.matmul.allow_tf32 = False
dev = "cuda:0"
for m, k, n in [(96000, 300, 300), (96000, 600, 300), (96000, 1200, 300), (62000, 300, 300), (98304, 304, 304), (8192, 8192, 8192)]:
    a = torch.randn(m, k, device=dev); b = torch.randn(k, n, device=dev); c = torch.empty(m, n, device=dev)
    ms, tf = t(lambda: torch.mm(a, b, out=c), 2.0 * m * k * n)
    print(f"NN  M={m:6d} K={k:5d} N={n:5d}: {ms:8.4f} ms {tf:7.2f} TF")
for kk, i, j in [(96000, 300, 300), (62000, 300, 300)]:
    g = torch.randn(kk, i, device=dev); x = torch.randn(kk, j, device=dev); c = torch.empty(i, j, device=dev)
    ms, tf = t(lambda: torch.mm(g.t(), x, out=c), 2.0 * kk * i * j)
    print(f"TN  K={kk:6d} I={i:5d} J={j:5d}: {ms:8.4f} ms {tf:7.2f} TF")

# BASELINE configs[4] (h = 768, bf16 storage): the cell's GEMM shapes on the vendor BLAS in bf16 (fp32 accumulation) -- the stated
# ceiling for the bf16 pipeline's dominant launches.  M = the node-compact row count of the bench batch.
print("-- bf16 (torch.mm on bf16 tensors -> hipBLASLt), configs[4] cell shapes")
for m, k, n in [(62000, 768, 768), (62000, 1536, 768), (96000, 768, 768), (96000, 1536, 768), (8192, 8192, 8192)]:
    a = torch.randn(m, k, device=dev).bfloat16(); b = torch.randn(k, n, device=dev).bfloat16()
    c = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    ms, tf = t(lambda: torch.mm(a, b, out=c), 2.0 * m * k * n)
    print(f"NN bf16 M={m:6d} K={k:5d} N={n:5d}: {ms:8.4f} ms {tf:7.2f} TF  ({tf / 2500:.3f} of 2.5 PF)")
    bt = b.t().contiguous()       # weights as stored [N][K] (the library's NT form)
    ms, tf = t(lambda: torch.mm(a, bt.t(), out=c), 2.0 * m * k * n)
    print(f"NT bf16 M={m:6d} K={k:5d} N={n:5d}: {ms:8.4f} ms {tf:7.2f} TF  ({tf / 2500:.3f} of 2.5 PF)")
for kk, i, j in [(62000, 768, 768), (96000, 768, 768)]:
    g = torch.randn(kk, i, device=dev).bfloat16(); x = torch.randn(kk, j, device=dev).bfloat16()
    c = torch.empty(i, j, device=dev, dtype=torch.bfloat16)
    ms, tf = t(lambda: torch.mm(g.t(), x, out=c), 2.0 * kk * i * j)
    print(f"TN bf16 K={kk:6d} I={i:5d} J={j:5d}: {ms:8.4f} ms {tf:7.2f} TF  ({tf / 2500:.3f} of 2.5 PF)")
