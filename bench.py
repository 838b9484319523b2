#!/usr/bin/env python
"""GET hot-path benchmark: claim-evidence pairs/s, forward + backward (+ gradient all-reduce + Adam),
on synthetic Snopes-shaped batches (BASELINE.json configs[1]: B=32 claims x 30 evidences, L=30, R=100,
D=H=300, 5 word heads / 2 evidence heads, window 3, gsl_rate 0.6, fp32).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = graph build from token ids + full model forward + cross-entropy + backward + one flat
gradient all-reduce (N>1) + fused Adam, with the batch already resident in HBM.  Rank 0 prints ONE
JSON line.  Weak scaling: every rank runs its own B=32 batch (claims are independent; the only
exchange is the gradient all-reduce).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from get_amd.synth import SynthConfig, make_embeddings, make_raw_batch  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense, spec
PEAK_BF16_MFMA_TFLOPS = 2500.0   # same guide: dense bf16 MFMA (the opt-in --gemm-mode bf16 prices gemm_big against this)
PEAK_HBM_GBPS = 8000.0           # HBM3E spec (6.3 TB/s achievable)


def flops_per_pair(cfg: SynthConfig, nnz_per_graph: float, real_nodes: float = None) -> dict:
    """Minimal-formulation FLOPs per claim-evidence pair (SURVEY.md 8(d) formulas as functions).
    `fwd_bwd`: every layer on all R padded node rows (the reference's shape of the work).
    `executed`: what the node-compact layout actually runs -- the first cell's forward and the scorer on R rows,
    everything else on the `real_nodes` (mean unique tokens per evidence) rows that can influence a result."""
    R, D, H, hw = cfg.len_right, cfg.emb_dim, cfg.hidden, cfg.word_heads

    def cell(rows, din, dout, agg):
        return 2 * rows * din * dout + 12 * rows * dout * dout + agg * dout

    def att(rows):
        return 2 * H * H + 2 * rows * H * H + 2 * rows * H * hw + 2 * rows * H * hw

    agg = 2 * nnz_per_graph
    fwd = cell(R, D, H, agg) + cell(R, H, 1, agg) + cell(R, H, H, agg) + att(R)
    out = {"fwd": fwd, "fwd_bwd": 3 * fwd - 2 * R * D * H}
    if real_nodes is not None:
        n = real_nodes
        fwd_x = cell(R, D, H, agg) + cell(R, H, 1, agg) + cell(n, H, H, agg) + att(n)
        bwd_x = 2 * (cell(n, D, H, agg) + cell(n, H, H, agg) + att(n)) - 2 * n * D * H
        out["executed"] = fwd_x + bwd_x
    return out


def build_workload(batch=32, n_evd=30, seed=20240229, device="cuda:0", cfg: SynthConfig = None, lr=1e-4, compact=None):
    """Model (random init, reference init scheme), a seeded synthetic batch resident on `device`, and
    the native-path kargs.  Also returns `oracle_slice(k)`: CPU-oracle logits of the first k claims."""
    from get_amd import modules, ops
    cfg = cfg or SynthConfig(batch=batch, n_evd=n_evd)
    emb, art, clm = make_embeddings(cfg, seed)
    torch.manual_seed(seed)
    model = modules.Graph_basedSemantiStructure(cfg.model_params(emb, art, clm)).to(device)
    raw = make_raw_batch(cfg, seed)
    from get_amd.batch import NativeBatch
    batch_obj = NativeBatch(raw["claim_tokens"], raw["claim_len"], raw["evd_tokens"], raw["evd_len"], raw["evd_counts"],
                            raw["doc_sources"], raw["query_sources"], raw["labels"], window=cfg.window,
                            n_max=cfg.fixed_num_evidences, device=device, compact=compact)
    b1 = batch_obj.b1
    labels = batch_obj.labels
    # per-step device work that replaces the reference's host graph construction + H2D of dense float64
    # adjacency: token ids -> packed graphs (interactions.py:334-351)
    make_inputs = batch_obj.inputs
    query, document, kargs = make_inputs()
    nnz = float(torch.count_nonzero(kargs["docs_adj"].to_dense()).item()) / max(b1, 1)

    def oracle_slice(k: int):
        from oracle import get_oracle as O
        from oracle.assemble import assemble_inputs
        sub_cfg = SynthConfig(**{**cfg.__dict__, "batch": k, "evd_counts": [int(c) for c in raw["evd_counts"][:k]]})
        nb1 = int(raw["evd_counts"][:k].sum())
        sub = dict(claim_tokens=raw["claim_tokens"][:k], claim_len=raw["claim_len"][:k],
                   evd_tokens=raw["evd_tokens"][:nb1], evd_len=raw["evd_len"][:nb1],
                   evd_counts=raw["evd_counts"][:k], doc_sources=raw["doc_sources"][:k],
                   query_sources=raw["query_sources"][:k], labels=raw["labels"][:k])
        inp = assemble_inputs(sub, sub_cfg, O.convert_text)
        p = {kk: v.detach().cpu().clone() for kk, v in model.state_dict().items()}
        T = torch.from_numpy
        phi, ww, ew = O.model_forward(p, sub_cfg.__dict__, T(inp["query"]), T(inp["document"]), T(inp["query_adj"]),
                                      T(inp["doc_ids"]), T(inp["doc_adj"]), T(inp["query_lens"]), inp["evd_counts"],
                                      T(inp["doc_sources"]), T(inp["query_sources"]))
        return dict(phi=phi.detach(), word_w=ww.detach(), inp=inp, params=p, cfg=sub_cfg)

    return dict(cfg=cfg, model=model, raw=raw, query=query, document=document, kargs=kargs, labels=labels, b1=b1,
                make_inputs=make_inputs, oracle_slice=oracle_slice, nnz_per_graph=nnz, compact=batch_obj.compact,
                m_real=batch_obj.m_real)


def box_reference(device):
    """Measured ceilings of THIS box, next to the datasheet peaks: a plain device copy (HBM read +
    write bytes per second) and the vendor BLAS fp32 GEMM rate at 8192^3 (torch.mm -> hipBLASLt/rocBLAS)."""
    x = torch.empty(256 << 20, device=device, dtype=torch.float32)      # 1 GiB
    y = torch.empty_like(x)
    a = torch.randn(8192, 8192, device=device)
    b = torch.randn(8192, 8192, device=device)
    c = torch.empty(8192, 8192, device=device)

    def timed(fn, reps):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e-3

    t_copy = timed(lambda: y.copy_(x), 10)
    t_mm = timed(lambda: torch.mm(a, b, out=c), 5)
    return {"copy_gbps": 2.0 * x.numel() * 4 / t_copy / 1e9, "blas_sgemm_8192_tflops": 2.0 * 8192 ** 3 / t_mm / 1e12}


def cpu_baseline(wl, budget_s=15.0, claims=2):
    """The CPU oracle (a port of the reference's PyTorch path) timed on this host's cores on a bounded
    sample: forward + backward of the first `claims` claims of the same batch, repeated for ~budget_s."""
    from oracle import get_oracle as O
    cores = os.cpu_count() or 1
    s = wl["oracle_slice"](claims)
    inp, cfg = s["inp"], s["cfg"]
    T = torch.from_numpy
    p = {k: v.clone().requires_grad_(v.is_floating_point() and k != "embedding.weight") for k, v in s["params"].items()}
    pairs = int(inp["evd_counts"].sum())

    def one():
        for v in p.values():
            v.grad = None
        phi, _, _ = O.model_forward(p, cfg.__dict__, T(inp["query"]), T(inp["document"]), T(inp["query_adj"]),
                                    T(inp["doc_ids"]), T(inp["doc_adj"]), T(inp["query_lens"]), inp["evd_counts"],
                                    T(inp["doc_sources"]), T(inp["query_sources"]))
        O.cross_entropy(phi, T(inp["labels"])).backward()

    torch.set_num_threads(min(8, cores))
    # torch's intra-op pool thrashes when handed every hardware thread of a 2-socket host for
    # matrices this small: probe a few pool sizes once, then spend the budget at the best one
    one()
    best_threads, best_dt = None, None
    for th in [t for t in (8, 16, 32, 64, 128) if t <= cores] or [cores]:
        torch.set_num_threads(th)
        one()
        t0 = time.time()
        one()
        d = time.time() - t0
        if best_dt is None or d < best_dt:
            best_threads, best_dt = th, d
        if d > 8.0:
            break
    torch.set_num_threads(best_threads)
    t0 = time.time()
    n = 0
    while True:
        one()
        n += 1
        if time.time() - t0 >= budget_s or n >= 200:
            break
    dt = time.time() - t0
    return {"value": pairs * n / dt, "unit": "pairs/s", "cores": best_threads, "kind": "port",
            "sample": f"oracle fwd+bwd (eval mode) on the first {claims} claims = {pairs} pairs of the same batch, "
                      f"{n} repeats in {dt:.1f} s, torch CPU with {best_threads} of {cores} host threads "
                      f"(best of a 8..128 probe)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="claims per GPU")
    ap.add_argument("--n-evd", type=int, default=30, help="evidences per claim (<=0: ragged U[1,30])")
    ap.add_argument("--len-right", type=int, default=100, help="evidence length R (configs[2]: 200)")
    ap.add_argument("--hidden", type=int, default=300, help="hidden size H (configs[4]: 768, run in fp32)")
    ap.add_argument("--word-heads", type=int, default=5)
    ap.add_argument("--window", type=int, default=3, help="gnn_window")
    ap.add_argument("--gsl-rate", type=float, default=0.6)
    ap.add_argument("--gemm-mode", choices=["fp32", "bf16"], default="fp32",
                    help="bf16: opt-in bf16-operand MFMA in the big NT/NN GEMMs (configs[4]); the headline metric is fp32")
    ap.add_argument("--eval-mode", action="store_true", help="disable dropout (parity mode)")
    ap.add_argument("--forward-only", action="store_true",
                    help="auxiliary serving measurement: evaluation-mode forward only (not the headline metric)")
    ap.add_argument("--padded", action="store_true",
                    help="run every layer on all R padded node rows (the reference's layout) instead of the node-compact one")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="skip the per-kernel HIP-event timing")
    ap.add_argument("--cpu-budget", type=float, default=15.0)
    args = ap.parse_args()

    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # GET_AMD_BENCH_BACKEND=gloo: rehearsal of the N>1 code path on a box with fewer GPUs than ranks (ranks share
        # devices, the all-reduce goes through the host).  The real run is one rank per GPU over RCCL ("nccl").
        backend = os.environ.get("GET_AMD_BENCH_BACKEND", "nccl")
        local_rank = local_rank % torch.cuda.device_count() if backend != "nccl" else local_rank
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    else:
        torch.cuda.set_device(0)
    assert world == args.gpus or world == 1, f"launched with WORLD_SIZE={world} but --gpus {args.gpus}"
    device = f"cuda:{local_rank if world > 1 else 0}"

    from get_amd import _lib
    from get_amd.dist import FlatTrainer
    _lib.load()
    _lib.set_gemm_mode(args.gemm_mode)

    cfg_in = SynthConfig(batch=args.batch, n_evd=args.n_evd, len_right=args.len_right, hidden=args.hidden,
                         word_heads=args.word_heads, window=args.window, gsl_rate=args.gsl_rate)
    wl = build_workload(seed=20240229 + rank, device=device, cfg=cfg_in, compact=False if args.padded else None)
    model, cfg = wl["model"], wl["cfg"]
    if world > 1:      # identical replicas: broadcast rank 0's parameters
        for p in model.parameters():
            dist.broadcast(p.data, src=0)
    trainer = FlatTrainer(model, lr=1e-4, weight_decay=1e-3)
    if world > 1:
        trainer.attach_overlap()     # 76 % of the gradient all-reduce runs underneath the first cell's backward
    model.train(not (args.eval_mode or args.forward_only))

    def step():
        if args.forward_only:
            with torch.no_grad():
                query, document, kargs = wl["make_inputs"]()
                return model(query, document, **kargs).sum()
        trainer.zero_grad()
        query, document, kargs = wl["make_inputs"]()
        phi = model(query, document, **kargs)
        loss = torch.nn.functional.cross_entropy(phi, wl["labels"])
        loss.backward()
        trainer.step()
        return loss

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    # Live roofline: inside the timed region only the dominant kernel (the 64x320 NT/NN GEMM, row "gemm_big") is
    # bracketed with HIP events -- instrumenting all ~100 launches of a step costs ~0.4 ms of queue bubbles per step.
    # The per-kernel table (`kernels`) comes from PROFILE_EXTRA_STEPS extra, untimed steps after the timed region.
    DOMINANT = "gemm_big"
    if not args.no_profile:
        _lib.profile_enable(True, only=[DOMINANT])
        _lib.profile_collect()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    barrier()
    dt = time.perf_counter() - t0
    prof = prof_dom = None
    PROFILE_EXTRA_STEPS = 5
    if not args.no_profile:
        prof_dom = _lib.profile_collect()[DOMINANT]
        _lib.profile_enable(True)                       # every instrumented kernel, outside the timed region
        for _ in range(PROFILE_EXTRA_STEPS):
            step()
        torch.cuda.synchronize()
        prof = _lib.profile_collect()
        _lib.profile_enable(False)
    tmax = torch.tensor([dt], device=device, dtype=torch.float64)
    pairs = torch.tensor([wl["b1"]], device=device, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(pairs, op=dist.ReduceOp.SUM)
    dt = float(tmax.item())
    total_pairs = float(pairs.item())

    if rank == 0:
        ms_per_step = 1e3 * dt / args.steps
        value = total_pairs * args.steps / dt
        real_nodes = wl["m_real"] / max(wl["b1"], 1)
        fl = flops_per_pair(cfg, wl["nnz_per_graph"], real_nodes if wl["compact"] else None)
        fl_run = fl.get("executed", fl["fwd_bwd"])
        out = {
            "metric": "claim-evidence pairs/sec fwd+bwd (B=32, h=300)" if not args.forward_only
                      else "claim-evidence pairs/sec forward only, evaluation mode (auxiliary)", "value": value, "unit": "pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.gemm_mode == "fp32" else "bf16 operands / f32 accumulate in the big NT GEMMs, f32 elsewhere",
            "data": "synthetic",
            "config": {"workload": ("BASELINE configs[1]: Snopes-shaped synthetic batch, " if (args.len_right, args.hidden, args.word_heads, args.window, args.gsl_rate) == (100, 300, 5, 3, 0.6)
                                    else "non-default shape (see flags): synthetic batch, ") +
                                   f"B={cfg.batch} claims x {args.n_evd if args.n_evd > 0 else 'U[1,30]'} evidences per GPU "
                                   f"(B1={wl['b1']} pairs), L_left={cfg.len_left}, L_right={cfg.len_right}, D=H={cfg.hidden}, "
                                   f"{cfg.word_heads} word heads / {cfg.evd_heads} evidence heads, gnn_window={cfg.window}, "
                                   f"gsl_rate={cfg.gsl_rate}",
                       "step": "device graph build + forward + CE loss + backward + flat grad all-reduce + fused Adam",
                       "mode": "eval (dropout off)" if args.eval_mode else "train (dropout on)",
                       "layout": (f"node-compact: {wl['m_real']} real-node rows of {wl['b1'] * cfg.len_right} padded rows "
                                  f"({real_nodes:.1f} unique tokens per {cfg.len_right}-token evidence); padding nodes only "
                                  "in the first cell's forward and the scorer") if wl["compact"]
                                 else "padded: every layer on all R node rows (reference layout)",
                       "parallelism": f"dp{world}", "pairs_per_gpu": wl["b1"], "loss": float(loss.item())},
            "path_tflops": {"flops_per_pair_executed": fl_run, "flops_per_pair_padded_form": fl["fwd_bwd"],
                            "achieved_tflops_per_gpu": fl_run * value / world / 1e12,
                            "frac_of_f32_mfma_peak": fl_run * value / world / 1e12 / PEAK_F32_MFMA_TFLOPS},
        }
        if prof is not None:
            kernels = {}
            for name, r in prof.items():
                if r["launches"] == 0:
                    continue
                per_step = r["ms"] / PROFILE_EXTRA_STEPS
                rate = r["work"] / (r["ms"] * 1e-3) if r["ms"] > 0 else 0.0
                entry = {"ms_per_step": per_step, "launches_per_step": r["launches"] / PROFILE_EXTRA_STEPS,
                         "avg_launch_ms": r["ms"] / r["launches"]}
                if name.startswith("gemm"):
                    pk = PEAK_BF16_MFMA_TFLOPS if (args.gemm_mode == "bf16" and name == "gemm_big") else PEAK_F32_MFMA_TFLOPS
                    entry.update(bound="mfma", achieved_tflops=rate / 1e12, frac=rate / 1e12 / pk)
                else:
                    entry.update(bound="hbm", achieved_gbps=rate / 1e9, frac=rate / 1e9 / PEAK_HBM_GBPS)
                kernels[name] = entry
            dom = max((k for k in kernels if k.startswith("gemm")), key=lambda k: kernels[k]["ms_per_step"])
            dom_note = None
            if dom != DOMINANT:      # unusual shapes: keep the live measurement of the instrumented kernel, say which one leads
                dom_note = f"{dom} takes more time per step than {DOMINANT} at this shape; the live roofline below is {DOMINANT}'s"
                dom = DOMINANT
            rate = prof_dom["work"] / (prof_dom["ms"] * 1e-3)
            dom_peak = PEAK_BF16_MFMA_TFLOPS if args.gemm_mode == "bf16" else PEAK_F32_MFMA_TFLOPS
            d = {"achieved_tflops": rate / 1e12, "frac": rate / 1e12 / dom_peak,
                 "avg_launch_ms": prof_dom["ms"] / prof_dom["launches"], "launches_per_step": prof_dom["launches"] / args.steps}
            traffic = None      # HBM bytes per launch from the committed rocprofv3 PMC passes of this same command
            try:
                pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))[dom]
                traffic = (pm["fetch_kib"] * pm["fetch_correction"] + pm["write_kib"]) * 1024.0
            except Exception:
                pass
            out["roofline"] = {"kernel": dom, "bound": "mfma", "achieved": d["achieved_tflops"],
                               "peak": dom_peak, "unit": "TFLOP/s", "frac": d["frac"], "traffic": traffic,
                               "avg_launch_ms": d["avg_launch_ms"], "launches_per_step": d["launches_per_step"],
                               "alg_flops_per_launch": prof_dom["work"] / prof_dom["launches"],
                               "measured": f"HIP events around every {dom} launch of the {args.steps} timed steps"}
            if dom_note:
                out["roofline"]["note"] = dom_note
            out["kernels_note"] = (f"per-kernel table from {PROFILE_EXTRA_STEPS} extra untimed steps with every library "
                                   "kernel bracketed by HIP events (costs ~0.4 ms/step, so it stays out of the timed region)")
            out["kernels"] = kernels
        if not args.no_profile:
            out["box_reference"] = box_reference(device)
        if world == 1 and not args.no_cpu_baseline:
            model.train(False)
            with torch.no_grad():
                q, d_, k_ = wl["make_inputs"]()
                phi_gpu = model(q, d_, **k_)[:4].cpu()
            par = wl["oracle_slice"](4)
            out["parity"] = {"max_abs_logit_diff_vs_cpu_oracle_first_4_claims": float((phi_gpu - par["phi"]).abs().max())}
            out["cpu_baseline"] = cpu_baseline(wl, budget_s=args.cpu_budget)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
