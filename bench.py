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


def build_workload(batch=32, n_evd=30, seed=20240229, device="cuda:0", cfg: SynthConfig = None, lr=1e-4, compact=None,
                   n_batches=1, evd_dist="fixed"):
    """Model (random init, reference init scheme), `n_batches` distinct seeded synthetic batches resident on `device`
    (seed, seed + 1000, ...; the step loop rotates through them so that no per-batch cost hides behind one reused
    batch), and `oracle_slice(k)`: CPU-oracle results of the first k claims of batch 0.
    evd_dist: "fixed" (cfg.n_evd per claim; <= 0 = U[1,30]) or "snopes" (empirical histogram, mean 6.9)."""
    from get_amd import modules
    from get_amd.batch import NativeBatch
    from get_amd.synth import snopes_evidence_counts
    cfg = cfg or SynthConfig(batch=batch, n_evd=n_evd)
    emb, art, clm = make_embeddings(cfg, seed)
    torch.manual_seed(seed)
    model = modules.Graph_basedSemantiStructure(cfg.model_params(emb, art, clm)).to(device)
    raws, batches = [], []
    for i in range(max(1, n_batches)):
        c = cfg
        if evd_dist == "snopes":
            counts = snopes_evidence_counts(np.random.default_rng(seed + 1000 * i + 17), cfg.batch)
            c = SynthConfig(**{**cfg.__dict__, "evd_counts": [int(x) for x in counts]})
        raw = make_raw_batch(c, seed + 1000 * i)
        raws.append(raw)
        batches.append(NativeBatch(raw["claim_tokens"], raw["claim_len"], raw["evd_tokens"], raw["evd_len"], raw["evd_counts"],
                                   raw["doc_sources"], raw["query_sources"], raw["labels"], window=cfg.window,
                                   n_max=cfg.fixed_num_evidences, device=device, compact=compact))
    raw, b0 = raws[0], batches[0]
    query, document, kargs = b0.inputs()
    nnz = float(torch.count_nonzero(kargs["docs_adj"].to_dense()).item()) / max(b0.b1, 1)

    def oracle_slice(k: int, return_aux=False):
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
        res = O.model_forward(p, sub_cfg.__dict__, T(inp["query"]), T(inp["document"]), T(inp["query_adj"]),
                              T(inp["doc_ids"]), T(inp["doc_adj"]), T(inp["query_lens"]), inp["evd_counts"],
                              T(inp["doc_sources"]), T(inp["query_sources"]), return_aux=return_aux)
        out = dict(phi=res[0].detach(), word_w=res[1].detach(), inp=inp, params=p, cfg=sub_cfg)
        if return_aux:
            out["keep"] = res[3]["keep"]
        return out

    return dict(cfg=cfg, model=model, raw=raw, batches=batches, oracle_slice=oracle_slice, nnz_per_graph=nnz,
                query=query, document=document, kargs=kargs, labels=b0.labels, make_inputs=b0.inputs,
                compact=b0.compact, b1=sum(b.b1 for b in batches) / len(batches), b1_each=[b.b1 for b in batches],
                m_real=sum(b.m_real for b in batches) / len(batches))


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


def _oracle_train_step_fn(wl, claims, train):
    """One oracle step on the first `claims` claims of batch 0: forward (+dropout when `train`), CE, backward and --
    in training mode -- one Adam(lr 1e-4, weight_decay 1e-3) update, as declare_fitter.py:58-61."""
    from oracle import get_oracle as O
    s = wl["oracle_slice"](claims)
    inp, cfg = s["inp"], s["cfg"]
    T = torch.from_numpy
    state = {"p": {k: v.clone().requires_grad_(v.is_floating_point() and k != "embedding.weight") for k, v in s["params"].items()},
             "adam": {}}
    pairs = int(inp["evd_counts"].sum())

    def one():
        p = state["p"]
        for v in p.values():
            v.grad = None
        phi, _, _ = O.model_forward(p, cfg.__dict__, T(inp["query"]), T(inp["document"]), T(inp["query_adj"]),
                                    T(inp["doc_ids"]), T(inp["doc_adj"]), T(inp["query_lens"]), inp["evd_counts"],
                                    T(inp["doc_sources"]), T(inp["query_sources"]), drop_p=0.2 if train else 0.0)
        O.cross_entropy(phi, T(inp["labels"])).backward()
        if train:
            with torch.no_grad():
                new = O.adam_step({k: v.detach() for k, v in p.items()}, {k: v.grad for k, v in p.items()}, state["adam"])
            state["p"] = {k: v.clone().requires_grad_(p[k].requires_grad) for k, v in new.items()}

    return one, pairs


def cpu_baseline(wl, threads=None, budget_s=60.0):
    """SURVEY.md 8(d) / BASELINE.md 3: the CPU oracle (a port of the reference's PyTorch path) on the FULL batch, training
    mode (dropout on) + Adam, 2 warm-up + 5 timed steps.  Threads: `threads`, i.e. the pool size the probe found fastest
    -- handing torch all os.cpu_count() = 256 hardware threads of the GPU box makes this workload 100x SLOWER (141 s per
    step measured, against ~1.5 s at 16 threads), and the default bench run has to finish within minutes; both numbers
    are stated in the output.  If one step exceeds budget_s / 7 the step counts are cut (and reported)."""
    cores = os.cpu_count() or 1
    use = int(threads) if threads else cores
    torch.set_num_threads(use)
    one, pairs = _oracle_train_step_fn(wl, wl["cfg"].batch, train=True)
    t0 = time.time()
    one()
    t1 = time.time() - t0
    warm, timed = (2, 5) if 7 * t1 <= budget_s else (1, max(1, int(budget_s / max(t1, 1e-3)) - 1))
    for _ in range(warm - 1):
        one()
    t0 = time.time()
    for _ in range(timed):
        one()
    dt = time.time() - t0
    return {"value": pairs * timed / dt, "unit": "pairs/s", "cores": use, "host_cpu_count": cores, "kind": "port",
            "s_per_step": dt / timed,
            "sample": f"oracle fwd+bwd+Adam, training mode (dropout 0.2), the full batch of {wl['cfg'].batch} claims = {pairs} pairs, "
                      f"{warm} warm-up + {timed} timed steps in {dt:.1f} s, torch CPU with {use} intra-op threads (fastest pool "
                      f"size of the probe) on a host with os.cpu_count() = {cores}"}


def cpu_baseline_probe(wl, budget_s=8.0, claims=2):
    """Second figure: the same oracle on a 2-claim sample in evaluation mode at the best torch pool size of a 8..128 probe
    (torch's intra-op pool thrashes when handed every hardware thread of a 2-socket host for matrices this small)."""
    cores = os.cpu_count() or 1
    one, pairs = _oracle_train_step_fn(wl, claims, train=False)
    torch.set_num_threads(min(8, cores))
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
        if d > 4.0:
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
    return {"value": pairs * n / dt, "unit": "pairs/s", "cores": best_threads, "host_cpu_count": cores, "kind": "port",
            "sample": f"oracle fwd+bwd (evaluation mode, no optimiser) on the first {claims} claims = {pairs} pairs, {n} repeats "
                      f"in {dt:.1f} s, best torch pool size of a 8..128 probe"}


def measure(args, wl, trainer, world, device, dist, steps, warmup, profile=True):
    """W warm-up + K timed steps (barrier + synchronize on both sides) of the rotating resident batches; returns
    (seconds, last loss, dominant-kernel profile row or None)."""
    from get_amd import _lib
    model = wl["model"]
    batches = wl["batches"]
    state = {"i": 0}

    def step():
        b = batches[state["i"] % len(batches)]
        state["i"] += 1
        if args.forward_only:
            with torch.no_grad():
                query, document, kargs = b.inputs()
                return model(query, document, **kargs).sum()
        trainer.zero_grad()
        query, document, kargs = b.inputs()
        phi = model(query, document, **kargs)
        loss = torch.nn.functional.cross_entropy(phi, b.labels)
        loss.backward()
        trainer.step()
        return loss

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    prof_dom = None
    # live roofline of the dominant kernel: HIP events around each of its launches in the FIRST `prof_steps` of the timed
    # steps (an event pair between two kernels costs ~10 us of dispatch overlap: on every launch of every timed step that
    # was 1.7 % of the reported throughput; on a quarter of the steps it is 0.4 %)
    prof_steps = min(steps, max(1, PROFILE_TIMED_STEPS)) if profile else 0
    if profile:
        _lib.profile_enable(True, only=[DOMINANT])
        _lib.profile_collect()
    barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        if profile and i == prof_steps:
            _lib.profile_enable(False)       # host-side switch, no device work
        loss = step()
    barrier()
    dt = time.perf_counter() - t0
    if profile:
        prof_dom = _lib.profile_collect()[DOMINANT]
        prof_dom["steps"] = prof_steps
        _lib.profile_enable(False)
    return dt, loss, prof_dom, step


def phase_split(wl, trainer, steps=5):
    """fwd / bwd / all-reduce / optimiser milliseconds per step: HIP events on the compute stream between the phases of
    `steps` extra, untimed steps (the events serialise the overlapped early all-reduce, so this is a breakdown, not the
    step time)."""
    from get_amd import ops
    model, batches = wl["model"], wl["batches"]
    ev = lambda: torch.cuda.Event(enable_timing=True)
    acc = np.zeros(5)
    for i in range(steps):
        b = batches[i % len(batches)]
        e = [ev() for _ in range(6)]
        trainer.zero_grad()
        e[0].record()
        query, document, kargs = b.inputs()
        e[1].record()
        loss = torch.nn.functional.cross_entropy(model(query, document, **kargs), b.labels)
        e[2].record()
        loss.backward()
        e[3].record()
        trainer.allreduce()
        e[4].record()
        trainer.t += 1
        ops.adam_step_flat(trainer.flat_p, trainer.flat_g, trainer.flat_m, trainer.flat_v, trainer.t, lr=trainer.lr,
                           betas=trainer.betas, eps=trainer.eps, weight_decay=trainer.weight_decay,
                           grad_scale=1.0 / trainer.world)
        ops.refresh_transposes(trainer._matrices)
        e[5].record()
        torch.cuda.synchronize()
        acc += np.array([e[j].elapsed_time(e[j + 1]) for j in range(5)])
    acc /= steps
    return {"graph_build_ms": acc[0], "forward_ms": acc[1], "backward_ms": acc[2], "allreduce_ms": acc[3],
            "optimizer_ms": acc[4], "note": f"HIP events between the phases of {steps} extra untimed steps"}


def parity_check(wl, k=4):
    """Evaluation-mode logits and GSL keep-sets of the first k claims of batch 0 against the CPU oracle."""
    model = wl["model"]
    was = model.training
    model.train(False)
    try:
        with torch.no_grad():
            q, d_, k_ = wl["batches"][0].inputs()
            phi_gpu = model(q, d_, **k_)[:k].cpu()
            keep_words = model.ggnn_with_gsl.last_keep.cpu().numpy().astype(np.uint64)      # (B1, W) bit words
        par = wl["oracle_slice"](k, return_aux=True)
        n_pairs = par["keep"].shape[0]
        r = par["keep"].shape[1]
        bits = ((keep_words[:n_pairs, :, None] >> np.arange(64, dtype=np.uint64)[None, None, :]) & np.uint64(1)).astype(bool)
        keep_gpu = bits.reshape(n_pairs, -1)[:, :r]
        real = par["inp"]["doc_ids"][:n_pairs] > 0
        mism = int(((keep_gpu != par["keep"].numpy()) & real).any(axis=1).sum())
        return {"max_abs_logit_diff_vs_cpu_oracle": float((phi_gpu - par["phi"]).abs().max()), "claims_checked": k,
                "graphs_checked": n_pairs, "graphs_with_real_node_keep_set_mismatch": mism}
    finally:
        model.train(was)


DOMINANT = "gemm_big"
PROFILE_TIMED_STEPS = 5           # timed steps whose dominant-kernel launches carry HIP events


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="claims per GPU (weak scaling: fixed per-GPU work)")
    ap.add_argument("--global-batch", type=int, default=0,
                    help="fixed GLOBAL claim count, split evenly over the ranks (SURVEY 8(e): 256 -> 256/N per GPU; strong scaling)")
    ap.add_argument("--n-evd", type=int, default=30, help="evidences per claim (<=0: ragged U[1,30])")
    ap.add_argument("--evd-dist", choices=["fixed", "snopes"], default="fixed",
                    help="snopes: evidence counts from the empirical Snopes histogram (mean 6.9, the realistic series)")
    ap.add_argument("--len-right", type=int, default=100, help="evidence length R (configs[2]: 200)")
    ap.add_argument("--hidden", type=int, default=300, help="hidden size H (configs[4]: 768)")
    ap.add_argument("--word-heads", type=int, default=5)
    ap.add_argument("--window", type=int, default=3, help="gnn_window")
    ap.add_argument("--gsl-rate", type=float, default=0.6)
    ap.add_argument("--batches", type=int, default=4, help="distinct resident batches the step loop rotates through")
    ap.add_argument("--gemm-mode", choices=["fp32", "bf16"], default="fp32",
                    help="bf16: opt-in bf16 MFMA path (configs[4]); the headline metric is fp32")
    ap.add_argument("--eval-mode", action="store_true", help="disable dropout (parity mode)")
    ap.add_argument("--forward-only", action="store_true",
                    help="auxiliary serving measurement: evaluation-mode forward only (not the headline metric)")
    ap.add_argument("--padded", action="store_true",
                    help="run every layer on all R padded node rows (the reference's layout) instead of the node-compact one")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="skip the per-kernel HIP-event timing")
    ap.add_argument("--no-series", action="store_true", help="skip the realistic evidence-count series")
    ap.add_argument("--cpu-threads", type=int, default=0, help="torch threads of the CPU baseline (default: the probe's fastest pool size)")
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
    from get_amd.dist import FlatTrainer, shard_claims
    _lib.load()
    _lib.set_gemm_mode(args.gemm_mode)

    per_rank = args.batch
    if args.global_batch > 0:
        per_rank = len(shard_claims(args.global_batch, rank, world))
    cfg_in = SynthConfig(batch=per_rank, n_evd=args.n_evd, len_right=args.len_right, hidden=args.hidden, emb_dim=args.hidden,
                         word_heads=args.word_heads, window=args.window, gsl_rate=args.gsl_rate)
    wl = build_workload(seed=20240229 + rank, device=device, cfg=cfg_in, compact=False if args.padded else None,
                        n_batches=args.batches, evd_dist=args.evd_dist)
    model, cfg = wl["model"], wl["cfg"]
    if world > 1:      # identical replicas: broadcast rank 0's parameters
        for p in model.parameters():
            dist.broadcast(p.data, src=0)
    trainer = FlatTrainer(model, lr=1e-4, weight_decay=1e-3)
    from get_amd import ops
    ops.bump_weight_epoch()            # parameters were (re)written through .data: drop cached transposes
    if world > 1:
        trainer.attach_overlap()     # 76 % of the gradient all-reduce runs underneath the first cell's backward
    model.train(not (args.eval_mode or args.forward_only))

    dt, loss, prof_dom, step = measure(args, wl, trainer, world, device, dist, args.steps, args.warmup,
                                       profile=not args.no_profile)
    prof = None
    PROFILE_EXTRA_STEPS = 5
    if not args.no_profile:
        _lib.profile_enable(True)                       # every instrumented kernel, outside the timed region
        _lib.profile_collect()
        for _ in range(PROFILE_EXTRA_STEPS):
            step()
        torch.cuda.synchronize()
        prof = _lib.profile_collect()
        _lib.profile_enable(False)
    split = None if args.forward_only else phase_split(wl, trainer)      # collective inside: every rank runs it
    tmax = torch.tensor([dt], device=device, dtype=torch.float64)
    pairs_done = sum(wl["b1_each"][i % len(wl["b1_each"])] for i in range(args.warmup, args.warmup + args.steps))
    cnt = torch.tensor([float(pairs_done), float(cfg.batch * args.steps)], device=device, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
    dt = float(tmax.item())
    total_pairs, total_claims = float(cnt[0].item()), float(cnt[1].item())

    if rank == 0:
        ms_per_step = 1e3 * dt / args.steps
        value = total_pairs / dt
        headline = (args.len_right, args.hidden, args.word_heads, args.window, args.gsl_rate, args.n_evd, args.evd_dist) == \
                   (100, 300, 5, 3, 0.6, 30, "fixed") and per_rank == 32 and args.gemm_mode == "fp32" and not args.padded \
                   and not args.eval_mode and not args.forward_only
        real_nodes = wl["m_real"] / max(wl["b1"], 1)
        fl = flops_per_pair(cfg, wl["nnz_per_graph"], real_nodes if wl["compact"] else None)
        fl_run = fl.get("executed", fl["fwd_bwd"])
        evd_txt = "Snopes-histogram (mean 6.9)" if args.evd_dist == "snopes" else (str(args.n_evd) if args.n_evd > 0 else "U[1,30]")
        out = {
            "metric": (f"claim-evidence pairs/sec fwd+bwd (B={cfg.batch}, h={cfg.hidden})" if not args.forward_only
                       else "claim-evidence pairs/sec forward only, evaluation mode (auxiliary)"),
            "value": value, "unit": "pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong" if args.global_batch > 0 else "weak", "vs_baseline": None,
            "dtype": "f32" if args.gemm_mode == "fp32" else "bf16 operands / f32 accumulate in the big GEMMs, f32 elsewhere",
            "data": "synthetic",
            "claims_per_s": total_claims / dt,
            "config": {"workload": ("BASELINE configs[1]: Snopes-shaped synthetic batch, " if headline
                                    else "non-headline shape (see flags): synthetic batch, ") +
                                   f"B={cfg.batch} claims x {evd_txt} evidences per GPU "
                                   f"(B1={wl['b1']:.0f} pairs), L_left={cfg.len_left}, L_right={cfg.len_right}, D=H={cfg.hidden}, "
                                   f"{cfg.word_heads} word heads / {cfg.evd_heads} evidence heads, gnn_window={cfg.window}, "
                                   f"gsl_rate={cfg.gsl_rate}",
                       "step": "device graph build + forward + CE loss + backward + flat grad all-reduce + fused Adam",
                       "batches": f"{len(wl['batches'])} distinct resident batches, rotated every step",
                       "mode": "eval (dropout off)" if args.eval_mode else "train (dropout on)",
                       "layout": (f"node-compact: {wl['m_real']:.0f} real-node rows of {wl['b1'] * cfg.len_right:.0f} padded rows "
                                  f"({real_nodes:.1f} unique tokens per {cfg.len_right}-token evidence); padding nodes only "
                                  "in the first cell's forward and the scorer") if wl["compact"]
                                 else "padded: every layer on all R node rows (reference layout)",
                       "parallelism": f"dp{world}", "pairs_per_gpu": wl["b1"],
                       "global_batch_claims": cfg.batch * world if args.global_batch <= 0 else args.global_batch,
                       "loss": float(loss.item())},
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
                    pk = PEAK_BF16_MFMA_TFLOPS if (args.gemm_mode == "bf16" and name in ("gemm_big", "gemm_big_tn")) else PEAK_F32_MFMA_TFLOPS
                    entry.update(bound="mfma", achieved_tflops=rate / 1e12, frac=rate / 1e12 / pk)
                elif name == "few_row_streams":
                    # claim-side aggregation / gate kernels and the evidence-level attention kernels: a few MB per launch, bound
                    # by launch latency, kept apart so that they do not dilute the activation-sized launches' rates
                    entry.update(bound="latency", achieved_gbps=rate / 1e9, frac=rate / 1e9 / PEAK_HBM_GBPS)
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
                 "avg_launch_ms": prof_dom["ms"] / prof_dom["launches"], "launches_per_step": prof_dom["launches"] / prof_dom["steps"]}
            # HBM bytes per launch from the committed rocprofv3 PMC passes -- only for the invocation they were measured
            # on (the headline shape); any other shape reports null rather than a number that belongs to another run
            traffic = None
            if headline and world == 1:
                try:
                    pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))[dom]
                    traffic = (pm["fetch_kib"] * pm["fetch_correction"] + pm["write_kib"]) * 1024.0
                except Exception:
                    traffic = None
            out["roofline"] = {"kernel": dom, "bound": "mfma", "achieved": d["achieved_tflops"],
                               "peak": dom_peak, "unit": "TFLOP/s", "frac": d["frac"], "traffic": traffic,
                               "avg_launch_ms": d["avg_launch_ms"], "launches_per_step": d["launches_per_step"],
                               "alg_flops_per_launch": prof_dom["work"] / prof_dom["launches"],
                               "measured": f"HIP events around every {dom} launch of the first {prof_dom['steps']} of the {args.steps} timed steps"}
            if dom_note:
                out["roofline"]["note"] = dom_note
            out["kernels_note"] = (f"per-kernel table from {PROFILE_EXTRA_STEPS} extra untimed steps with every library "
                                   "kernel bracketed by HIP events (costs ~0.4 ms/step, so it stays out of the timed region)")
            out["kernels"] = kernels
        if split is not None:
            out["step_split_ms"] = split
        if not args.no_profile and world == 1:
            out["box_reference"] = box_reference(device)
        if world == 1 and not args.no_series and not args.forward_only and args.evd_dist == "fixed" and headline:
            # SURVEY 8(d) "realistic series": evidence counts from the empirical Snopes histogram (mean 6.9 per claim);
            # at B = 32 that is ~220 pairs per step, at B = 139 about the headline's 960
            series = []
            for bsz in (32, 139):
                c2 = SynthConfig(**{**cfg.__dict__, "batch": bsz})
                w2 = build_workload(seed=20240229, device=device, cfg=c2, n_batches=args.batches, evd_dist="snopes")
                w2["model"].train(True)
                t2 = FlatTrainer(w2["model"], lr=1e-4, weight_decay=1e-3)
                ops.bump_weight_epoch()
                S_W, S_K = 8, 30        # (3 + 10 steps were too few for a fresh model: allocator / workspace warm-up leaked in)
                dt2, _, _, _ = measure(args, w2, t2, 1, device, dist, S_K, S_W, profile=False)
                done = sum(w2["b1_each"][i % len(w2["b1_each"])] for i in range(S_W, S_W + S_K))
                series.append({"claims": bsz, "pairs_per_step": w2["b1"], "pairs_per_s": done / dt2,
                               "claims_per_s": bsz * S_K / dt2, "ms_per_step": 1e3 * dt2 / S_K})
                del w2, t2
            ops.bump_weight_epoch()
            out["realistic_series"] = {"evidence_counts": "empirical Snopes histogram (get_amd.synth.SNOPES_EVD_HIST, mean 6.9, max 26)",
                                       "rows": series}
        if world == 1 and not args.no_cpu_baseline and not args.forward_only:
            out["parity"] = parity_check(wl)
            probe = cpu_baseline_probe(wl)
            cb = cpu_baseline(wl, threads=args.cpu_threads or probe["cores"])
            cb["speedup_gpu_over_cpu"] = value / cb["value"]
            out["cpu_baseline"] = cb
            out["cpu_baseline_probe"] = probe
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
