"""The N>1 data-parallel path with the REAL model on a GPU: 2 processes sharing cuda:0 (gloo stands in for RCCL on a
1-GPU box), claims sharded by `shard_claims`, gradients in the flat bucket, the early part of the all-reduce started
from inside backward (`attach_overlap`).  The averaged bucket must equal the single-process gradient of the union
batch (1e-5 relative), and one fused Adam step must leave both replicas bit-identical."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make(cfg, seed, dev, claims=None, compact=True):
    from get_amd import modules
    from get_amd.batch import NativeBatch
    from get_amd.synth import make_embeddings, make_raw_batch, make_state_dict
    emb, art, clm = make_embeddings(cfg, seed)
    model = modules.Graph_basedSemantiStructure(cfg.model_params(emb, art, clm))
    full = model.state_dict()
    full.update({k: torch.from_numpy(v) for k, v in make_state_dict(cfg, seed).items()})
    model.load_state_dict(full, strict=True)
    model = model.to(dev).train(False)          # evaluation mode: no dropout, so every rank/union run is deterministic
    raw = make_raw_batch(cfg, seed)
    counts = raw["evd_counts"]
    offs = np.concatenate([[0], np.cumsum(counts)])
    if claims is None:
        claims = range(cfg.batch)
    cl = list(claims)
    rows = np.concatenate([np.arange(offs[c], offs[c + 1]) for c in cl])
    nb = NativeBatch(raw["claim_tokens"][cl], raw["claim_len"][cl], raw["evd_tokens"][rows], raw["evd_len"][rows],
                     counts[cl], raw["doc_sources"][cl], raw["query_sources"][cl], raw["labels"][cl], window=cfg.window,
                     n_max=cfg.fixed_num_evidences, device=dev, compact=compact)
    return model, nb


def _cfg():
    from get_amd.synth import SynthConfig
    return SynthConfig(batch=8, emb_dim=64, hidden=64, vocab=500, n_article_src=20, n_claim_src=10, src_dim=16,
                       evd_counts=[3, 7, 1, 30, 12, 5, 2, 9])


def _worker(rank, world, port, q, compact):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    from get_amd.dist import FlatTrainer, shard_claims
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg, seed = _cfg(), 5
        model, nb = _make(cfg, seed, "cuda:0", shard_claims(cfg.batch, rank, world), compact)
        tr = FlatTrainer(model, check_overlap=True)
        tr.attach_overlap()
        for step in range(2):
            tr.zero_grad()
            q_, d_, k_ = nb.inputs()
            loss = torch.nn.functional.cross_entropy(model(q_, d_, **k_), nb.labels)
            loss.backward()
            assert tr._early_work is not None, "the milestone hook did not start the early all-reduce"
            if step == 0:
                tr.allreduce()
                g_avg = (tr.flat_g / world).cpu().clone()
                q.put(("grad", rank, g_avg.numpy(), list(tr.live_names), [int(p.numel()) for p in tr.params]))
                # the same reduced bucket feeds the optimiser: finish the step by hand (allreduce() already ran)
                from get_amd import ops
                tr.t += 1
                ops.adam_step_flat(tr.flat_p, tr.flat_g, tr.flat_m, tr.flat_v, tr.t, lr=tr.lr, betas=tr.betas, eps=tr.eps,
                                   weight_decay=tr.weight_decay, grad_scale=1.0 / world)
                ops.refresh_transposes(tr._matrices)
            else:
                tr.step()
        q.put(("params", rank, tr.flat_p.cpu().numpy()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("compact", [True, False])
def test_two_ranks_real_model_match_the_union_batch(compact):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, compact)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=600) for _ in range(4)]
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    grads = {r: (g, names, sizes) for kind, r, g, names, sizes in [x for x in got if x[0] == "grad"]}
    params = {x[1]: x[2] for x in got if x[0] == "params"}
    assert np.array_equal(grads[0][0], grads[1][0]), "ranks disagree on the reduced bucket"
    assert np.array_equal(params[0], params[1]), "replicas diverged after two optimiser steps"
    # single process, union batch, plain autograd (no trainer, no bucket)
    cfg, seed = _cfg(), 5
    model, nb = _make(cfg, seed, "cuda:0", None, compact)
    q_, d_, k_ = nb.inputs()
    torch.nn.functional.cross_entropy(model(q_, d_, **k_), nb.labels).backward()
    named = dict(model.named_parameters())
    g, names, sizes = grads[0]
    off = 0
    worst = 0.0
    for n, sz in zip(names, sizes):
        ref = named[n].grad.detach().reshape(-1).cpu().numpy().astype(np.float64)
        mine = g[off:off + sz].astype(np.float64)
        scale = max(np.abs(ref).max(), 1e-8)
        worst = max(worst, np.abs(mine - ref).max() / scale)
        off += (sz + 63) // 64 * 64
    # fp32 summation order differs between a 4-claim shard and the 8-claim union (split-K chunking of the weight-gradient
    # GEMMs, row-tile boundaries): the worst parameter sits at ~1e-5 of its own gradient scale (1e-5 rel)
    assert worst <= 3e-5, f"averaged 2-rank gradient differs from the union-batch gradient by {worst:.2e} (relative)"


def _worker_rccl_single(port, q, use_group):
    """One rank on cuda:0.  use_group: a world_size-1 RCCL ("nccl") group with always_reduce -- librccl is loaded, a
    communicator is created on the device, the early range goes out as an asynchronous device all-reduce from inside
    backward and the late range as a blocking one, all on RCCL's own stream with the event hand-offs torch.distributed
    inserts.  Without the group the same two steps run with no collective at all."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    from get_amd.dist import FlatTrainer, pin_rccl_for_parity
    torch.cuda.set_device(0)
    pinned = pin_rccl_for_parity()
    if use_group:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        cfg, seed = _cfg(), 5
        model, nb = _make(cfg, seed, "cuda:0", None, True)
        tr = FlatTrainer(model, check_overlap=use_group, always_reduce=use_group)
        if use_group:
            tr.broadcast_parameters(0)
        tr.attach_overlap()
        early_seen = 0
        for step in range(3):
            tr.zero_grad()
            q_, d_, k_ = nb.inputs()
            loss = torch.nn.functional.cross_entropy(model(q_, d_, **k_), nb.labels)
            loss.backward()
            early_seen += int(tr._early_work is not None)
            tr.step()
        torch.cuda.synchronize()
        backend = dist.get_backend() if use_group else "none"
        q.put((use_group, tr.flat_p.cpu().numpy(), tr.comm_calls, tr.comm_bytes, early_seen, backend, tr.numel, pinned))
        if use_group:
            dist.barrier()
    finally:
        if use_group:
            dist.destroy_process_group()


def test_rccl_world_size_one_runs_the_overlapped_allreduce_on_the_device():
    """RCCL itself had never executed.  A 1-GPU box can still load RCCL, build a communicator and push
    the flat bucket through ncclAllReduce (sum over one rank = identity): three FlatTrainer steps with attach_overlap()
    under a world_size-1 "nccl" group must leave the parameters BIT-identical to the same steps without any group."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    res = {}
    for use_group in (True, False):
        q = ctx.Queue()
        p = ctx.Process(target=_worker_rccl_single, args=(_free_port(), q, use_group))
        p.start()
        got = q.get(timeout=600)
        p.join(300)
        assert p.exitcode == 0
        res[use_group] = got
    _, p_rccl, calls, nbytes, early_seen, backend, numel, pinned = res[True]
    _, p_plain, calls0, nbytes0, early0, _, _, _ = res[False]
    assert backend == "nccl"
    assert pinned == {"NCCL_ALGO": "Ring", "NCCL_PROTO": "Simple"}
    assert early_seen == 3, "the milestone hook did not start the asynchronous early all-reduce"
    assert calls == 6 and nbytes == 3 * numel * 4           # early + late range per step = the whole bucket once
    assert calls0 == 0 and early0 == 0
    assert np.array_equal(p_rccl, p_plain), "three steps through RCCL differ from three steps without a collective"


def _worker_libcomm_single(port, q, use_comm):
    """One rank on cuda:0 with the LIBRARY's communicator (gh_comm_init / gh_flat_allreduce, include/get_hip.h) instead of
    torch.distributed's: the 128-byte id travels through a world_size-1 gloo group (LibComm.from_process_group), the
    broadcast and both all-reduce ranges run as RCCL calls enqueued by libget_hip.so on the trainer's own streams."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    from get_amd.dist import FlatTrainer, LibComm
    torch.cuda.set_device(0)
    comm = None
    if use_comm:
        dist.init_process_group("gloo", rank=0, world_size=1)
        comm = LibComm.from_process_group(device="cuda:0")
    try:
        cfg, seed = _cfg(), 5
        model, nb = _make(cfg, seed, "cuda:0", None, True)
        tr = FlatTrainer(model, check_overlap=use_comm, always_reduce=use_comm, comm=comm)
        if use_comm:
            tr.broadcast_parameters(0)
        tr.attach_overlap()
        early_seen = 0
        for step in range(3):
            tr.zero_grad()
            q_, d_, k_ = nb.inputs()
            loss = torch.nn.functional.cross_entropy(model(q_, d_, **k_), nb.labels)
            loss.backward()
            early_seen += int(tr._early_work is not None)
            tr.step()
        torch.cuda.synchronize()
        extra = None
        if use_comm:
            # the collective itself: sum over one rank is the identity, on a buffer the trainer does not own
            t = torch.arange(1 << 20, device="cuda:0", dtype=torch.float32)
            ref = t.clone()
            comm.all_reduce(t)
            comm.broadcast(t, 0)
            torch.cuda.synchronize()
            extra = (bool(torch.equal(t, ref)), comm.info(), comm.library, tr.world)
            comm.close()
        q.put((use_comm, tr.flat_p.cpu().numpy(), tr.comm_calls, tr.comm_bytes, early_seen, tr.numel, extra))
    finally:
        if use_comm:
            dist.destroy_process_group()


def test_library_owned_rccl_communicator_world_size_one():
    """SURVEY 8(b) `flat_allreduce` as a C-ABI export: three overlapped FlatTrainer steps whose collectives are
    gh_flat_allreduce / gh_flat_broadcast calls on a communicator built by gh_comm_init must leave the parameters
    BIT-identical to the same steps without any collective, and the loaded librccl is reported."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    res = {}
    for use_comm in (True, False):
        q = ctx.Queue()
        p = ctx.Process(target=_worker_libcomm_single, args=(_free_port(), q, use_comm))
        p.start()
        got = q.get(timeout=600)
        p.join(300)
        assert p.exitcode == 0
        res[use_comm] = got
    _, p_lib, calls, nbytes, early_seen, numel, extra = res[True]
    _, p_plain, calls0, nbytes0, early0, _, _ = res[False]
    ident, (rank, world), library, tr_world = extra
    assert ident and (rank, world) == (0, 1) and tr_world == 1
    assert "rccl" in library
    assert early_seen == 3, "the milestone hook did not start the early all-reduce on the library's communicator"
    assert calls == 6 and nbytes == 3 * numel * 4
    assert calls0 == 0 and early0 == 0
    assert np.array_equal(p_lib, p_plain), "three steps through gh_flat_allreduce differ from three steps without a collective"
