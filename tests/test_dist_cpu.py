"""N>1 data-parallel path on CPU: 2 processes, gloo.  Covers claim sharding, the flat gradient bucket
(live-parameter selection, grad views) and the single all-reduce; the Adam kernel itself is GPU-only
and covered by tests/test_gpu_ops.py."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from get_amd.dist import DEAD_PREFIXES, FlatTrainer, live_parameters, shard_claims


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class Toy(torch.nn.Module):
    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.live = torch.nn.Linear(6, 3)
        self.trans = torch.nn.Linear(4, 4)          # dead by name, as in GET
        self.frozen = torch.nn.Embedding(5, 6)
        self.frozen.weight.requires_grad = False

    def forward(self, x):
        return self.live(x)


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    model = Toy()
    tr = FlatTrainer(model)
    assert tr.live_names == ["live.weight", "live.bias"]
    assert tr.world == world
    torch.manual_seed(123)
    x = torch.randn(8, 6)
    y = torch.randint(0, 3, (8,))
    idx = list(shard_claims(8, rank, world))
    tr.zero_grad()
    loss = torch.nn.functional.cross_entropy(model(x[idx]), y[idx])
    loss.backward()
    assert model.live.weight.grad.data_ptr() == tr.flat_g.data_ptr()        # autograd wrote into the bucket
    tr.allreduce()
    g = torch.cat([v.reshape(-1) for v in tr._views]) / world
    if rank == 0:
        ref = Toy()
        torch.nn.functional.cross_entropy(ref(x), y).backward()
        exp = torch.cat([ref.live.weight.grad.reshape(-1), ref.live.bias.grad.reshape(-1)])
        out.put(float((g - exp).abs().max()))
    dist.barrier()
    dist.destroy_process_group()


class Toy2(torch.nn.Module):
    """late(x) -> h -> early(h): in backward `early` is final when the gradient w.r.t. h exists."""

    def __init__(self):
        super().__init__()
        torch.manual_seed(1)
        self.late = torch.nn.Linear(6, 5)
        self.early = torch.nn.Linear(5, 3)
        self.grad_milestone_hook = None

    def forward(self, x):
        h = torch.tanh(self.late(x))
        if self.grad_milestone_hook is not None:
            hook = self.grad_milestone_hook
            h.register_hook(lambda g: (hook(), g)[1])
        return self.early(h)


def _worker_overlap(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    model = Toy2()
    tr = FlatTrainer(model, late_prefixes=("late.",))
    assert tr.live_names == ["early.weight", "early.bias", "late.weight", "late.bias"]      # early-final first
    assert tr.n_early == 64 + 64 and tr.numel == 4 * 64          # every parameter owns a 256-byte aligned slot
    assert all(v.data_ptr() % 256 == tr.flat_g.data_ptr() % 256 for v in tr._views)
    tr.attach_overlap(model)
    torch.manual_seed(321)
    x = torch.randn(8, 6)
    y = torch.randint(0, 3, (8,))
    idx = list(shard_claims(8, rank, world))
    fired = []
    orig = tr.allreduce_early_async
    model.grad_milestone_hook = lambda: (fired.append(model.late.weight.grad.abs().sum().item()), orig())
    for _ in range(2):                                   # two steps: the in-flight handle is consumed and re-armed
        tr.zero_grad()
        torch.nn.functional.cross_entropy(model(x[idx]), y[idx]).backward()
        assert tr._early_work is not None                # started from inside backward ...
        tr.allreduce()
        assert tr._early_work is None
    assert fired == [0.0, 0.0]                           # ... before the late gradients existed
    g = torch.cat([v.reshape(-1) for v in tr._views]) / world
    if rank == 0:
        ref = Toy2()
        torch.nn.functional.cross_entropy(ref(x), y).backward()
        exp = torch.cat([ref.early.weight.grad.reshape(-1), ref.early.bias.grad.reshape(-1),
                         ref.late.weight.grad.reshape(-1), ref.late.bias.grad.reshape(-1)])
        out.put(float((g - exp).abs().max()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_overlapped_allreduce_matches_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_overlap, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) <= 1e-6


def test_two_rank_flat_allreduce_matches_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) <= 1e-6


def test_live_parameter_selection_on_the_real_model():
    from get_amd import modules
    from get_amd.synth import make_embeddings
    from oracle.cases_model import MODEL_CASES
    from tests.util import load
    cfg, seed = MODEL_CASES["small"]
    emb, art, clm = make_embeddings(cfg, seed)
    model = modules.Graph_basedSemantiStructure(cfg.model_params(emb, art, clm))
    z, meta = load("g7_model_small.npz")
    live = live_parameters(model)
    none = set(meta["none_grads"])
    assert {n for n, _ in live} == {n for n, _ in model.named_parameters()} - none     # same split as the reference
    assert sum(p.numel() for _, p in live) == meta["n_live"]
    assert all(n.startswith(DEAD_PREFIXES) or n == "embedding.weight" for n in none)


def test_shard_claims_partitions():
    got = sorted(i for r in range(4) for i in shard_claims(256, r, 4))
    assert got == list(range(256))
    assert len(shard_claims(256, 3, 8)) == 32


def _worker_second_backward(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    model = Toy2()
    tr = FlatTrainer(model, late_prefixes=("late.",))
    tr.attach_overlap(model)
    torch.manual_seed(7)
    x, y = torch.randn(4, 6), torch.randint(0, 3, (4,))
    tr.zero_grad()
    torch.nn.functional.cross_entropy(model(x), y).backward()
    try:                                                  # gradient accumulation under overlap must be rejected loudly
        torch.nn.functional.cross_entropy(model(x), y).backward()
        ok = False
    except RuntimeError as e:
        ok = "second backward" in str(e)
    tr.allreduce()                                        # the pending collective is still consumed cleanly
    tr.detach_overlap()                                   # ... and accumulation works once the overlap is detached
    tr.zero_grad()
    for _ in range(2):
        torch.nn.functional.cross_entropy(model(x), y).backward()
    tr.allreduce()
    if rank == 0:
        out.put(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_second_backward_under_overlap_is_rejected():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_second_backward, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_trainable_word_embedding_sits_in_the_late_range():
    """embedding.weight receives its gradient from the first evidence cell / the claim cell, i.e. AFTER the
    overlap milestone -- it must not be part of the early all-reduce."""
    from get_amd import modules
    from get_amd.synth import make_embeddings
    from oracle.cases_model import MODEL_CASES
    cfg, seed = MODEL_CASES["small"]
    emb, art, clm = make_embeddings(cfg, seed)
    params = cfg.model_params(emb, art, clm)
    params["embedding_freeze"] = False
    model = modules.Graph_basedSemantiStructure(params)
    tr = FlatTrainer(model)
    assert "embedding.weight" in tr.live_names
    off = 0
    for n, p in zip(tr.live_names, tr.params):
        if n == "embedding.weight":
            assert off >= tr.n_early, "trainable embedding table landed in the early all-reduce range"
        off += (p.numel() + 63) // 64 * 64


def test_sort_then_stripe_balances_snopes_shaped_batches():
    """SURVEY 8(e) / VERDICT r2: on the empirical Snopes evidence histogram (mean 6.9, max 26) the striped shards must
    carry nearly equal pair counts (max/min B1 per rank <= 1.1), keep equal claim counts and partition the batch."""
    from get_amd.synth import snopes_evidence_counts
    worst_striped, worst_contig = 1.0, 1.0
    for seed in range(20):
        counts = snopes_evidence_counts(np.random.default_rng(seed), 256)
        for world in (2, 4, 8):
            shards = [shard_claims(256, r, world, counts) for r in range(world)]
            assert sorted(i for s in shards for i in s) == list(range(256))
            assert all(len(s) == 256 // world for s in shards)
            assert all(list(s) == sorted(s) for s in shards)
            b1 = [int(counts[list(s)].sum()) for s in shards]
            worst_striped = max(worst_striped, max(b1) / min(b1))
            b1c = [int(counts[list(shard_claims(256, r, world))].sum()) for r in range(world)]
            worst_contig = max(worst_contig, max(b1c) / min(b1c))
    assert worst_striped <= 1.1, worst_striped
    assert worst_contig > worst_striped          # (contiguous slices reach ~1.5 at 8 ranks)
    # B = 32 claims per rank at 8 ranks, the configs[3] split
    assert shard_claims(16, 1, 2, np.arange(16)) == [1, 2, 5, 6, 9, 10, 13, 14]


def _worker_broadcast(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)                      # replicas start DIFFERENT
    model = Toy()
    with torch.no_grad():
        for p in model.parameters():
            p.add_(torch.randn_like(p))
    tr = FlatTrainer(model)
    calls = []
    orig = dist.broadcast
    dist.broadcast = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        tr.broadcast_parameters(src=0)
    finally:
        dist.broadcast = orig
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    if rank == 0:
        out.put((len(calls), all(torch.equal(g, gathered[0]) for g in gathered),
                 model.live.weight.data_ptr() == tr.flat_p.data_ptr()))
    dist.barrier()
    dist.destroy_process_group()


def test_broadcast_parameters_is_two_collectives_and_covers_dead_and_frozen_tensors():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_broadcast, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    ncalls, same, still_view = q.get(timeout=5)
    assert ncalls == 2 and same and still_view


def test_single_rank_group_reduces_only_when_asked():
    """always_reduce: a world_size-1 group still issues the collectives (the RCCL smoke test on a 1-GPU box relies on it)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        for flag, calls in ((False, 0), (True, 2)):
            model = Toy2()
            tr = FlatTrainer(model, late_prefixes=("late.",), always_reduce=flag)
            tr.attach_overlap(model)
            x, y = torch.randn(4, 6), torch.randint(0, 3, (4,))
            tr.zero_grad()
            torch.nn.functional.cross_entropy(model(x), y).backward()
            tr.allreduce()
            assert tr.comm_calls == calls
            assert tr.comm_bytes == (tr.numel * 4 if flag else 0)
    finally:
        dist.destroy_process_group()


def _worker_world8_real_model(rank, world, port, out):
    """One rank of the world-8 rehearsal: the REAL model's parameter set (small case, CPU tensors -- its forward is GPU-only, so a
    backward pass is emulated by writing rank-dependent gradients into the bucket's views in the order the real backward
    produces them: early range, milestone hook, late range)."""
    from get_amd import modules
    from get_amd.dist import LATE_PREFIXES
    from get_amd.synth import make_embeddings
    from oracle.cases_model import MODEL_CASES
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg, seed = MODEL_CASES["small"]
    emb, art, clm = make_embeddings(cfg, seed)
    torch.manual_seed(1000 + rank)                         # replicas start DIFFERENT: the broadcast must make them equal
    model = modules.Graph_basedSemantiStructure(cfg.model_params(emb, art, clm))
    tr = FlatTrainer(model)
    assert tr.world == world and 0 < tr.n_early < tr.numel
    tr.broadcast_parameters(src=0)
    tr.attach_overlap()                                     # model.ggnn_with_gsl.grad_milestone_hook = allreduce_early_async
    hook = model.ggnn_with_gsl.grad_milestone_hook
    assert hook is not None
    names = tr.live_names
    early = [n for n in names if not n.startswith(LATE_PREFIXES)]
    late = [n for n in names if n.startswith(LATE_PREFIXES)]
    assert names == early + late and early and late
    params = dict(model.named_parameters())
    worst = 0.0
    for step in range(3):
        tr.zero_grad()
        fill = lambda i: float((rank + 1) * (i % 7 + 1) + step)      # gradient of parameter i on this rank
        for i, n in enumerate(names):
            if n in late:
                continue
            params[n].grad.fill_(fill(i))
        assert tr._early_work is None
        hook()                                              # what GGNN_with_GSL's backward hook does at the milestone
        assert tr._early_work is not None, "the early range must go out asynchronously at world 8"
        for i, n in enumerate(names):
            if n in late:
                params[n].grad.fill_(fill(i))
        tr.allreduce()                                      # waits for the early range, reduces the late one
        assert tr._early_work is None
        for i, n in enumerate(names):
            exp = sum((r + 1) * (i % 7 + 1) + step for r in range(world))
            worst = max(worst, float((params[n].grad - exp).abs().max()))
    # parameters identical on every rank after the broadcast (dead and frozen tensors included)
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    if rank == 0:
        out.put((worst, all(torch.equal(g, gathered[0]) for g in gathered), tr.comm_calls))
    dist.barrier()
    dist.destroy_process_group()


def test_world8_overlapped_allreduce_on_the_real_models_bucket():
    """the first RCCL run at 8 ranks must not also be the first world-8 run of the overlap logic.  Eight gloo
    ranks, FlatTrainer on the real model's parameters with attach_overlap(): broadcast from rank 0, then three emulated steps --
    early gradients, milestone hook (asynchronous early all-reduce), late gradients, allreduce() -- and every gradient equals
    the sum over the eight ranks' contributions."""
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_world8_real_model, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    worst, same, calls = q.get(timeout=5)
    assert worst == 0.0 and same
    assert calls == 3 * 2                                   # per step: the early range (asynchronous) + the late range (comm_calls counts all-reduces)
