"""Data-parallel training glue: one flat fp32 bucket of the LIVE parameters, one RCCL all-reduce of
its gradient per step, one fused Adam kernel over it.

The reference is single-process; claims are independent (8(e)), so the path shards
by claims with exactly one exchange step: all-reduce(sum) of the flat gradient, scaled by 1/world in
the optimiser.  Parameters that never receive a gradient in GET -- the two dead LSTMs, ``trans`` and
the GSL word scorer (no gradient flows through top-k, wrapper.py:219) -- are left out of the bucket
and of the optimiser, matching torch.optim.Adam's skipping of ``grad is None`` (no weight decay).
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence

import torch
import torch.distributed as dist


def pin_rccl_for_parity(env=None):
    """Fix RCCL's algorithm / protocol choice BEFORE init_process_group: the summation order of an
    all-reduce is a function of the algorithm (ring vs tree) and protocol RCCL picks per message size and topology; with
    both pinned, two runs of the same world size reduce in the same order, so multi-GPU parity runs compare like with
    like.  Values already present in the environment win."""
    env = os.environ if env is None else env
    env.setdefault("NCCL_ALGO", "Ring")
    env.setdefault("NCCL_PROTO", "Simple")
    return {k: env[k] for k in ("NCCL_ALGO", "NCCL_PROTO")}

DEAD_PREFIXES = ("bilstm.", "query_bilstm.", "trans.", "ggnn_with_gsl.word_scorer1.")
# Parameters whose gradients are final only at the very end of the backward pass (the first evidence cell, the claim
# branch and the claim-source table, which autograd schedules after it).  Everything else -- head, both attentions,
# the article-source table and the second evidence cell, 76 % of the bucket -- is final once the gradient w.r.t. the
# first cell's output exists, so its share of the all-reduce can run underneath the rest of the backward.
# The word-embedding table belongs here too: when it is trainable (embedding_freeze=False) its gradient is scattered
# by the FIRST evidence cell's and the claim cell's backward, i.e. after the milestone.
LATE_PREFIXES = ("ggnn_with_gsl.feat_prop1.", "ggnn4claim_1.", "claim_source_embs.", "embedding.")


def live_parameters(model: torch.nn.Module):
    return [(n, p) for n, p in model.named_parameters()
            if p.requires_grad and not n.startswith(DEAD_PREFIXES)]


def shard_claims(n_claims: int, rank: int, world: int, evd_counts=None):
    """Claim indices of `rank`'s shard of a global batch.  Every rank gets the same NUMBER of claims (equal claim counts
    keep the mean-reduced loss of the union equal to the average of the per-rank losses).

    evd_counts None: contiguous slices of the (already shuffled) claim list.
    evd_counts (n_claims,): sort-then-stripe -- the work of a claim is its evidence count (pairs, the
    per-claim slices of char_man_fitter_query_repr1.py:207-217), which is ragged on real data (Snopes: mean 6.9, max 26),
    so contiguous slices leave the ranks with unequal B1 and the step waits for the heaviest.  Claims are sorted by
    evidence count (descending, stable) and dealt to the ranks in boustrophedon order (0..W-1, W-1..0, ...): each round
    of 2W claims gives every rank one claim from the heavy and one from the light half of the round."""
    assert n_claims % world == 0, "global batch must divide by the number of ranks"
    per = n_claims // world
    if evd_counts is None:
        return range(rank * per, (rank + 1) * per)
    import numpy as np
    c = np.asarray(evd_counts, dtype=np.int64).reshape(-1)
    assert c.shape[0] == n_claims, "one evidence count per claim"
    order = np.argsort(-c, kind="stable")
    pos = np.arange(n_claims)
    rnd, k = pos // world, pos % world
    owner = np.where(rnd % 2 == 0, k, world - 1 - k)
    mine = order[owner == rank]
    assert mine.shape[0] == per
    return [int(i) for i in np.sort(mine)]          # ascending: keeps the claim-major order of the batch tensors


class _StreamWork:
    """Work handle of a collective that was ENQUEUED on a HIP stream (library-owned communicator): ``wait()`` orders the
    caller's current stream behind it, as the work object of an asynchronous torch.distributed collective does."""

    def __init__(self, event):
        self._event = event

    def wait(self):
        torch.cuda.current_stream().wait_event(self._event)
        return True

    def is_completed(self):
        return self._event.query()


class LibComm:
    """RCCL communicator owned by libget_hip.so (``gh_comm_*`` / ``gh_flat_allreduce``, include/get_hip.h): what a
    non-Python host binds for the path's one exchange step.  Collectives are enqueued on the
    CURRENT torch stream of the communicator's device and are in place.

    The 128-byte id that rank 0 creates has to reach every rank by some channel; ``from_process_group`` uses an existing
    torch.distributed group of any backend (gloo is enough) for that one hand-over, ``single`` builds a 1-rank
    communicator (RCCL loaded, communicator on the device, device all-reduce = identity) for single-GPU boxes."""

    def __init__(self, id_bytes: bytes, rank: int, world: int, device=None):
        import ctypes
        from . import _lib
        assert len(id_bytes) == 128
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if dev.type != "cuda":
            raise ValueError(f"get_amd: LibComm needs a GPU device, got {dev}")
        if dev.index is None:          # "cuda" without an index never equals a tensor's device ("cuda:0"): bind to the current one
            dev = torch.device("cuda", torch.cuda.current_device())
        self.device, self.rank, self.world = dev, int(rank), int(world)
        self._comm_stream = None       # asynchronous collectives get a stream of their own (all_reduce_async)
        # One communicator serialises its collectives: a collective enqueued on the caller's stream while an asynchronous one
        # is still pending on the communicator's stream would let two streams drive the same RCCL communicator concurrently
        # (possible deadlock / cross-rank misordering).  Every enqueue therefore first orders its stream behind the last
        # asynchronous collective (_pending): FlatTrainer's wait()-before-next-collective discipline is no longer load-bearing.
        self._pending = None
        buf = ctypes.create_string_buffer(bytes(id_bytes), 128)
        out = ctypes.c_void_p()
        with torch.cuda.device(dev):
            _lib.call("gh_comm_init", ctypes.cast(buf, ctypes.c_void_p), self.rank, self.world, ctypes.addressof(out))
        self._comm = out.value
        self.library = (_lib.load().gh_comm_library() or b"").decode()

    @staticmethod
    def new_id() -> bytes:
        import ctypes
        from . import _lib
        buf = ctypes.create_string_buffer(128)
        _lib.call("gh_comm_unique_id", ctypes.cast(buf, ctypes.c_void_p))
        return buf.raw

    @classmethod
    def single(cls, device=None) -> "LibComm":
        return cls(cls.new_id(), 0, 1, device)

    @classmethod
    def from_process_group(cls, group=None, device=None) -> "LibComm":
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [cls.new_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        return cls(box[0], rank, world, device)

    def info(self):
        import ctypes
        from . import _lib
        r, w = ctypes.c_int(-1), ctypes.c_int(-1)
        _lib.call("gh_comm_info", self._comm, ctypes.addressof(r), ctypes.addressof(w))
        return r.value, w.value

    def _check(self, t: torch.Tensor):
        if self._comm is None:
            raise RuntimeError("get_amd: LibComm used after close()")
        if not (t.is_cuda and t.device == self.device and t.dtype == torch.float32 and t.is_contiguous()):
            raise RuntimeError("get_amd: LibComm collectives take contiguous fp32 tensors on the communicator's device")

    def all_reduce(self, t: torch.Tensor):
        """In-place sum over the ranks, enqueued on the current stream."""
        from . import _lib
        self._check(t)
        with torch.cuda.device(self.device):
            self._order_behind_pending(torch.cuda.current_stream(self.device))
            _lib.call("gh_flat_allreduce", self._comm, t.data_ptr(), t.numel(), _lib.stream())

    def _order_behind_pending(self, stream):
        if self._pending is not None:
            stream.wait_event(self._pending)
            self._pending = None

    def all_reduce_async(self, t: torch.Tensor) -> "_StreamWork":
        """In-place sum over the ranks on the communicator's OWN stream, ordered behind everything the current stream has
        been given so far; returns a work handle whose wait() orders the then-current stream behind the collective.
        Nothing the current stream is given afterwards waits for the collective (enqueued in the auxiliary stream itself,
        the early all-reduce used to hold up that stream's later weight-gradient launches and, through the backward's final
        join, the main stream)."""
        from . import _lib
        self._check(t)
        with torch.cuda.device(self.device):
            if self._comm_stream is None:
                self._comm_stream = torch.cuda.Stream(device=self.device)
            cs = self._comm_stream
            cs.wait_stream(torch.cuda.current_stream(self.device))
            t.record_stream(cs)
            # (two asynchronous collectives follow each other on the communicator's stream: ordered by the stream itself)
            _lib.call("gh_flat_allreduce", self._comm, t.data_ptr(), t.numel(), cs.cuda_stream)
            ev = torch.cuda.Event()
            ev.record(cs)
            self._pending = ev
        return _StreamWork(ev)

    def broadcast(self, t: torch.Tensor, root: int = 0):
        from . import _lib
        self._check(t)
        with torch.cuda.device(self.device):
            self._order_behind_pending(torch.cuda.current_stream(self.device))
            _lib.call("gh_flat_broadcast", self._comm, t.data_ptr(), t.numel(), int(root), _lib.stream())

    def close(self):
        from . import _lib
        if self._comm is not None:
            torch.cuda.synchronize(self.device)
            _lib.call("gh_comm_destroy", self._comm)
            self._comm = None


class FlatTrainer:
    """Owns flat parameter / gradient / Adam-moment buffers; ``param.data`` and ``param.grad`` of every
    live parameter are views into them, so autograd accumulates straight into the all-reduce bucket."""

    def __init__(self, model: torch.nn.Module, lr=1e-4, weight_decay=1e-3, betas=(0.9, 0.999), eps=1e-8,
                 process_group=None, late_prefixes: Sequence[str] = LATE_PREFIXES, check_overlap: bool = False,
                 always_reduce: bool = False, comm: "LibComm" = None):
        self.model = model
        # comm: a library-owned RCCL communicator (LibComm) carries the collectives instead of torch.distributed -- the same
        # calls a C / C++ host of the library would make (gh_flat_allreduce on the caller's stream, stream-ordered)
        self.comm = comm
        # always_reduce: issue the collectives even in a 1-rank group (a world_size-1 RCCL group on a single-GPU box then
        # exercises communicator set-up, the device all-reduce and the asynchronous early range; tests/test_gpu_dist.py)
        self._always_reduce = bool(always_reduce)
        self.comm_bytes = 0          # bytes handed to all-reduce since construction (bench.py reports them per step)
        self.comm_calls = 0
        self.lr, self.weight_decay, self.betas, self.eps = lr, weight_decay, betas, eps
        self.group = process_group
        live = live_parameters(model)
        # bucket order: early-final gradients first, late ones at the end (one contiguous range each)
        late_prefixes = tuple(late_prefixes)
        live = [x for x in live if not x[0].startswith(late_prefixes)] + [x for x in live if x[0].startswith(late_prefixes)]
        self._early_work = None
        self._overlap_ok = True
        # check_overlap: verify on every step that no "early" gradient changed after its all-reduce was started (one
        # extra reduction + host sync per step: a debugging aid for new model variants, off in production)
        self._check_overlap = bool(check_overlap)
        self._early_snapshot = None
        self.live_names: List[str] = [n for n, _ in live]
        self.params = [p for _, p in live]
        # every parameter starts on a 256-byte boundary of the flat buffers: the kernels take weights and gradients
        # straight from these views and need 16-byte aligned rows (a 2-element bias would otherwise shift everything
        # behind it onto the slow, scalar-load GEMM path); the padding elements stay zero under Adam
        ALIGN = 64
        slot = lambda p: (p.numel() + ALIGN - 1) // ALIGN * ALIGN
        total = sum(slot(p) for p in self.params)
        self.n_early = sum(slot(p) for n, p in live if not n.startswith(late_prefixes))
        dev = self.params[0].device
        self.flat_p = torch.zeros(total, device=dev, dtype=torch.float32)
        self.flat_g = torch.zeros(total, device=dev, dtype=torch.float32)
        self.flat_m = torch.zeros(total, device=dev, dtype=torch.float32)
        self.flat_v = torch.zeros(total, device=dev, dtype=torch.float32)
        off = 0
        self._views = []
        for p in self.params:
            n = p.numel()
            self.flat_p[off:off + n].copy_(p.data.reshape(-1))
            p.data = self.flat_p[off:off + n].view_as(p)
            gv = self.flat_g[off:off + n].view_as(p)
            p.grad = gv
            p._gh_direct_grad = True          # kernels may accumulate straight into this view (ops._direct)
            self._views.append(gv)
            off += slot(p)
        self.numel = total
        self.t = 0
        # matrices whose transposed copy the forward GEMMs consume (everything but embedding tables)
        emb_ids = {id(m.weight) for m in model.modules() if isinstance(m, torch.nn.Embedding)}
        self._matrices = [p for p in self.params if p.dim() == 2 and id(p) not in emb_ids and min(p.shape) > 1]

    # -- checkpoint / resume (the reference saves only net.state_dict(), char_man_fitter_query_repr1.py:143-144;
    #    multi-GPU runs also need the optimiser moments and step count to resume bit-identically)
    def state_dict(self) -> dict:
        return {"t": self.t, "m": self.flat_m.detach().cpu().clone(), "v": self.flat_v.detach().cpu().clone(),
                "names": list(self.live_names), "numel": self.numel,
                "hyper": {"lr": self.lr, "weight_decay": self.weight_decay, "betas": tuple(self.betas), "eps": self.eps}}

    def load_state_dict(self, sd: dict):
        assert sd["names"] == self.live_names and sd["numel"] == self.numel, "optimizer state does not match this model"
        self.t = int(sd["t"])
        self.flat_m.copy_(sd["m"].to(self.flat_m.device))
        self.flat_v.copy_(sd["v"].to(self.flat_v.device))

    @property
    def world(self) -> int:
        if self.comm is not None:
            return self.comm.world
        return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

    def _reducing(self) -> bool:
        return self.world > 1 or (self._always_reduce and (self.comm is not None or (dist.is_available() and dist.is_initialized())))

    def _all_reduce(self, t: torch.Tensor, async_op: bool = False):
        self.comm_bytes += t.numel() * t.element_size()
        self.comm_calls += 1
        if self.comm is not None:
            # blocking form: enqueued on the current stream; asynchronous form: on the communicator's own stream, behind
            # the current one -- the caller later orders its stream behind the returned handle
            if async_op:
                return self.comm.all_reduce_async(t)
            self.comm.all_reduce(t)
            return None
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)

    def broadcast_parameters(self, src: int = 0):
        """Make every replica identical to rank `src` with TWO collectives: the flat bucket of the live parameters, and
        one packed buffer of everything outside it that the forward reads (the never-trained GSL scorer, frozen
        embedding tables, the dead-but-present LSTM / trans parameters a checkpoint carries).  The per-tensor loop this
        replaces issued ~90 broadcasts."""
        if self.comm is None and not (dist.is_available() and dist.is_initialized()):
            return
        from . import ops
        bcast = (lambda t: self.comm.broadcast(t, src)) if self.comm is not None else (lambda t: dist.broadcast(t, src=src, group=self.group))
        bcast(self.flat_p)
        live = {id(p) for p in self.params}
        rest = [p for p in self.model.parameters() if id(p) not in live]
        rest += [b for b in self.model.buffers() if b.is_floating_point()]
        if rest:
            pack = torch.cat([t.detach().reshape(-1).float() for t in rest])
            bcast(pack)
            off = 0
            with torch.no_grad():
                for t in rest:
                    n = t.numel()
                    t.copy_(pack[off:off + n].view_as(t))
                    off += n
        ops.bump_weight_epoch()          # parameters were rewritten in place: drop every cached derivative

    def zero_grad(self):
        assert self._early_work is None, "zero_grad() while an all-reduce is in flight (call step() / allreduce() first)"
        self.flat_g.zero_()
        for p, gv in zip(self.params, self._views):
            if p.grad is None or p.grad.data_ptr() != gv.data_ptr():
                p.grad = gv                      # re-attach if something replaced the view

    def allreduce_early_async(self):
        """Start the all-reduce of the early-final part of the bucket (call it once those gradients are complete in
        stream order, e.g. from the hook ``attach_overlap`` installs).  No-op for a single rank.  A second call while
        the first collective is still pending means a second backward pass ran before ``allreduce()`` / ``step()``
        (gradient accumulation): its early gradients would be added to an already reduced range, so this is rejected
        -- accumulate with the overlap detached, or call ``allreduce()`` after every backward."""
        from . import ops
        if self._early_work is not None:
            raise RuntimeError("get_amd: a second backward pass reached the all-reduce milestone while the first early "
                               "all-reduce is still pending; gradient accumulation needs detach_overlap() (one "
                               "all-reduce per step) or an allreduce() after every backward")
        if not (self._reducing() and 0 < self.n_early < self.numel and self._overlap_ok):
            return
        # Early-range gradients come from two streams: the caller's (head, attentions' dX side) and the auxiliary one
        # (few-row weight gradients, the second evidence cell's weight-gradient GEMMs).  The collective is issued FROM the
        # auxiliary stream after it has been ordered behind the caller's stream: it then waits for both, while the
        # caller's stream -- the first evidence cell's dX chain -- is not held up by the weight-gradient GEMMs.
        side = ops.pending_side_stream()
        if self._check_overlap:
            ops.side_join()
            side = None
            self._early_snapshot = self.flat_g[:self.n_early].clone()
        try:
            if side is not None:
                side.wait_stream(torch.cuda.current_stream(side.device))
                with torch.cuda.stream(side):
                    self._early_work = self._all_reduce(self.flat_g[:self.n_early], async_op=True)
            else:
                ops.side_join()
                self._early_work = self._all_reduce(self.flat_g[:self.n_early], async_op=True)
        except Exception as e:      # a backend without async collectives: keep training with the single all-reduce
            self._overlap_ok = False
            self._early_work = None
            import warnings
            warnings.warn(f"get_amd: overlapped all-reduce disabled ({e!r}); falling back to one all-reduce per step")

    def allreduce(self):
        """All-reduce(sum) of the gradient bucket (RCCL over xGMI on GPUs, gloo in CPU tests): the whole bucket in one
        collective, or -- when the early part is already in flight -- a wait on it plus the late remainder."""
        from . import ops
        ops.side_join()
        if self._reducing():
            if self._early_work is not None:
                self._early_work.wait()
                self._early_work = None
                if self._check_overlap and self._early_snapshot is not None:
                    # every rank contributed its snapshot, so the reduced range must equal the sum of the snapshots:
                    # reduce the snapshots again and compare (a gradient that landed after the milestone shows up here)
                    ref = self._early_snapshot
                    if self.comm is not None:
                        self.comm.all_reduce(ref)
                    else:
                        dist.all_reduce(ref, op=dist.ReduceOp.SUM, group=self.group)
                    bad = float((ref - self.flat_g[:self.n_early]).abs().max())
                    self._early_snapshot = None
                    if bad != 0.0:
                        raise RuntimeError(f"get_amd: an early-bucket gradient changed after its all-reduce started (max "
                                           f"diff {bad:.3e}); a parameter is missing from late_prefixes")
                self._all_reduce(self.flat_g[self.n_early:])
            else:
                self._all_reduce(self.flat_g)

    def attach_overlap(self, module=None):
        """Overlap the early part of the all-reduce with the tail of the backward pass: the module that owns the
        first evidence cell (``model.ggnn_with_gsl``) calls ``grad_milestone_hook`` when the gradient w.r.t. that
        cell's output has been produced, i.e. when every early gradient is final."""
        module = module if module is not None else getattr(self.model, "ggnn_with_gsl", None)
        if module is None:
            raise ValueError("attach_overlap: no module with a grad_milestone_hook slot")
        module.grad_milestone_hook = self.allreduce_early_async
        self._overlap_module = module

    def detach_overlap(self):
        """Back to one all-reduce per step (required for gradient accumulation over several backward passes)."""
        m = getattr(self, "_overlap_module", None)
        if m is not None:
            m.grad_milestone_hook = None
            self._overlap_module = None

    def step(self):
        """all-reduce + fused Adam on the flat bucket (gradient averaged over ranks inside the kernel)."""
        from . import ops
        self.allreduce()
        self.t += 1
        ops.adam_step_flat(self.flat_p, self.flat_g, self.flat_m, self.flat_v, self.t, lr=self.lr, betas=self.betas,
                           eps=self.eps, weight_decay=self.weight_decay, grad_scale=1.0 / self.world)
        ops.refresh_transposes(self._matrices)     # k-major copies for the next forward, one launch
