"""Autograd-aware host wrappers over the C-ABI (one ``torch.autograd.Function`` per fused op).

PyTorch supplies device memory, streams and the autograd tape; every numerical
step runs in libget_hip.so.  All tensors are fp32/contiguous on a ROCm device.
"""
from __future__ import annotations

import os
import weakref
from typing import Optional

import torch

from . import _lib
from ._lib import call, ptr, stream

# GET_AMD_CLAIM_STREAM=0 keeps the claim branch on the caller's stream (modules.Graph_basedSemantiStructure.forward)
CLAIM_SIDE_STREAM = os.environ.get("GET_AMD_CLAIM_STREAM", "1") != "0"
_SIDE_STREAMS: dict = {}


def side_stream(device) -> "torch.cuda.Stream":
    """One auxiliary HIP stream per device for work that is independent of the main chain of launches."""
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    s = _SIDE_STREAMS.get(idx)
    if s is None:
        s = torch.cuda.Stream(device=idx)
        _SIDE_STREAMS[idx] = s
    return s


# Weight gradients of the few-row layers (evidence-level attention, head) leave the critical path of backward: with a
# FlatTrainer (gradients accumulate straight into the flat bucket) they are issued on the auxiliary stream and joined
# at the end of the backward pass (and before any all-reduce).  GET_AMD_WGRAD_STREAM=0 keeps them in line.
WGRAD_SIDE_STREAM = os.environ.get("GET_AMD_WGRAD_STREAM", "1") != "0"
WGRAD_FEW_ROWS = 4096
_side_pending: set = set()


def side_join():
    """Make the current stream wait for everything issued on the auxiliary stream so far (no-op when nothing is)."""
    for idx in list(_side_pending):
        torch.cuda.current_stream(idx).wait_stream(_SIDE_STREAMS[idx])
        _side_pending.discard(idx)


def pending_side_stream():
    """The auxiliary stream of the current device if work of the running backward pass is pending on it, else None."""
    idx = torch.cuda.current_device() if torch.cuda.is_available() else None
    return _SIDE_STREAMS.get(idx) if idx in _side_pending else None


def side_mark_backward(device):
    """Called from inside a backward pass when work of that pass runs (or is about to run) on the auxiliary stream -- the
    claim branch's backward, which autograd replays on its forward stream: marks the device pending and queues the join
    for the end of the pass.  With a FlatTrainer the branch's backward returns no gradients to autograd (they land in
    the flat bucket directly), so the engine itself never learns about the side stream and would not join it."""
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    if idx in _SIDE_STREAMS and idx not in _side_pending:
        _side_pending.add(idx)
        torch.autograd.Variable._execution_engine.queue_callback(side_join)


def _side_wgrad(dev, tensors, launch):
    """Run `launch()` (weight-gradient launches only) on the auxiliary stream after the current stream's work;
    `tensors` are the operands it reads.  The join is queued for the end of the running backward pass."""
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    side = side_stream(dev)
    main = torch.cuda.current_stream(idx)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        _lib.ensure_workspace(dev)
        launch()
    for t in tensors:
        if t is not None:
            t.record_stream(side)
    if idx not in _side_pending:
        _side_pending.add(idx)
        torch.autograd.Variable._execution_engine.queue_callback(side_join)


_WEIGHT_EPOCH = 0
_WT_CACHE: dict = {}


def bump_weight_epoch():
    """Invalidate EVERY cached derivative of the weights (transposes, bias sums, packed scalar gates, bf16 twins).  The
    public invalidation API: call it after writing parameters through `.data` / raw pointers (EMA swaps, hand-written
    optimisers) -- such writes do not bump `_version`, so nothing else can notice them."""
    global _WEIGHT_EPOCH
    _WEIGHT_EPOCH += 1
    _WT_CACHE.clear()
    _BF16_CACHE.clear()
    _DERIVED_CACHE.clear()
    if _lib.gemm_mode() == "fp32x3p":       # the library's pre-split weight images: drop them (weights may have been replaced)
        call("gh_fp32x3_clear")


def _bump_trainer_epoch():
    """FlatTrainer.step's own invalidation: the fused optimiser rewrites exactly the parameters of its bucket, so entries
    derived from tensors OUTSIDE the bucket (`frozen`: the never-trained GSL scorer's packed gates, the bf16 twin of a
    frozen embedding table) stay valid."""
    global _WEIGHT_EPOCH
    _WEIGHT_EPOCH += 1
    _WT_CACHE.clear()
    _BF16_CACHE.clear()
    for k in [k for k, v in _DERIVED_CACHE.items() if v[0][-1] != -1]:
        del _DERIVED_CACHE[k]
    if _lib.gemm_mode() == "fp32x3p":       # every image stale; refresh_transposes re-makes them all in one launch
        call("gh_weights_changed")


_X3P_SIG: dict = {}


def fp32x3p_guard(module):
    """gemm mode "fp32x3p" only: the library reads the weight operand of its activation-sized GEMMs from pre-split images
    and cannot see in-place updates by a torch optimiser.  Called at the top of the model's forward: when any parameter's
    `_version` or address moved since the last call, every image is marked stale (re-made by the launch that meets it).
    FlatTrainer rewrites parameters through raw pointers and tells the library itself (_bump_trainer_epoch)."""
    sig = 0
    for q in module.parameters():
        sig = (sig * 1000003 + q._version * 31 + q.data_ptr()) & 0xFFFFFFFFFFFF
    if _X3P_SIG.get(id(module)) != sig:
        _X3P_SIG[id(module)] = sig
        call("gh_weights_changed")


def _direct(p) -> bool:
    """True when gradients of `p` may be accumulated by the kernels straight into `p.grad` (a live view of
    the trainer's flat bucket, see dist.FlatTrainer) instead of being returned to autograd, which
    would add them there with one extra elementwise kernel per parameter."""
    g = getattr(p, "grad", None)
    return (getattr(p, "_gh_direct_grad", False) and g is not None and g.dtype == torch.float32
            and g.is_contiguous() and g.shape == p.shape)


def _f32(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def transposed(w: torch.Tensor) -> torch.Tensor:
    """W[n_out][n_in] -> Wt[n_in][n_out], cached per parameter OBJECT (weak reference), storage
    address, in-place version and weight epoch -- never by address alone, which the caching
    allocator recycles."""
    key = id(w)
    hit = _WT_CACHE.get(key)
    if hit is not None:
        ref, dptr, ver, epoch, wt = hit
        if ref() is w and dptr == w.data_ptr() and ver == w._version and epoch == _WEIGHT_EPOCH:
            return wt
    wc = _f32(w.detach())
    if _lib.gemm_mode() == "fp32x3p":
        # the library keys its pre-split images by operand address: keep the k-major copy in the parameter's persistent
        # buffer (as refresh_transposes does) so that the address stays, and mark the images stale
        hit_p = _WT_PERSIST.get(key)
        if hit_p is None or hit_p[0]() is not w or hit_p[1].shape != (wc.shape[1], wc.shape[0]) or hit_p[1].device != w.device:
            hit_p = (weakref.ref(w), torch.empty((wc.shape[1], wc.shape[0]), device=w.device, dtype=torch.float32))
            _WT_PERSIST[key] = hit_p
        wt = hit_p[1]
        call("gh_weights_changed")
    else:
        wt = torch.empty((wc.shape[1], wc.shape[0]), device=w.device, dtype=torch.float32)
    call("gh_transpose", ptr(wc), ptr(wt), wc.shape[0], wc.shape[1], stream())
    if len(_WT_CACHE) > 512:
        for k in [k for k, v in _WT_CACHE.items() if v[0]() is None]:
            del _WT_CACHE[k]
    _WT_CACHE[key] = (weakref.ref(w), w.data_ptr(), w._version, _WEIGHT_EPOCH, wt)
    return wt


_DERIVED_CACHE: dict = {}
_WT_PERSIST: dict = {}
# bf16 storage mode (BASELINE configs[4]): bf16 twins of a weight matrix and of its transpose, in persistent buffers (their
# addresses are part of the composite entry points' cached descriptor), validity tracked like the transposes'
_BF16_CACHE: dict = {}       # id(w) -> (ref, data_ptr, version, epoch, w16, wt16)
_BF16_PERSIST: dict = {}     # id(w) -> (ref, w16, wt16)


def _bf16_buffers(w: torch.Tensor):
    hit = _BF16_PERSIST.get(id(w))
    if hit is None or hit[0]() is not w or hit[1].shape != w.shape or hit[1].device != w.device:
        if len(_BF16_PERSIST) > 512:
            for k in [k for k, v in _BF16_PERSIST.items() if v[0]() is None]:
                del _BF16_PERSIST[k]
        hit = (weakref.ref(w), torch.empty(tuple(w.shape), device=w.device, dtype=torch.bfloat16),
               torch.empty((w.shape[1], w.shape[0]), device=w.device, dtype=torch.bfloat16))
        _BF16_PERSIST[id(w)] = hit
    return hit[1], hit[2]


def bf16_twins(w: torch.Tensor):
    """(bf16(W) [n_out][n_in], bf16(W^T) [n_in][n_out]) of a weight matrix, cached per parameter object / storage address /
    in-place version / weight epoch; re-made together with the fp32 transpose in ONE launch (gh_weights_refresh), by
    refresh_transposes after the optimiser step or here on a miss."""
    hit = _BF16_CACHE.get(id(w))
    if hit is not None:
        ref, dptr, ver, epoch, w16, wt16 = hit
        if ref() is w and dptr == w.data_ptr() and ver == w._version and epoch == _WEIGHT_EPOCH:
            return w16, wt16
    _bf16_buffers(w)
    refresh_transposes([w])
    hit = _BF16_CACHE[id(w)]
    return hit[4], hit[5]


def derived(tag: str, tensors, fn, frozen: bool = False):
    """Small per-parameter-set cache for values derived from weights (bias sums, packed scalar gates): recomputed only
    when one of the source tensors was replaced, modified in place (``_version``) or the weight epoch moved on (the
    fused optimiser updates parameters through raw pointers).  Saves a dozen tiny elementwise launches per step."""
    key = (tag,) + tuple(id(t) for t in tensors)
    # frozen: the sources are never touched by the optimiser (requires_grad False), so the weight epoch does not matter
    sig = tuple((t.data_ptr(), t._version) for t in tensors) + ((-1,) if frozen else (_WEIGHT_EPOCH,))
    hit = _DERIVED_CACHE.get(key)
    if hit is not None and hit[0] == sig and all(r() is t for r, t in zip(hit[1], tensors)):
        return hit[2]
    with torch.no_grad():
        val = fn()
    if len(_DERIVED_CACHE) > 512:
        for k in [k for k, v in _DERIVED_CACHE.items() if any(r() is None for r in v[1])]:
            del _DERIVED_CACHE[k]
    _DERIVED_CACHE[key] = (sig, tuple(weakref.ref(t) for t in tensors), val)
    return val


def refresh_transposes(weights):
    """Transpose many weight matrices in ONE launch and prime the cache (called by the trainer right
    after the optimiser step, so the next forward finds every k-major operand ready)."""
    import ctypes
    ws = [w for w in weights if w.dim() == 2]
    if not ws:
        return
    n = len(ws)
    # the transposed copies live in PERSISTENT buffers (one per parameter object, rewritten in place every step): their
    # addresses are part of the composite entry points' cached descriptor (get_amd/fused.py), and nothing is allocated
    wts = []
    for w in ws:
        hit = _WT_PERSIST.get(id(w))
        if hit is None or hit[0]() is not w or hit[1].shape != (w.shape[1], w.shape[0]) or hit[1].device != w.device:
            hit = (weakref.ref(w), torch.empty((w.shape[1], w.shape[0]), device=w.device, dtype=torch.float32))
            _WT_PERSIST[id(w)] = hit
        wts.append(hit[1])
    src = (ctypes.c_void_p * n)(*[w.data_ptr() for w in ws])
    dst = (ctypes.c_void_p * n)(*[t.data_ptr() for t in wts])
    rows = (ctypes.c_int * n)(*[w.shape[0] for w in ws])
    cols = (ctypes.c_int * n)(*[w.shape[1] for w in ws])
    cast = lambda a: ctypes.cast(a, ctypes.c_void_p)
    # bf16 twins ride in the same launch for every matrix that has them (i.e. that a bf16-storage cell has asked for)
    twins = [_BF16_PERSIST.get(id(w)) for w in ws]
    twins = [t if (t is not None and t[0]() is w and t[1].device == w.device) else None for t, w in zip(twins, ws)]
    if any(t is not None for t in twins):
        w16 = (ctypes.c_void_p * n)(*[(t[1].data_ptr() if t is not None else None) for t in twins])
        t16 = (ctypes.c_void_p * n)(*[(t[2].data_ptr() if t is not None else None) for t in twins])
        call("gh_weights_refresh", n, cast(src), cast(dst), cast(w16), cast(t16), cast(rows), cast(cols), stream())
        for w, t in zip(ws, twins):
            if t is not None:
                _BF16_CACHE[id(w)] = (weakref.ref(w), w.data_ptr(), w._version, _WEIGHT_EPOCH, t[1], t[2])
    else:
        call("gh_transpose_batch", n, cast(src), cast(dst), cast(rows), cast(cols), stream())
    if _lib.gemm_mode() == "fp32x3p":
        call("gh_fp32x3_refresh", stream())
    for w, wt in zip(ws, wts):
        _WT_CACHE[id(w)] = (weakref.ref(w), w.data_ptr(), w._version, _WEIGHT_EPOCH, wt)


# --------------------------------------------------------------------------- packed adjacency
class PackedAdj:
    """Adjacency of ``n`` graphs with ``r`` padded nodes in the library's packed form
    (include/get_hip.h): neighbour bit rows + either dinv (normalised binary graph built by
    ``graph_build``) or dense fp32 values (any adjacency the reference API hands over), plus an
    optional GSL keep-set."""

    __slots__ = ("bits", "dinv", "vals", "keep", "n", "r", "plan")

    def __init__(self, bits, dinv, vals, keep, n, r, plan=None):
        self.bits, self.dinv, self.vals, self.keep, self.n, self.r = bits, dinv, vals, keep, n, r
        self.plan = plan        # optional RaggedPlan (node-compact row layout) for callers that opt in

    @property
    def words(self):
        return (self.r + 63) // 64

    @property
    def device(self):
        return self.bits.device

    @staticmethod
    def from_dense(adj: torch.Tensor) -> "PackedAdj":
        """(N,R,R) float64/float32 -> packed (exact values kept, i.e. the reference's `.float()`)."""
        _lib.require_cuda(adj)
        assert adj.dim() == 3 and adj.shape[1] == adj.shape[2], "adjacency must be (N,R,R)"
        n, r, _ = adj.shape
        if adj.dtype not in (torch.float64, torch.float32):
            adj = adj.float()
        adj = adj.contiguous()
        w = (r + 63) // 64
        bits = torch.empty((n, r, w), device=adj.device, dtype=torch.int64)
        vals = torch.empty((n, r, r), device=adj.device, dtype=torch.float32)
        fn = "gh_adj_pack_f64" if adj.dtype == torch.float64 else "gh_adj_pack_f32"
        call(fn, ptr(adj), n, r, ptr(bits), ptr(vals), stream())
        return PackedAdj(bits, None, vals, None, n, r)

    def with_keep(self, keep: Optional[torch.Tensor]) -> "PackedAdj":
        return PackedAdj(self.bits, self.dinv, self.vals, keep, self.n, self.r, self.plan)

    def with_plan(self, plan: "RaggedPlan") -> "PackedAdj":
        return PackedAdj(self.bits, self.dinv, self.vals, self.keep, self.n, self.r, plan)

    def to_dense(self) -> torch.Tensor:
        out = torch.empty((self.n, self.r, self.r), device=self.device, dtype=torch.float32)
        call("gh_adj_unpack", ptr(self.bits), ptr(self.dinv), ptr(self.vals), ptr(self.keep), self.n, self.r,
             ptr(out), stream())
        return out

    def _args(self):
        return ptr(self.bits), ptr(self.dinv), ptr(self.vals), ptr(self.keep)


def graph_build(tokens: torch.Tensor, lengths: torch.Tensor, window: int):
    """interactions.py:334-351 on device.  tokens (N,R) raw ids, lengths (N,).

    Returns (PackedAdj, node_ids (N,R) int32, n_nodes (N,) int32)."""
    _lib.require_cuda(tokens, lengths)
    tokens = tokens.to(torch.int32).contiguous()
    lengths = lengths.to(torch.int32).contiguous()
    n, r = tokens.shape
    w = (r + 63) // 64
    dev = tokens.device
    node_ids = torch.empty((n, r), device=dev, dtype=torch.int32)
    n_nodes = torch.empty((n,), device=dev, dtype=torch.int32)
    bits = torch.empty((n, r, w), device=dev, dtype=torch.int64)
    dinv = torch.empty((n, r), device=dev, dtype=torch.float32)
    call("gh_graph_build", ptr(tokens), ptr(lengths), n, r, int(window), ptr(node_ids), ptr(n_nodes), ptr(bits),
         ptr(dinv), stream())
    return PackedAdj(bits, dinv, None, None, n, r), node_ids, n_nodes


def as_packed(adj) -> PackedAdj:
    return adj if isinstance(adj, PackedAdj) else PackedAdj.from_dense(adj)


class RaggedPlan:
    """Node-compact row layout of ``n`` graphs padded to ``r`` nodes (include/get_hip.h): the real nodes of
    all graphs back to back (graph g at rows ``goff[g] .. goff[g+1]``), every padding node after them.

    The reference runs all n*r padded rows through every layer (graph_based_semantic_structure.py:99-107);
    padding nodes have no edges and are masked out of the word attention (:180), so only the first cell's
    forward and the scorer (whose padding scores compete in GSL's top-k, wrapper.py:216-219) need them.
    ``m_real`` (sum of node counts) must be known on the HOST: it sizes the launches."""

    __slots__ = ("n", "r", "m_real", "m_tot", "goff", "rowg", "src", "cids", "maskf")

    def __init__(self, n_nodes: torch.Tensor, node_ids: torch.Tensor, m_real: int):
        _lib.require_cuda(n_nodes, node_ids)
        n, r = node_ids.shape
        dev = node_ids.device
        self.n, self.r, self.m_real, self.m_tot = int(n), int(r), int(m_real), int(n * r)
        assert 0 <= self.m_real <= self.m_tot
        n_nodes = n_nodes.to(torch.int32).contiguous()
        node_ids = node_ids.to(torch.int32).contiguous()
        self.goff = torch.empty((n + 1,), device=dev, dtype=torch.int32)
        self.rowg = torch.empty((n * r,), device=dev, dtype=torch.int32)
        self.src = torch.empty((n * r,), device=dev, dtype=torch.int32)
        self.cids = torch.empty((n * r,), device=dev, dtype=torch.int32)
        self.maskf = torch.empty((n * r,), device=dev, dtype=torch.float32)     # (cids >= 1): the word attention's mask
        call("gh_ragged_plan", ptr(n_nodes), ptr(node_ids), n, r, ptr(self.goff), ptr(self.rowg), ptr(self.src),
             ptr(self.cids), ptr(self.maskf), stream())

    @classmethod
    def empty(cls, n: int, r: int, m_real: int, device) -> "RaggedPlan":
        """Buffers of a plan that a later gh_get_prepare / gh_ragged_plan call fills (get_amd.batch.NativeBatch)."""
        self = cls.__new__(cls)
        self.n, self.r, self.m_real, self.m_tot = int(n), int(r), int(m_real), int(n * r)
        assert 0 <= self.m_real <= self.m_tot
        i32 = lambda k: torch.empty((k,), device=device, dtype=torch.int32)
        self.goff, self.rowg, self.src, self.cids = i32(n + 1), i32(n * r), i32(n * r), i32(n * r)
        self.maskf = torch.empty((n * r,), device=device, dtype=torch.float32)
        return self

    def to_padded(self, x: torch.Tensor, rows: Optional[int] = None) -> torch.Tensor:
        """Compact rows (m, ...) -> padded (n, r, ...) with zeros where no compact row was given."""
        m = x.shape[0] if rows is None else rows
        out = torch.zeros((self.m_tot,) + tuple(x.shape[1:]), device=x.device, dtype=x.dtype)
        out.index_copy_(0, self.src[:m].long(), x[:m])
        return out.view(self.n, self.r, *x.shape[1:])

    def from_padded(self, x: torch.Tensor, rows: Optional[int] = None) -> torch.Tensor:
        """Padded (n, r, ...) -> compact rows (m_tot or `rows`, ...)."""
        m = self.m_tot if rows is None else rows
        return x.reshape(self.m_tot, *x.shape[2:]).index_select(0, self.src[:m].long())


# --------------------------------------------------------------------------- aggregation (a = A x)
class _Spmm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, adj: PackedAdj, plan):
        x = _f32(x)
        h = x.shape[-1]
        if plan is not None:
            assert x.dim() == 2 and x.shape[0] == plan.m_real, "node-compact spmm takes the (m_real, h) real rows"
        else:
            assert x.dim() == 3 and x.shape[0] == adj.n and x.shape[1] == adj.r
        y = torch.empty_like(x)
        call("gh_spmm", *adj._args(), *_plan_args(plan), ptr(x), ptr(y), adj.n, adj.r, h, 0, 0, stream())
        ctx.adj, ctx.plan = adj, plan
        return y

    @staticmethod
    def backward(ctx, g):
        g = _f32(g)
        dx = torch.empty_like(g)
        call("gh_spmm", *ctx.adj._args(), *_plan_args(ctx.plan), ptr(g), ptr(dx), ctx.adj.n, ctx.adj.r, g.shape[-1], 1, 0,
             stream())
        return dx, None, None


def _plan_args(plan):
    """(goff, m_real) of the C-ABI: NULL/0 selects the padded layout."""
    return (None, 0) if plan is None else (ptr(plan.goff), plan.m_real)


def spmm(adj: PackedAdj, x: torch.Tensor, plan: "RaggedPlan" = None) -> torch.Tensor:
    return _Spmm.apply(x, adj, plan)


# --------------------------------------------------------------------------- GGNN cell
class _GGNNCell(torch.autograd.Function):
    """Models/BiDAF/wrapper.py:188-208 (without the input dropout, applied by the caller)."""

    @staticmethod
    def forward(ctx, x, ids, adj: PackedAdj, w_p, w_z0, b_z0, w_z1, b_z1, w_r0, b_r0, w_r1, b_r1, w_h0, b_h0, w_h1,
                b_h1, drop_p=0.0, drop_seed=0, plan=None, rows=0, score=None):
        x = _f32(x)
        n, r = adj.n, adj.r
        h, din = w_p.shape
        m = n * r if plan is None else int(rows)      # rows the forward computes (node-compact: m_real or m_tot)
        mb = n * r if plan is None else plan.m_real    # rows the backward computes
        if plan is not None:
            assert plan.n == n and plan.r == r and plan.m_real <= m <= plan.m_tot
        if ids is not None:
            assert ids.dtype == torch.int32 and ids.numel() >= m and x.dim() == 2 and x.shape[1] == din
            ids = ids.contiguous()
        elif plan is not None:
            assert x.dim() == 2 and x.shape[1] == din and x.shape[0] >= m, "node-compact cell takes (rows, din) features"
        else:
            assert x.numel() == m * din, f"x has {x.numel()} elements, expected {m}x{din}"
        dev = x.device
        # forward products x.W^T take the weights as stored; the backward's dX = g.W takes the cached transposes
        ws = [_f32(w.detach()) for w in (w_p, w_z0, w_z1, w_r0, w_r1, w_h0, w_h1)]
        wts = [transposed(w) for w in (w_p, w_z0, w_z1, w_r0, w_r1, w_h0, w_h1)]
        bs = [_f32(t.detach()) for t in (b_z0, b_z1, b_r0, b_r1, b_h0, b_h1)]      # the epilogues add b?0 + b?1
        # bf16 STORAGE pipeline (BASELINE configs[4], _lib.set_gemm_mode("bf16")): activations and weights of the cell live
        # in HBM as bf16, the GEMMs run v_mfma_f32_16x16x32_bf16 with fp32 accumulation; the fp32 cell output is kept for the
        # consumers outside the cell.  Only for shapes the bf16 kernels take (16-byte rows, activation-sized launches).
        bf = bf16_cell_ok(din, h, m, mb)
        ctx.bf = bf
        if bf:
            wkeys = (w_p, w_z0, w_z1, w_r0, w_r1, w_h0, w_h1)
            if all(t.dtype == torch.float32 and t.is_contiguous() for t in wkeys):
                tw = [bf16_twins(t) for t in wkeys]          # persistent buffers, refreshed with the transposes in one launch
                ws_b, wts_b = tuple(t[0] for t in tw), tuple(t[1] for t in tw)
            else:
                ws_b = derived("cell_w_bf16", wkeys, lambda: tuple(t.to(torch.bfloat16) for t in ws))
                wts_b = derived("cell_wt_bf16", wkeys, lambda: tuple(t.to(torch.bfloat16) for t in wts))
            if ids is not None:
                xb = derived("table_bf16", (x,), lambda: x.detach().to(torch.bfloat16), frozen=not x.requires_grad)
            else:
                xb = getattr(x, "_gh_bf16", None)
                if xb is None or xb.shape != x.shape:
                    xb = x.detach().to(torch.bfloat16)
            bufb = torch.empty((7, m, h), device=dev, dtype=torch.bfloat16)
            xp, a, z, rr, rx, hh, outb = bufb.unbind(0)
            out = torch.empty((m, h), device=dev, dtype=torch.float32)
            sx = None
            sc = (None, None, 0.0, 0)
            if score is not None:
                sw, sp, sseed = score
                sx = torch.empty((m,), device=dev, dtype=torch.float32)
                sc = (ptr(_f32(sw.detach().reshape(-1))), ptr(sx), float(sp), int(sseed))
            call("gh_ggnn_cell_fwd_bf16", *adj._args(), *_plan_args(plan), m, ptr(xb), ptr(ids), n, r, din, h,
                 *[ptr(t) for t in ws_b], *[ptr(t) for t in bs],
                 ptr(xp), ptr(a), ptr(z), ptr(rr), ptr(rx), ptr(hh), ptr(outb), ptr(out), float(drop_p), int(drop_seed), *sc,
                 stream())
            ctx.adj, ctx.ids, ctx.dims, ctx.plan, ctx.rows = adj, ids, (n, r, din, h), plan, (m, mb)
            ctx.drop = (float(drop_p), int(drop_seed))
            ctx.params = (w_p, w_z0, b_z0, w_z1, b_z1, w_r0, b_r0, w_r1, b_r1, w_h0, b_h0, w_h1, b_h1)
            ctx.save_for_backward(xb, bufb, *wts_b)
            ctx.x_needs_grad = ctx.needs_input_grad[0]
            ctx.x_shape = tuple(x.shape)
            res = out.view(n, r, h) if plan is None else out
            res._gh_bf16 = outb.view(n, r, h) if plan is None else outb        # bf16 twin for a following bf16 cell
            if score is not None:
                ctx.mark_non_differentiable(sx)
                return res, sx
            return res
        buf = torch.empty((7, m, h), device=dev, dtype=torch.float32)
        xp, a, z, rr, rx, hh, out = buf.unbind(0)
        sx = None
        sc = (None, None, 0.0, 0)
        if score is not None:      # (scorer proj weight (h,), dropout p, seed): projection fused into the last epilogue
            sw, sp, sseed = score
            sx = torch.empty((m,), device=dev, dtype=torch.float32)
            sc = (ptr(_f32(sw.detach().reshape(-1))), ptr(sx), float(sp), int(sseed))
        call("gh_ggnn_cell_fwd", *adj._args(), *_plan_args(plan), m, ptr(x), ptr(ids), n, r, din, h,
             *[ptr(t) for t in ws], *[ptr(t) for t in bs],
             ptr(xp), ptr(a), ptr(z), ptr(rr), ptr(rx), ptr(hh), ptr(out), float(drop_p), int(drop_seed), *sc, stream())
        ctx.adj, ctx.ids, ctx.dims, ctx.plan, ctx.rows = adj, ids, (n, r, din, h), plan, (m, mb)
        ctx.drop = (float(drop_p), int(drop_seed))
        ctx.params = (w_p, w_z0, b_z0, w_z1, b_z1, w_r0, b_r0, w_r1, b_r1, w_h0, b_h0, w_h1, b_h1)
        ctx.save_for_backward(x, buf, *wts)
        ctx.x_needs_grad = ctx.needs_input_grad[0]
        res = out.view(n, r, h) if plan is None else out
        if score is not None:
            ctx.mark_non_differentiable(sx)
            return res, sx
        return res

    @staticmethod
    def backward(ctx, g, *_unused):
        x, buf, w_p, w_z0, w_z1, w_r0, w_r1, w_h0, w_h1 = ctx.saved_tensors      # the TRANSPOSED weights
        xp, a, z, rr, rx, hh, _ = buf.unbind(0)
        n, r, din, h = ctx.dims
        m_fwd, m = ctx.rows                   # the backward runs on the real-node rows only (m == m_fwd when padded)
        plan = ctx.plan
        dev = g.device
        g = _f32(g).reshape(m_fwd, h)
        _lib.ensure_workspace(dev)
        bf = getattr(ctx, "bf", False)
        scratch = torch.empty((5, max(m, 1), h), device=dev, dtype=torch.bfloat16 if bf else torch.float32)
        dhp, dzp, drp, dxp, da = scratch.unbind(0)
        ids = ctx.ids
        want_dx = ctx.x_needs_grad
        dx = None
        if want_dx:
            x_rows = m if ids is not None else x.shape[0] if plan is not None else m
            if bf and ids is None and plan is None:
                x_rows = m
            dx = torch.empty((x_rows, din), device=dev, dtype=torch.float32)
            if x_rows > m:
                dx[m:].zero_()               # padding rows (and unused tail rows of x) get no gradient
        P = ctx.params
        direct = all(_direct(p) for p in P)
        if direct:     # weight/bias gradients land in the parameters' own .grad (flat bucket) -- no adds, no fills
            pw_p, pz0, bz0, pz1, bz1, pr0, br0, pr1, br1, ph0, bh0, ph1, bh1 = (p.grad for p in P)
            gw = [pw_p, pz0, pz1, pr0, pr1, ph0, ph1]
            gb = [bz0, br0, bh0, bz1, br1, bh1]
        else:
            dw_p = torch.zeros((h, din), device=dev, dtype=torch.float32)
            dws = torch.zeros((6, h, h), device=dev, dtype=torch.float32)
            dbs = torch.zeros((3, h), device=dev, dtype=torch.float32)
            gw = [dw_p] + [dws[i] for i in range(6)]
            gb = [dbs[0], dbs[1], dbs[2], None, None, None]
        call("gh_ggnn_cell_bwd_bf16" if bf else "gh_ggnn_cell_bwd", *ctx.adj._args(), *_plan_args(plan), ptr(x), ptr(ids), n, r, din, h,
             ptr(w_p), ptr(w_z0), ptr(w_z1), ptr(w_r0), ptr(w_r1), ptr(w_h0), ptr(w_h1),
             ptr(xp), ptr(a), ptr(z), ptr(rr), ptr(rx), ptr(hh), ptr(g),
             ptr(dhp), ptr(dzp), ptr(drp), ptr(dxp), ptr(da),
             ptr(dx), *[ptr(t) for t in gw], *[ptr(t) for t in gb], ctx.drop[0], ctx.drop[1], stream())
        if want_dx:
            if ids is not None:      # trainable embedding table: scatter the row gradients
                demb = torch.zeros(x.shape, device=dev, dtype=torch.float32)
                demb.index_add_(0, ids[:m].long(), dx)
                dx = demb
            else:
                dx = dx.view(ctx.x_shape if bf else x.shape)
        if direct:
            return (dx, None, None) + (None,) * 18
        dz0, dz1, dr0, dr1, dh0, dh1 = dws.unbind(0)
        bz, br, bh = dbs.unbind(0)
        return (dx, None, None, dw_p, dz0, bz, dz1, bz, dr0, br, dr1, br, dh0, bh, dh1, bh, None, None, None, None, None)


def ggnn_cell(adj: PackedAdj, x, ids, params, drop_p: float = 0.0, drop_seed: int = 0, plan: "RaggedPlan" = None,
              rows: int = 0, score=None):
    """params: (w_p, w_z0, b_z0, w_z1, b_z1, w_r0, b_r0, w_r1, b_r1, w_h0, b_h0, w_h1, b_h1).
    drop_p > 0 applies the cell's input dropout inside the first GEMM (stateless hash mask keyed by drop_seed).
    plan: node-compact layout -- x is (>= rows, din) (or table + ids), the forward computes the first `rows`
    rows (plan.m_real or plan.m_tot), the backward the plan.m_real real-node rows; returns (rows, h).
    score = (w (h,), p, seed): also returns the GSL word scorer's projection dropout_p(out) . w per row, produced by
    the last GEMM's epilogue (see `scorer_fusable`); the result is then (out, score_x)."""
    if plan is not None and not rows:
        rows = plan.m_real
    return _GGNNCell.apply(x, ids, adj, *params, drop_p, drop_seed, plan, rows, score)


def bf16_cell_ok(din: int, h: int, m_fwd: int, m_bwd: int) -> bool:
    """True when a cell call takes the bf16 storage pipeline: mode "bf16", 16-byte bf16 rows, activation-sized launches."""
    return (_lib.gemm_mode() == "bf16" and din % 8 == 0 and h % 8 == 0 and din <= h and min(m_fwd, m_bwd) >= 8192)


def scorer_fusable(h: int) -> bool:
    """The scorer projection rides in the h-gate GEMM's epilogue when a whole output row lives in one workgroup."""
    return h % 4 == 0 and 4 <= h <= 320


def fused_dropout_ok(din: int, h: int) -> bool:
    """The in-kernel dropout lives in the float4 fast path of the GEMM."""
    return din % 4 == 0 and h % 4 == 0 and 4 <= din <= h


def new_dropout_seed() -> int:
    """Fresh 31-bit seed from torch's CPU generator (follows torch.manual_seed, no device sync)."""
    return int(torch.randint(0, 2 ** 31 - 1, (1,)).item())


def dropout_mask_reference(seed: int, rows: int, cols: int, p: float):
    """Host replica of the kernels' stateless mask (tests): bool (rows, cols), True = kept."""
    import numpy as np
    idx = (np.arange(rows, dtype=np.uint64)[:, None] * np.uint64(cols) + np.arange(cols, dtype=np.uint64)[None, :])
    x = (idx * np.uint64(0x9E3779B1) + np.uint64(seed)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x85EBCA6B)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(13)
    x = (x * np.uint64(0xC2B2AE35)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(16)
    t = p * 4294967296.0
    thresh = np.uint64(4294967295 if t >= 4294967295.0 else int(t))
    return x >= thresh


# --------------------------------------------------------------------------- word scorer + GSL (no gradient)
@torch.no_grad()
def scorer_gsl(adj: PackedAdj, feat: torch.Tensor, w_p, gate12: torch.Tensor, k: int, drop_p: float = 0.0,
               drop_seed: int = 0, plan: "RaggedPlan" = None, collapsed: bool = False, score_x: torch.Tensor = None):
    """GGNN(h->1) score of every node and the top-k keep set (wrapper.py:167-168, :215-219).
    feat: (N,R,H), or node-compact (N*R, H) incl. the padding rows when `plan` is given; alternatively
    score_x (rows,) = the projections proj(dropout(feat)) already produced by ggnn_cell(..., score=...).
    Returns (score (N,R) fp32, keep (N,W) int64 bit words) -- both in padded node indexing."""
    src = score_x if score_x is not None else feat
    src = _f32(src.detach())
    if plan is not None:
        need = min(plan.m_real + 1, plan.m_tot) if collapsed else plan.m_tot
        assert src.shape[0] == need and src.dim() == (1 if score_x is not None else 2), \
            "the scorer needs the padding rows too (they compete in top-k)"
        assert not (collapsed and drop_p > 0.0), "collapsed padding rows are an evaluation-mode layout"
        n, r = plan.n, plan.r
    else:
        n, r = adj.n, adj.r
        assert src.numel() == n * r * (1 if score_x is not None else src.shape[-1])
    h = w_p.numel()
    score = torch.empty((n, r), device=src.device, dtype=torch.float32)
    keep = torch.empty((n, adj.words), device=src.device, dtype=torch.int64)
    w_p = _f32(w_p.detach().reshape(-1))
    gate12 = _f32(gate12.detach())
    call("gh_scorer_gsl", ptr(adj.bits), ptr(adj.dinv), ptr(adj.vals), _plan_args(plan)[0], 1 if (collapsed and plan is not None) else 0,
         None if score_x is not None else ptr(src), ptr(src) if score_x is not None else None, ptr(w_p), ptr(gate12), n, r, h,
         int(k), ptr(score), ptr(keep), float(drop_p), int(drop_seed), stream())
    return score, keep


@torch.no_grad()
def gsl_topk(score: torch.Tensor, k: int) -> torch.Tensor:
    score = _f32(score.detach())
    n, r = score.shape
    keep = torch.empty((n, (r + 63) // 64), device=score.device, dtype=torch.int64)
    call("gh_gsl_topk", ptr(score), n, r, int(k), ptr(keep), stream())
    return keep


# --------------------------------------------------------------------------- concat attention
class _ConcatAtt(torch.autograd.Function):
    """two_branches_attention.py:121-148 (left given) / self_attention.py:75-100 (left None)."""

    @staticmethod
    def forward(ctx, left, right, mask, w1, w2, plan=None):
        right = _f32(right)
        if plan is not None:     # node-compact: right (m_real, dr), mask (m_real,), weights come back (m_real, heads)
            assert right.dim() == 2 and right.shape[0] == plan.m_real and mask.numel() == plan.m_real
            b, l, dr = plan.n, plan.r, right.shape[1]
            m = plan.m_real
        else:
            b, l, dr = right.shape
            m = b * l
        ha, inp = w1.shape
        heads = w2.shape[0]
        xl = 0
        if left is not None:
            left = _f32(left)
            xl = left.shape[1]
        assert inp == xl + dr, "linear1 input width must equal left + right widths"
        dev = right.device
        maskf = _f32(mask.to(torch.float32))
        w1c, w2c = _f32(w1.detach()), _f32(w2.detach())
        w1t = transposed(w1)          # for the backward's dX products
        u = torch.empty((b, ha), device=dev, dtype=torch.float32)
        t = torch.empty((m, ha), device=dev, dtype=torch.float32)
        e = torch.empty((m, heads), device=dev, dtype=torch.float32)
        weights = torch.empty((m, heads) if plan is not None else (b, l, heads), device=dev, dtype=torch.float32)
        attended = torch.empty((b, dr, heads), device=dev, dtype=torch.float32)
        pl = (None, None, 0) if plan is None else (ptr(plan.goff), ptr(plan.rowg), plan.m_real)
        call("gh_concat_att_fwd", ptr(left), ptr(right), ptr(maskf), *pl, b, l, xl, dr, ha, heads, ptr(w1c), ptr(w2c),
             ptr(u), ptr(t), ptr(e), ptr(weights), ptr(attended), stream())
        ctx.dims = (b, l, xl, dr, ha, heads, m)
        ctx.plan = plan
        ctx.params = (w1, w2)
        ctx.has_left = left is not None
        ctx.save_for_backward(left if left is not None else right.new_empty(0), right, w1t, w2c, t, weights)
        return attended, weights

    @staticmethod
    def backward(ctx, g_att, g_w):
        left, right, w1t, w2, t, weights = ctx.saved_tensors
        b, l, xl, dr, ha, heads, m = ctx.dims
        plan = ctx.plan
        dev = right.device
        if not ctx.has_left:
            left = None
        _lib.ensure_workspace(dev)
        g_att = _f32(g_att) if g_att is not None else torch.zeros((b, dr, heads), device=dev)
        g_w = _f32(g_w) if g_w is not None else None
        de = torch.empty((m, heads), device=dev, dtype=torch.float32)
        dpre = torch.empty((m, ha), device=dev, dtype=torch.float32)
        du = torch.empty((b, ha), device=dev, dtype=torch.float32)
        dleft = torch.empty((b, xl), device=dev, dtype=torch.float32) if left is not None else None
        dright = torch.empty_like(right)
        direct = all(_direct(p) for p in ctx.params)
        if direct:
            dw1, dw2 = ctx.params[0].grad, ctx.params[1].grad
        else:
            dw1 = torch.zeros((ha, xl + dr), device=dev, dtype=torch.float32)
            dw2 = torch.zeros((heads, ha), device=dev, dtype=torch.float32)
        pa = (ptr(left), ptr(right), *_plan_args(plan), b, l, xl, dr, ha, heads, ptr(w1t), ptr(w2), ptr(t), ptr(weights),
              ptr(g_att), ptr(g_w), ptr(de), ptr(dpre), ptr(du), ptr(dleft))
        if direct and WGRAD_SIDE_STREAM and b * l <= WGRAD_FEW_ROWS and right.is_cuda:
            # evidence level: everything but dW1 in line, dW1 = dpre^T [left | right] on the auxiliary stream
            call("gh_concat_att_bwd", *pa, ptr(dright), None, ptr(dw2), stream())
            _side_wgrad(dev, (left, right, dpre, du),
                        lambda: call("gh_concat_att_bwd", *pa, None, ptr(dw1), ptr(dw2), stream()))
        else:
            call("gh_concat_att_bwd", *pa, ptr(dright), ptr(dw1), ptr(dw2), stream())
        if direct:
            return dleft, dright, None, None, None, None
        return dleft, dright, None, dw1, dw2, None


def concat_att(left, right, mask, w1, w2, plan: "RaggedPlan" = None):
    """plan: node-compact `right` (m_real, dr) / `mask` (m_real,); weights are returned compact (m_real, heads)."""
    return _ConcatAtt.apply(left, right, mask, w1, w2, plan)


# --------------------------------------------------------------------------- linear
class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        x2 = _f32(x).reshape(-1, x.shape[-1])
        m, k = x2.shape
        n = w.shape[0]
        wc = _f32(w.detach())
        y = torch.empty((m, n), device=x.device, dtype=torch.float32)
        bc = _f32(b.detach()) if b is not None else None
        call("gh_linear_fwd", ptr(x2), ptr(wc), ptr(bc), ptr(y), m, k, n, stream())
        ctx.save_for_backward(x2, transposed(w), wc)
        ctx.params = (w, b)
        ctx.has_bias = b is not None
        ctx.xshape = x.shape
        return y.view(*x.shape[:-1], n)

    @staticmethod
    def backward(ctx, g):
        x2, wt, wc = ctx.saved_tensors
        m, k = x2.shape
        n = wt.shape[1]
        g2 = _f32(g).reshape(m, n)
        _lib.ensure_workspace(g.device)
        dx = torch.empty((m, k), device=g.device, dtype=torch.float32) if ctx.needs_input_grad[0] else None
        pw, pb = ctx.params
        direct = _direct(pw) and (pb is None or _direct(pb))
        if direct:
            dw, db = pw.grad, (pb.grad if pb is not None else None)
        else:
            dw = torch.zeros((n, k), device=g.device, dtype=torch.float32)
            db = torch.zeros((n,), device=g.device, dtype=torch.float32) if ctx.has_bias else None
        if direct and WGRAD_SIDE_STREAM and m <= WGRAD_FEW_ROWS and g2.is_cuda:
            if dx is not None:
                call("gh_linear_bwd", ptr(x2), ptr(wt), ptr(wc), ptr(g2), m, k, n, ptr(dx), None, None, stream())
            _side_wgrad(g2.device, (x2, g2),
                        lambda: call("gh_linear_bwd", ptr(x2), ptr(wt), ptr(wc), ptr(g2), m, k, n, None, ptr(dw), ptr(db), stream()))
        else:
            call("gh_linear_bwd", ptr(x2), ptr(wt), ptr(wc), ptr(g2), m, k, n, ptr(dx), ptr(dw), ptr(db), stream())
        dxo = dx.view(ctx.xshape) if dx is not None else None
        if direct:
            return dxo, None, None
        return dxo, dw, db


def linear(x, w, b=None):
    return _Linear.apply(x, w, b)


def linear_wgrad_bf16(g16: torch.Tensor, x16: torch.Tensor, dw: torch.Tensor, db: torch.Tensor = None):
    """dw[n][k] += g16^T x16, db[n] += colsum(g16) for bf16 activations / upstream gradients [m][n], [m][k] (row views with a
    leading dimension are taken as they are) and fp32 accumulators -- gh_linear_wgrad_bf16, the weight-gradient GEMM of the
    bf16 storage pipeline (gh_ggnn_cell_bwd_bf16) on its own.  Returns dw."""
    assert g16.dtype == torch.bfloat16 and x16.dtype == torch.bfloat16 and dw.dtype == torch.float32
    assert g16.dim() == 2 and x16.dim() == 2 and g16.shape[0] == x16.shape[0] and g16.stride(1) == 1 and x16.stride(1) == 1
    m, n = g16.shape
    k = x16.shape[1]
    assert tuple(dw.shape) == (n, k) and dw.stride(1) == 1 and (db is None or (db.dtype == torch.float32 and db.numel() == n and db.is_contiguous()))
    _lib.ensure_workspace(g16.device)
    # (row views with a leading dimension: the raw addresses, `ptr` insists on contiguous tensors)
    call("gh_linear_wgrad_bf16", g16.data_ptr(), g16.stride(0), x16.data_ptr(), x16.stride(0), m, n, k, dw.data_ptr(), dw.stride(0), ptr(db), stream())
    return dw


# --------------------------------------------------------------------------- ragged helpers
class Segments:
    """Claim -> evidence-pair segmentation of one batch (prefix sums live on the device)."""

    def __init__(self, counts: torch.Tensor, b1: int, n_max: int):
        _lib.require_cuda(counts)
        counts = counts.to(torch.int64).contiguous()
        self.b = counts.shape[0]
        self.b1 = int(b1)
        self.n_max = int(n_max)
        self.offsets = torch.empty((self.b + 1,), device=counts.device, dtype=torch.int32)
        self.pair2claim = torch.empty((max(self.b1, 1),), device=counts.device, dtype=torch.int32)
        self.has = torch.empty((self.b, 1), device=counts.device, dtype=torch.float32)      # 1.0 where the claim has evidences
        call("gh_seg_offsets", ptr(counts), self.b, ptr(self.offsets), ptr(self.pair2claim), self.b1, ptr(self.has), stream())


class _SegBroadcast(torch.autograd.Function):       # basic_fc_model.py:80-92 _pad_left_tensor
    @staticmethod
    def forward(ctx, src, seg: Segments):
        src = _f32(src)
        x = src.shape[1]
        dst = torch.empty((seg.b1, x), device=src.device, dtype=torch.float32)
        call("gh_seg_broadcast", ptr(src), ptr(seg.pair2claim), ptr(dst), seg.b1, x, stream())
        ctx.seg = seg
        return dst

    @staticmethod
    def backward(ctx, g):
        g = _f32(g)
        seg = ctx.seg
        x = g.shape[1]
        d = torch.empty((seg.b, x), device=g.device, dtype=torch.float32)
        call("gh_seg_sum", ptr(g), ptr(seg.offsets), ptr(d), seg.b, x, stream())
        return d, None


class _SegPad(torch.autograd.Function):             # basic_fc_model.py:94-121 _pad_right_tensor
    @staticmethod
    def forward(ctx, src, seg: Segments):
        src = _f32(src)
        x = src.shape[1]
        dst = torch.empty((seg.b, seg.n_max, x), device=src.device, dtype=torch.float32)
        call("gh_seg_pad", ptr(src), ptr(seg.offsets), ptr(dst), seg.b, seg.n_max, x, x, stream())
        ctx.seg = seg
        return dst

    @staticmethod
    def backward(ctx, g):
        g = _f32(g)
        seg = ctx.seg
        x = g.shape[2]
        d = torch.zeros((seg.b1, x), device=g.device, dtype=torch.float32)     # rows no slot maps to (counts > n_max) get 0
        call("gh_seg_unpad", ptr(g), ptr(seg.offsets), ptr(d), seg.b, seg.n_max, x, x, stream())
        return d, None


def seg_broadcast(src, seg):
    return _SegBroadcast.apply(src, seg)


def seg_pad(src, seg):
    return _SegPad.apply(src, seg)


class _EvdAssemble(torch.autograd.Function):
    """graph_based_semantic_structure.py:157-170,195-215: pad_right(avg) ++ article_source_embs(sources) and the slot mask."""

    @staticmethod
    def forward(ctx, avg, table, seg: Segments, sources, document):
        avg = _f32(avg)
        xa = avg.shape[1]
        b, n = seg.b, seg.n_max
        ds = 0
        tb = None
        if table is not None:
            tb = _f32(table.detach())
            ds = tb.shape[1]
            if sources.dtype not in (torch.int32, torch.int64):
                sources = sources.long()
            sources = sources.contiguous()
            assert sources.numel() == b * n
        if document.dtype not in (torch.int32, torch.int64):
            document = document.long()
        document = document.contiguous()
        r = document.shape[-1]
        assert document.numel() == b * n * r
        right = torch.empty((b, n, xa + ds), device=avg.device, dtype=torch.float32)
        mask = torch.empty((b, n), device=avg.device, dtype=torch.float32)
        call("gh_evd_assemble_fwd", ptr(avg), ptr(seg.offsets), ptr(tb), tb.shape[0] if tb is not None else 0,
             ptr(sources) if table is not None else None,
             1 if (table is not None and sources.dtype == torch.int64) else 0, ptr(document),
             1 if document.dtype == torch.int64 else 0, b, n, xa, ds, r, ptr(right), ptr(mask), stream())
        ctx.seg, ctx.dims, ctx.table = seg, (xa, ds), table
        ctx.sources = sources if table is not None else None
        ctx.mark_non_differentiable(mask)
        return right, mask

    @staticmethod
    def backward(ctx, g, _gm):
        seg = ctx.seg
        xa, ds = ctx.dims
        g = _f32(g)
        table = ctx.table
        # rows no slot maps to (a claim with more than n_max evidences, reachable through the dense compatibility shim) get 0
        d_avg = torch.zeros((seg.b1, xa), device=g.device, dtype=torch.float32) if ctx.needs_input_grad[0] else None
        d_table = None
        ret_table = None
        if table is not None and table.requires_grad:
            if _direct(table):
                d_table = table.grad
            else:
                d_table = torch.zeros_like(table, dtype=torch.float32)
                ret_table = d_table
        src = ctx.sources
        call("gh_evd_assemble_bwd", ptr(g), ptr(seg.offsets), ptr(src), 1 if (src is not None and src.dtype == torch.int64) else 0,
             table.shape[0] if table is not None else 0, seg.b, seg.n_max, xa, ds, ptr(d_avg), ptr(d_table), stream())      # ds is also g's row pitch: always the real width
        return d_avg, ret_table, None, None, None


def evd_assemble(avg, table, seg, sources, document):
    """-> (right (B, n, Xa + Ds), mask (B, n) float).  table: the article-source embedding weight or None."""
    return _EvdAssemble.apply(avg, table, seg, sources, document)


class _MaskedMean(torch.autograd.Function):         # graph_based_semantic_structure.py:145,153
    @staticmethod
    def forward(ctx, hid, ids, lens):
        hid = _f32(hid)
        b, l, h = hid.shape
        lens = _f32(lens.to(torch.float32))
        ids = ids.to(torch.int32).contiguous()
        dst = torch.empty((b, h), device=hid.device, dtype=torch.float32)
        call("gh_masked_mean_fwd", ptr(hid), ptr(ids), ptr(lens), ptr(dst), b, l, h, stream())
        ctx.save_for_backward(ids, lens)
        ctx.dims = (b, l, h)
        return dst

    @staticmethod
    def backward(ctx, g):
        ids, lens = ctx.saved_tensors
        b, l, h = ctx.dims
        g = _f32(g)
        d = torch.empty((b, l, h), device=g.device, dtype=torch.float32)
        call("gh_masked_mean_bwd", ptr(g), ptr(ids), ptr(lens), ptr(d), b, l, h, stream())
        return d, None, None


def masked_mean(hid, ids, lens):
    return _MaskedMean.apply(hid, ids, lens)


def cross_entropy(phi, labels):
    """Mean cross-entropy (losses.py:29-32) with its gradient in one launch; see get_amd.fused.cross_entropy."""
    from .fused import cross_entropy as _ce
    return _ce(phi, labels)


def backward(loss):
    """loss.backward() without autograd's root fill + scale launches (get_amd.fused.backward)."""
    from .fused import backward as _bw
    _bw(loss)


# --------------------------------------------------------------------------- optimiser
def adam_step_flat(p, g, m, v, step, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-3, grad_scale=1.0):
    """One Adam step on flat fp32 buffers (declare_fitter.py:58-61 semantics)."""
    call("gh_adam_step", ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), float(lr), float(betas[0]), float(betas[1]),
         float(eps), float(weight_decay), int(step), float(grad_scale), stream())
    _bump_trainer_epoch()
