"""The whole GET forward / backward as ONE library call each (include/get_hip.h: gh_get_forward / gh_get_backward).

``Graph_basedSemantiStructure.forward`` routes through :func:`forward` whenever the model/batch qualify (frozen word
table, float4-shaped widths with d <= h <= 1024 -- every shape BASELINE.json names, in every arithmetic mode incl. the
bf16 storage pipeline of configs[4]); anything else keeps the module-by-module path of get_amd/modules.py + get_amd/ops.py, which stays the API for callers that use
the reference's modules one by one.  Same kernels underneath; what disappears is ~120 Python -> ctypes -> autograd round
trips per training step (~2.2 ms of host time, the bound of the realistic-evidence-count regime) and the at::native
glue between them.
"""
from __future__ import annotations

import ctypes
import os
import weakref

import torch

from . import _lib, ops
from .keywords import KeyWordSettings as K
from .ops import PackedAdj

ENABLED = os.environ.get("GET_AMD_FUSED", "1") != "0"

_P, _I, _F, _U, _L = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_uint32, ctypes.c_int64

_CELL_W = ["w_p", "w_z0", "w_z1", "w_r0", "w_r1", "w_h0", "w_h1"]
_CELL_B = ["b_z0", "b_z1", "b_r0", "b_r1", "b_h0", "b_h1"]


class CellParams(ctypes.Structure):
    _fields_ = ([(n, _P) for n in _CELL_W] + [(n, _P) for n in _CELL_B] + [("wt" + n[1:], _P) for n in _CELL_W] +
                [("d" + n, _P) for n in _CELL_W] + [("d" + n, _P) for n in _CELL_B])


class CellBf16(ctypes.Structure):
    _fields_ = [(n, _P) for n in _CELL_W] + [("wt" + n[1:], _P) for n in _CELL_W]


class AttParams(ctypes.Structure):
    _fields_ = [("w1", _P), ("w2", _P), ("w1t", _P), ("dw1", _P), ("dw2", _P)]


class GetModel(ctypes.Structure):
    _fields_ = [("d", _I), ("h", _I), ("word_heads", _I), ("evd_heads", _I), ("n_classes", _I),
                ("claim_src_dim", _I), ("article_src_dim", _I), ("claim_src_rows", _I), ("article_src_rows", _I),
                ("embedding", _P),
                ("claim", CellParams), ("cell1", CellParams), ("cell2", CellParams),
                ("scorer_w", _P), ("scorer_gate", _P),
                ("att_word", AttParams), ("att_evd", AttParams),
                ("claim_src_table", _P), ("article_src_table", _P),
                ("d_claim_src_table", _P), ("d_article_src_table", _P),
                ("out0_w", _P), ("out0_b", _P), ("out0_wt", _P),
                ("out1_w", _P), ("out1_b", _P), ("out1_wt", _P),
                ("d_out0_w", _P), ("d_out0_b", _P), ("d_out1_w", _P), ("d_out1_b", _P),
                ("storage", _I), ("embedding16", _P), ("cell1_16", CellBf16), ("cell2_16", CellBf16),
                ("att_word_w1_16", _P), ("att_word_w1t_16", _P)]


class GetBatch(ctypes.Structure):
    _fields_ = [("b", _I), ("b1", _I), ("l", _I), ("r", _I), ("n_max", _I),
                ("m_real", _I), ("collapsed", _I), ("k_keep", _I),
                ("q_ids", _P), ("q_lens", _P), ("q_lens_kind", _I),
                ("q_bits", _P), ("q_dinv", _P), ("q_vals", _P),
                ("d_ids", _P), ("d_bits", _P), ("d_dinv", _P), ("d_vals", _P),
                ("goff", _P), ("rowg", _P), ("cids", _P), ("maskf", _P),
                ("counts", _P), ("counts_fit", _I),
                ("doc_sources", _P), ("doc_sources_i64", _I),
                ("query_sources", _P), ("query_sources_i64", _I),
                ("document", _P), ("document_i64", _I),
                ("drop_claim", _F), ("drop_gnn", _F),
                ("seed_claim", _U), ("seed_cell1", _U), ("seed_scorer", _U), ("seed_cell2", _U)]


class GetPlan(ctypes.Structure):
    _fields_ = [("fwd_floats", _L), ("bwd_floats", _L), ("obs_floats", _L), ("phi", _L), ("word_w", _L), ("evd_w", _L), ("score", _L),
                ("keep", _L)]


def _cell_tensors(cell):
    """(weights[7], biases[6]) of a modules.GGNN in the C struct's order."""
    g = lambda m: m.linear
    ws = [cell.proj.linear.weight, g(cell.linearz0).weight, g(cell.linearz1).weight, g(cell.linearr0).weight,
          g(cell.linearr1).weight, g(cell.linearh0).weight, g(cell.linearh1).weight]
    bs = [g(cell.linearz0).bias, g(cell.linearz1).bias, g(cell.linearr0).bias, g(cell.linearr1).bias,
          g(cell.linearh0).bias, g(cell.linearh1).bias]
    return ws, bs


class Binding:
    """Everything of one model the composite calls need, in the C struct's order: the parameter tensors (weights as stored),
    their cached transposes, and where the gradients go.  Rebuilt only when a pointer moved (checked per call with one
    tuple comparison of the data pointers)."""

    def __init__(self, model):
        self._model = weakref.ref(model)      # (the model owns the binding: no reference cycle around the arena pool's buffers)
        gw = model.ggnn_with_gsl
        self.cells = [model.ggnn4claim_1, gw.feat_prop1, gw.feat_prop2]
        self.params = []          # live parameters in a fixed order (gradient outputs follow the same order)
        self.mats = []            # those that need a transpose in the backward
        for c in self.cells:
            ws, bs = _cell_tensors(c)
            self.params += ws + bs
            self.mats += ws
        aw, ae = model.self_att_word, model.self_att_evd
        self.params += [aw.linear1.weight, aw.linear2.weight, ae.linear1.weight, ae.linear2.weight]
        self.mats += [aw.linear1.weight, ae.linear1.weight]
        self.params += [model.out[0].weight, model.out[0].bias, model.out[1].weight, model.out[1].bias]
        self.mats += [model.out[0].weight, model.out[1].weight]
        self.tables = []
        if model.use_claim_source:
            self.tables.append(model.claim_source_embs.weight)
        if model.use_article_source:
            self.tables.append(model.article_source_embs.weight)
        self.struct = {False: None, True: None}      # cached GetModel per need_grads
        self._sig = {False: None, True: None}
        self._mat_ids = [id(m) for m in self.mats]
        self._checked = False
        self.pool = _ArenaPool()

    @property
    def model(self):
        return self._model()

    def eligible(self) -> bool:
        m = self.model
        d = m.embedding.weight.shape[1]
        h = m.hidden_size
        cs = m.claim_emb_size if m.use_claim_source else 0
        as_ = m.article_emb_size if m.use_article_source else 0
        return (not m.embedding.weight.requires_grad and d % 4 == 0 and h % 4 == 0 and 4 <= d <= h <= 1024 and
                1 <= m.num_att_heads_for_words <= 8 and 1 <= m.num_att_heads_for_evds <= 8 and cs % 4 == 0 and as_ % 4 == 0 and
                m.ggnn4claim_1.proj.linear.weight.shape == (h, d) and m.out[1].weight.shape[1] == h and
                all(p.dtype == torch.float32 and p.is_contiguous() for p in self.params))

    def get(self, need_grads: bool):
        """-> (GetModel struct, direct, grad_buffer or None, views or None).  direct: the kernels accumulate straight into the
        parameters' .grad (FlatTrainer bucket); otherwise a zeroed flat buffer receives them and `views` maps id(parameter)
        to its slice (returned to autograd)."""
        if not self._checked:          # the ctypes mirrors must have the C structs' sizes
            sz = (ctypes.c_int64 * 5)()
            _lib.call("gh_get_struct_sizes", ctypes.cast(sz, ctypes.c_void_p))
            mine = (ctypes.sizeof(GetModel), ctypes.sizeof(GetBatch), ctypes.sizeof(GetPlan), ctypes.sizeof(CellParams),
                    ctypes.sizeof(CellBf16))
            if tuple(sz) != mine:
                raise RuntimeError(f"get_amd: struct layout mismatch between fused.py {mine} and libget_hip.so {tuple(sz)}")
            self._checked = True
        m = self.model
        direct = need_grads and all(ops._direct(p) for p in self.params) and all((not t.requires_grad) or ops._direct(t) for t in self.tables)
        wts = [ops.transposed(w) for w in self.mats] if need_grads else None
        gscorer = m.ggnn_with_gsl._gate12()
        # bf16 storage mode (BASELINE configs[4]): bf16 twins of the two evidence cells' matrices (persistent buffers, refreshed
        # with the transposes after the optimiser step) and of the frozen word table
        bf = _lib.gemm_mode() == "bf16" and S_bf16_shapes(m)
        twins = emb16 = None
        if bf:
            twins = [[ops.bf16_twins(w) for w in _cell_tensors(c)[0]] for c in self.cells[1:]]
            twins.append([ops.bf16_twins(m.self_att_word.linear1.weight)])
            emb = m.embedding.weight
            emb16 = ops.derived("table_bf16", (emb,), lambda: emb.detach().to(torch.bfloat16), frozen=not emb.requires_grad)
        sig = (tuple(p.data_ptr() for p in self.params), tuple(t.data_ptr() for t in self.tables),
               (emb16.data_ptr(), tuple(t.data_ptr() for cell in twins for pair in cell for t in pair)) if bf else None,
               tuple(t.data_ptr() for t in wts) if wts else None,
               tuple(p.grad.data_ptr() for p in self.params) if direct else None,
               tuple((t.grad.data_ptr() if t.requires_grad else 0) for t in self.tables) if direct else None,
               m.embedding.weight.data_ptr(), gscorer.data_ptr(), m.ggnn_with_gsl.word_scorer1.proj.linear.weight.data_ptr())
        gbuf = views = None
        if need_grads and not direct:
            sizes = [(p.numel() + 63) // 64 * 64 for p in self.params] + [(t.numel() + 63) // 64 * 64 for t in self.tables]
            gbuf = torch.zeros(sum(sizes), device=self.params[0].device, dtype=torch.float32)
            views, off = {}, 0
            for t, sz in zip(self.params + self.tables, sizes):
                views[id(t)] = gbuf[off:off + t.numel()].view_as(t)
                off += sz
        if self.struct[need_grads] is not None and sig == self._sig[need_grads] and gbuf is None:
            return self.struct[need_grads], direct, None, None
        S = GetModel()
        S.d, S.h = m.embedding.weight.shape[1], m.hidden_size
        S.word_heads, S.evd_heads, S.n_classes = m.num_att_heads_for_words, m.num_att_heads_for_evds, m.out[1].weight.shape[0]
        S.claim_src_dim = m.claim_emb_size if m.use_claim_source else 0
        S.article_src_dim = m.article_emb_size if m.use_article_source else 0
        S.claim_src_rows = m.claim_source_embs.weight.shape[0] if m.use_claim_source else 0
        S.article_src_rows = m.article_source_embs.weight.shape[0] if m.use_article_source else 0
        S.embedding = m.embedding.weight.data_ptr()
        S.storage = 1 if bf else 0
        if bf:
            S.embedding16 = emb16.data_ptr()
            S.att_word_w1_16, S.att_word_w1t_16 = twins[2][0][0].data_ptr(), twins[2][0][1].data_ptr()
            for c16, cell in zip((S.cell1_16, S.cell2_16), twins[:2]):
                for name, (w16, wt16) in zip(_CELL_W, cell):
                    setattr(c16, name, w16.data_ptr())
                    setattr(c16, "wt" + name[1:], wt16.data_ptr())
        gptr = None
        if need_grads:
            gl = [p.grad for p in self.params] if direct else [views[id(p)] for p in self.params]
            gptr = [g.data_ptr() for g in gl]
        wt_of = {i: t.data_ptr() for i, t in zip(self._mat_ids, wts)} if wts else {}
        k = 0
        for cs_, cell in zip((S.claim, S.cell1, S.cell2), self.cells):
            ws, bs = _cell_tensors(cell)
            for name, t in zip(_CELL_W, ws):
                setattr(cs_, name, t.data_ptr())
                if need_grads:
                    setattr(cs_, "wt" + name[1:], wt_of[id(t)])
                    setattr(cs_, "d" + name, gptr[k])
                k += 1
            for name, t in zip(_CELL_B, bs):
                setattr(cs_, name, t.data_ptr())
                if need_grads:
                    setattr(cs_, "d" + name, gptr[k])
                k += 1
        S.scorer_w = m.ggnn_with_gsl.word_scorer1.proj.linear.weight.data_ptr()
        S.scorer_gate = gscorer.data_ptr()
        self._keep_alive = (gscorer, wts, twins, emb16)
        for a_, att in zip((S.att_word, S.att_evd), (m.self_att_word, m.self_att_evd)):
            a_.w1, a_.w2 = att.linear1.weight.data_ptr(), att.linear2.weight.data_ptr()
            if need_grads:
                a_.w1t = wt_of[id(att.linear1.weight)]
                a_.dw1, a_.dw2 = gptr[k], gptr[k + 1]
            k += 2
        S.out0_w, S.out0_b = m.out[0].weight.data_ptr(), m.out[0].bias.data_ptr()
        S.out1_w, S.out1_b = m.out[1].weight.data_ptr(), m.out[1].bias.data_ptr()
        if need_grads:
            S.out0_wt, S.out1_wt = wt_of[id(m.out[0].weight)], wt_of[id(m.out[1].weight)]
            S.d_out0_w, S.d_out0_b, S.d_out1_w, S.d_out1_b = gptr[k], gptr[k + 1], gptr[k + 2], gptr[k + 3]
        k += 4
        for use, fld, tab in ((m.use_claim_source, "claim_src_table", getattr(m, "claim_source_embs", None)),
                              (m.use_article_source, "article_src_table", getattr(m, "article_source_embs", None))):
            if not use:
                continue
            w = tab.weight
            setattr(S, fld, w.data_ptr())
            if need_grads and w.requires_grad:
                setattr(S, "d_" + fld, w.grad.data_ptr() if direct else views[id(w)].data_ptr())
        if gbuf is None:
            self.struct[need_grads], self._sig[need_grads] = S, sig
        return S, direct, gbuf, views


def S_bf16_shapes(m) -> bool:
    """Widths the bf16 storage cells take (16-byte bf16 rows); the row-count half of the rule (>= 8192 real node rows) is
    checked per batch inside the library (model_ops.hip make_dims), as ops.bf16_cell_ok does on the per-module path."""
    return m.embedding.weight.shape[1] % 8 == 0 and m.hidden_size % 8 == 0


def _binding(model) -> Binding:
    b = model.__dict__.get("_gh_binding")
    if b is None:
        b = Binding(model)
        model.__dict__["_gh_binding"] = b
    return b


def _i32(t):
    return t if t.dtype == torch.int32 and t.is_contiguous() else t.to(torch.int32).contiguous()


class _Prepared:
    """One forward's batch descriptor plus the tensors it points into (kept alive until the backward has run)."""
    __slots__ = ("struct", "keep", "b", "b1", "r", "plan", "n_max", "hw", "he", "rows", "w", "c")


def _prepare(model, query, document, kargs, q_adj: PackedAdj, d_adj: PackedAdj, plan, doc):
    m = model
    B, L = query.shape
    n_max = int(kargs[K.FIXED_NUM_EVIDENCES])
    b1, R = doc.shape
    S = GetBatch()
    S.b, S.b1, S.l, S.r, S.n_max = B, b1, L, R, n_max
    training = m.training
    S.collapsed = 0
    if plan is not None:
        S.m_real = plan.m_real
        S.collapsed = 0 if training else 1
        S.goff, S.rowg, S.cids, S.maskf = plan.goff.data_ptr(), plan.rowg.data_ptr(), plan.cids.data_ptr(), plan.maskf.data_ptr()
        rows = plan.m_real
    else:
        S.m_real = -1
        rows = b1 * R
    S.k_keep = int(m.ggnn_with_gsl.gsl1.rate * R)
    q_ids = _i32(query)
    d_ids = _i32(doc)
    q_lens = kargs[K.Query_lens]
    if q_lens.dtype not in (torch.float32, torch.int32, torch.int64):
        q_lens = q_lens.float()
    q_lens = q_lens.contiguous()
    counts = kargs[K.EvidenceCountPerQuery]
    fit = getattr(counts, "_gh_fit", None) == counts._version      # NativeBatch's promise, void after any in-place edit
    if counts.dtype != torch.int64 or not counts.is_contiguous():
        counts = counts.to(torch.int64).contiguous()
    S.q_ids, S.q_lens = q_ids.data_ptr(), q_lens.data_ptr()
    S.q_lens_kind = {torch.float32: 0, torch.int32: 1, torch.int64: 2}[q_lens.dtype]
    S.q_bits = q_adj.bits.data_ptr()
    S.q_dinv = q_adj.dinv.data_ptr() if q_adj.dinv is not None else None
    S.q_vals = q_adj.vals.data_ptr() if q_adj.vals is not None else None
    S.d_ids = d_ids.data_ptr()
    S.d_bits = d_adj.bits.data_ptr()
    S.d_dinv = d_adj.dinv.data_ptr() if d_adj.dinv is not None else None
    S.d_vals = d_adj.vals.data_ptr() if d_adj.vals is not None else None
    S.counts, S.counts_fit = counts.data_ptr(), 1 if fit else 0
    keep = [q_ids, d_ids, q_lens, counts, q_adj, d_adj, plan]
    if m.use_article_source:
        src = kargs[K.DocSources]
        if src.dtype not in (torch.int32, torch.int64):
            src = src.long()
        src = src.contiguous()
        assert src.numel() == B * n_max
        S.doc_sources, S.doc_sources_i64 = src.data_ptr(), 1 if src.dtype == torch.int64 else 0
        keep.append(src)
    if m.use_claim_source:
        qs = kargs[K.QuerySources]
        if qs.dtype not in (torch.int32, torch.int64):
            qs = qs.long()
        qs = qs.contiguous()
        assert qs.numel() == B
        S.query_sources, S.query_sources_i64 = qs.data_ptr(), 1 if qs.dtype == torch.int64 else 0
        keep.append(qs)
    dc = document
    if dc.dtype not in (torch.int32, torch.int64):
        dc = dc.long()
    dc = dc.contiguous()
    assert dc.numel() == B * n_max * R
    S.document, S.document_i64 = dc.data_ptr(), 1 if dc.dtype == torch.int64 else 0
    keep.append(dc)
    p_claim = float(m.ggnn4claim_1.dropout.p) if (training and hasattr(m.ggnn4claim_1, "dropout")) else 0.0
    fp1 = m.ggnn_with_gsl.feat_prop1
    p_gnn = float(fp1.dropout.p) if (training and hasattr(fp1, "dropout")) else 0.0
    S.drop_claim, S.drop_gnn = p_claim, p_gnn
    if p_claim > 0.0 or p_gnn > 0.0:
        seeds = torch.randint(0, 2 ** 31 - 1, (4,)).tolist()      # torch's CPU generator (follows torch.manual_seed)
        S.seed_claim, S.seed_cell1, S.seed_scorer, S.seed_cell2 = seeds
    P = _Prepared()
    P.struct, P.keep, P.b, P.b1, P.r, P.plan, P.n_max = S, keep, B, b1, R, plan, n_max
    P.hw, P.he, P.rows, P.w, P.c = m.num_att_heads_for_words, m.num_att_heads_for_evds, rows, (R + 63) // 64, m.out[1].weight.shape[0]
    return P


def _arena_floats(n: int) -> int:
    """Arena sizes in coarse classes (multiples of 1/16 of the enclosing power of two, at least 4 Mi floats: at most 1/8 above the request): the node count differs from
    batch to batch, and a caching allocator that sees a new multi-GB size every step keeps returning blocks to the driver
    and asking for new ones (measured: the first 20-step block of a B = 256 run took 3.0 s instead of 0.23 s)."""
    n = int(n)
    if n <= (1 << 22):
        return n
    g = 1 << max(22, n.bit_length() - 4)
    return (n + g - 1) // g * g


class _ArenaPool:
    """Persistent activation / scratch arenas of one model (per device), for forwards that will be followed by a backward.
    A training step used to allocate its two multi-GB arenas from torch's caching allocator in size classes; batches whose
    node counts fall into different classes made the allocator hand blocks back to the driver and ask for new ones for many
    steps (the B = 256 strong-scaling leg of round 4 needed 40 steps to settle, 90 -> 42 ms per step).  take() hands out the
    smallest free buffer that holds the request and is at most SLACK times larger; otherwise it allocates in the coarse size
    class and drops the free buffers the new one supersedes (those less than SLACK times smaller), so the pool converges to
    one buffer per kind (forward arena, backward scratch) sized for the largest batch seen.  A buffer comes back when its
    backward has been issued: gh_get_backward joins its side stream into the caller's stream, so the next forward's writes
    are ordered behind the backward's reads.  A forward whose graph is dropped without a backward never returns its buffer
    (the tensor dies with the graph); no-grad forwards do not use the pool at all (evaluation keeps nothing resident)."""
    MAX_FREE = 4
    SLACK = 1.30

    def __init__(self):
        self.free = {}          # device index -> list of tensors
        self.side = {}          # device index -> the side stream that may still read / write a pooled buffer

    def note_side(self, dev, side):
        if side is not None:
            dev = torch.device(dev)
            self.side[(dev.type, dev.index if dev.index is not None else torch.cuda.current_device())] = side

    def _release(self, t: torch.Tensor):
        """A buffer leaves the pool for good (superseded, evicted, cleared): it goes back to the caching allocator, which may hand it
        to another stream -- tell the allocator about the side stream's use first (pooled buffers skip record_stream while pooled)."""
        s_ = self.side.get((t.device.type, t.device.index))
        if s_ is not None:
            t.record_stream(s_)

    def _list(self, dev):
        dev = torch.device(dev)
        idx = dev.index if (dev.index is not None or dev.type != "cuda") else torch.cuda.current_device()
        return self.free.setdefault((dev.type, idx), [])

    def take(self, n: int, dev) -> torch.Tensor:
        lst = self._list(dev)
        best = None
        for i, t in enumerate(lst):
            if n <= t.numel() <= max(n * self.SLACK, n + (1 << 22)) and (best is None or t.numel() < lst[best].numel()):
                best = i
        if best is not None:
            return lst.pop(best)
        size = _arena_floats(n)
        for t in lst:
            if size / self.SLACK <= t.numel() < size:
                self._release(t)
        lst[:] = [t for t in lst if not (size / self.SLACK <= t.numel() < size)]
        return torch.empty(size, device=dev, dtype=torch.float32)

    def give(self, t: torch.Tensor):
        lst = self._list(t.device)
        lst.append(t)
        while len(lst) > self.MAX_FREE:
            self._release(lst.pop(min(range(len(lst)), key=lambda i: lst[i].numel())))

    def clear(self):
        """Hand every pooled buffer back to torch's caching allocator (callers that need the memory: `model_pool(model).clear()`,
        then torch.cuda.empty_cache())."""
        for lst in self.free.values():
            for t in lst:
                self._release(t)
        self.free.clear()


POOL_ARENAS = os.environ.get("GET_AMD_ARENA_POOL", "1") != "0"


class _GetFused(torch.autograd.Function):
    """graph_based_semantic_structure.py:76-125 in one forward and one backward library call."""

    @staticmethod
    def forward(ctx, binding: Binding, prep: _Prepared, side, *anchors):
        dev = anchors[0].device
        M, _, _, _ = binding.get(False)
        plan = GetPlan()
        _lib.call("gh_get_plan_buffers", ctypes.addressof(M), ctypes.addressof(prep.struct), ctypes.addressof(plan))
        pool = binding.pool if (POOL_ARENAS and any(ctx.needs_input_grad)) else None
        if pool is None and binding.pool.free:
            binding.pool.clear()          # a no-grad forward (validation pass): the training arenas go back to the allocator
        arena = pool.take(plan.fwd_floats, dev) if pool is not None else torch.empty(_arena_floats(plan.fwd_floats), device=dev, dtype=torch.float32)
        # the observables (logits, attention weights, scores, keep-sets: a few MB) get a buffer of their own: whoever keeps
        # them -- the model's last_score / last_keep, batched_predict's per-chunk lists -- does not pin the activation arena
        obs = torch.empty(int(plan.obs_floats), device=dev, dtype=torch.float32)
        main = _lib.stream()
        side_raw = side.cuda_stream if side is not None else main
        # split-K scratch of the few-row products and the per-block head-score partials of wide hidden layers (one per stream)
        _lib.ensure_workspace(dev)
        if side is not None and not _lib.has_workspace(dev, side_raw):
            with torch.cuda.stream(side):
                _lib.ensure_workspace(dev)
        if side is not None:
            if pool is None:
                arena.record_stream(side)      # (pooled arenas never go back to the allocator while work is in flight)
            else:
                pool.note_side(dev, side)      # (... and tell the allocator about the side stream when they finally do: _ArenaPool._release)
            obs.record_stream(side)
        _lib.call("gh_get_forward", ctypes.addressof(M), ctypes.addressof(prep.struct), arena.data_ptr(), obs.data_ptr(), main, side_raw)
        B, b1, R = prep.b, prep.b1, prep.r
        phi = obs[plan.phi:plan.phi + B * prep.c].view(B, prep.c)
        word_w = obs[plan.word_w:plan.word_w + prep.rows * prep.hw].view(prep.rows, prep.hw)
        evd_w = obs[plan.evd_w:plan.evd_w + B * prep.n_max * prep.he].view(B, prep.n_max, prep.he)
        score = obs[plan.score:plan.score + b1 * R].view(b1, R)
        keep = obs[plan.keep:plan.keep + b1 * prep.w * 2].view(torch.int64).view(b1, prep.w)
        ctx.binding, ctx.prep, ctx.side, ctx.arena, ctx.obs, ctx.plan_bwd = binding, prep, side, arena, obs, int(plan.bwd_floats)
        ctx.anchor_ids = [id(a) for a in anchors]
        ctx.mark_non_differentiable(score, keep)
        ctx.set_materialize_grads(False)      # no zero-filled gradient tensors for unused outputs (fill launches per step)
        ctx.pool = pool
        return phi, word_w, evd_w, score, keep

    @staticmethod
    def backward(ctx, g_phi, g_word_w=None, g_evd_w=None, *_unused):
        binding, prep, side, arena, obs = ctx.binding, ctx.prep, ctx.side, ctx.arena, ctx.obs
        if arena is None:
            raise RuntimeError("get_amd: the fused GET backward ran twice on one forward (retain_graph / autograd.grad followed "
                               "by backward): the activation arena is released by the first backward.  Run the forward again, or "
                               "set GET_AMD_FUSED=0 for the module-by-module path, which supports retain_graph.")
        dev = arena.device
        if g_phi is None and g_word_w is None and g_evd_w is None:          # nothing upstream depends on the outputs
            ctx.arena = None
            return (None, None, None) + (None,) * len(ctx.anchor_ids)
        if g_phi is None:
            g_phi = torch.zeros((prep.b, prep.c), device=dev, dtype=torch.float32)
        g_phi = ops._f32(g_phi)
        g_word_w = ops._f32(g_word_w) if g_word_w is not None else None      # loss terms on the attention weights
        g_evd_w = ops._f32(g_evd_w) if g_evd_w is not None else None
        M, direct, gbuf, views = binding.get(True)
        main = _lib.stream()
        side_raw = side.cuda_stream if side is not None else main
        _lib.ensure_workspace(dev)
        if side is not None and not _lib.has_workspace(dev, side_raw):
            with torch.cuda.stream(side):
                _lib.ensure_workspace(dev)
        pool = ctx.pool
        work = pool.take(ctx.plan_bwd, dev) if pool is not None else torch.empty(_arena_floats(ctx.plan_bwd), device=dev, dtype=torch.float32)
        if side is not None:
            for t in ((None if pool is not None else work), g_phi, gbuf, g_word_w, g_evd_w):
                if t is not None:
                    t.record_stream(side)
        args = (ctypes.addressof(M), ctypes.addressof(prep.struct), arena.data_ptr(), obs.data_ptr(), work.data_ptr(), g_phi.data_ptr(),
                g_word_w.data_ptr() if g_word_w is not None else None, g_evd_w.data_ptr() if g_evd_w is not None else None)
        hook = binding.model.ggnn_with_gsl.grad_milestone_hook
        if hook is None:
            _lib.call("gh_get_backward", *args, 0, main, side_raw)
        else:
            # data-parallel overlap (dist.FlatTrainer.attach_overlap): every gradient outside the first evidence cell and
            # the claim branch is final in stream order after phase 1 -- the early all-reduce starts underneath phase 2
            _lib.call("gh_get_backward", *args, 1, main, side_raw)
            if side is not None:
                ops._side_pending.add(dev.index if dev.index is not None else torch.cuda.current_device())
            try:
                hook()
            except BaseException:
                # phase 1 returned with the side stream NOT joined: before the arena / scratch can go anywhere, order the caller's
                # stream behind it (the graph is about to be dropped with work still in flight on both streams)
                if side is not None:
                    torch.cuda.current_stream(dev).wait_stream(side)
                    for t in (arena, work):
                        t.record_stream(side)
                ctx.arena = None
                raise
            _lib.call("gh_get_backward", *args, 2, main, side_raw)
        ctx.arena = None                  # (a second backward on this graph raises above)
        if pool is not None:              # both streams are joined at the end of gh_get_backward: later launches are ordered behind it
            pool.give(arena)
            pool.give(work)
        if direct:
            return (None, None, None) + (None,) * len(ctx.anchor_ids)
        return (None, None, None) + tuple(views.get(i) for i in ctx.anchor_ids)


def model_pool(model) -> _ArenaPool:
    """The persistent arena pool of `model` (public: `get_amd.fused.model_pool(model).clear()` returns its multi-GB buffers to torch's
    caching allocator -- they are invisible to torch.cuda.empty_cache() while pooled; a no-grad forward clears them as well)."""
    return _binding(model).pool


def eligible(model, query, kargs) -> bool:
    if not ENABLED or not query.is_cuda:
        return False
    if K.DocContentNoPaddingEvidence not in kargs or kargs[K.DocContentNoPaddingEvidence].shape[0] == 0:
        return False
    return _binding(model).eligible()


def forward(model, query, document, kargs):
    """Returns (phi, word_w (rows, hw) -- compact rows when the batch carries a plan --, evd_w, plan)."""
    binding = _binding(model)
    doc = kargs[K.DocContentNoPaddingEvidence]
    q_adj = ops.as_packed(kargs[K.Query_Adj])
    d_adj = ops.as_packed(kargs[K.Evd_Docs_Adj])
    plan = d_adj.plan
    if plan is not None and plan.m_real <= 0:
        plan = None
    if plan is None and AUTO_COMPACT and not isinstance(kargs[K.Evd_Docs_Adj], PackedAdj):
        plan = _plan_from_dense(d_adj, doc)
    prep = _prepare(model, query, document, kargs, q_adj, d_adj, plan, doc)
    side = ops.side_stream(query.device) if ops.CLAIM_SIDE_STREAM else None
    # anchors: the tensors autograd tracks.  With a FlatTrainer every gradient lands in the bucket directly, so ONE
    # parameter is enough to keep the graph connected; otherwise all live parameters are inputs and get their gradients
    # back from the backward call.
    params = binding.params + [t for t in binding.tables if t.requires_grad]
    if torch.is_grad_enabled() and all(ops._direct(p) for p in params):
        anchors = (params[-1],)
    else:
        anchors = tuple(params)
    phi, word_w, evd_w, score, keep = _GetFused.apply(binding, prep, side, *anchors)
    gw = model.ggnn_with_gsl
    gw.last_score, gw.last_keep = score, keep
    return phi, word_w, evd_w, plan


# Dense adjacencies handed over through the reference API (handlers/mz_sampler.py:146-160) carry no node-compact plan, so
# the model used to run the reference's padded layout for them: every layer on all R rows of every evidence, a third of
# them padding nodes (0.80 instead of 0.56 GFLOP per pair).  The plan only needs the node counts, and those are in the
# ids: convert_text zero-pads the node list behind the unique tokens (interactions.py:349).  One 8-byte read-back per
# forward (the total m_real sizes the launches) -- the fitter that hands over dense tensors synchronises after every step
# anyway (loss.item(), char_man_fitter_query_repr1.py:123).  Guard: an adjacency whose padding rows are not empty (never
# produced by the reference, but legal input) keeps the padded layout.  GET_AMD_AUTO_COMPACT=0 turns this off.
AUTO_COMPACT = os.environ.get("GET_AMD_AUTO_COMPACT", "1") != "0"


def _plan_from_dense(d_adj: PackedAdj, doc: torch.Tensor):
    n, r = d_adj.n, d_adj.r
    with torch.no_grad():
        real = doc.reshape(n, r) >= 1
        n_nodes = real.sum(1, dtype=torch.int32)
        prefix = torch.arange(r, device=doc.device, dtype=torch.int32)[None, :] < n_nodes[:, None]
        row_has_edges = (d_adj.bits != 0).any(-1)
        bad = ((real != prefix) | (row_has_edges & ~real)).sum()          # ids not prefix-shaped, or a padding node with edges
        m_real, bad = torch.stack([n_nodes.sum().to(torch.int64), bad.to(torch.int64)]).tolist()
    if bad or m_real <= 0 or m_real >= n * r:
        return None
    return ops.RaggedPlan(n_nodes, _i32(doc.reshape(n, r)), int(m_real))


class _CrossEntropy(torch.autograd.Function):
    """losses.py:29-32 (mean CE): loss and its gradient in one kernel; the backward is one scale."""

    @staticmethod
    def forward(ctx, phi, labels):
        phi = ops._f32(phi)
        b, c = phi.shape
        labels = labels.to(torch.int64).contiguous()
        out = torch.empty(1 + b * c, device=phi.device, dtype=torch.float32)
        _lib.call("gh_cross_entropy", phi.data_ptr(), labels.data_ptr(), b, c, out.data_ptr(), out[1:].data_ptr(), _lib.stream())
        ctx.save_for_backward(out)
        ctx.shape = (b, c)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        (out,) = ctx.saved_tensors
        if g.data_ptr() == unit_seed(g.device).data_ptr():      # backward(loss) below: the upstream gradient IS the constant 1
            # (a fresh tensor, not a view of the saved one: autograd may accumulate into a gradient it solely owns -- phi feeding a
            #  second loss term -- and retain_graph / a second backward must find `out` unchanged; the copy is 4 bytes x B x C)
            return out[1:].view(ctx.shape).clone(), None
        return out[1:].view(ctx.shape) * g, None          # (one scale launch; g is the scalar upstream gradient)


_UNIT = {}


def unit_seed(device) -> torch.Tensor:
    """The constant 1.0 on `device` (one per device, never written): the root gradient `backward(loss)` hands to autograd."""
    key = (device.type, device.index)
    t = _UNIT.get(key)
    if t is None:
        t = _UNIT[key] = torch.ones((), device=device, dtype=torch.float32)
    return t


def backward(loss: torch.Tensor) -> None:
    """`loss.backward()` for a loss made by `cross_entropy`, without the two scalar launches autograd adds at the root (a fill for
    ones_like(loss), a multiply of the saved logit gradient by it -- 11 us + a dispatch gap per step on MI355X): the root gradient
    is a cached constant 1, which the loss's backward recognises by its address.  Any other loss (not a 0-d fp32 tensor: a
    grad_tensors mismatch otherwise) takes the plain path."""
    if loss.dim() != 0 or loss.dtype != torch.float32 or not loss.is_cuda:
        loss.backward()
        return
    torch.autograd.backward(loss, grad_tensors=[unit_seed(loss.device)])


def cross_entropy(phi: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    """Mean cross-entropy of (B, C) logits against (B,) labels (losses.py:29-32), one launch forward + one backward."""
    _lib.require_cuda(phi, labels)
    return _CrossEntropy.apply(phi, labels)
