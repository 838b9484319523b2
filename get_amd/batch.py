"""Device-resident mini-batches for the GET hot path.

The reference re-materialises dense ``(N,30,100,100)`` float64 adjacency on the host every epoch
(handlers/mz_sampler.py:115-176), de-pads it per claim in a Python loop with two device syncs per claim
(Fitting/FittingFC/char_man_fitter_query_repr1.py:204-250) and ships 76.8 MB per step over PCIe.
:class:`NativeBatch` keeps only token ids and counts on the device and rebuilds the packed graphs per
step with ``gh_graph_build``; :func:`kargs_from_reference_tensors` is the compatibility shim for callers
that still hand over the reference's dense tensors.
"""
from __future__ import annotations

import os

import numpy as np
import torch

from . import ops
from .keywords import KeyWordSettings as K


class NativeBatch:
    """Raw token form of one mini-batch, resident in HBM.

    claim_tokens (B,L) / claim_len (B,): post-padded claim token ids and lengths;
    evd_tokens (B1,R) / evd_len (B1,): the de-padded evidences of all claims, claim-major;
    evd_counts (B,): evidences per claim (sum = B1); doc_sources (B,n_max) with -1 padding;
    query_sources (B,1); labels (B,).
    """

    def __init__(self, claim_tokens, claim_len, evd_tokens, evd_len, evd_counts, doc_sources, query_sources, labels,
                 window: int, n_max: int = 30, device="cuda:0", compact: bool = None, pinned: bool = False):
        dev = torch.device(device)
        counts_host = np.asarray(evd_counts.cpu() if torch.is_tensor(evd_counts) else evd_counts, dtype=np.int64)
        self.window, self.n_max, self.device = int(window), int(n_max), dev
        self._counts_host = counts_host
        self.b, self.b1 = int(counts_host.shape[0]), int(counts_host.sum())
        if counts_host.min(initial=0) < 0 or counts_host.max(initial=0) > self.n_max:
            raise ValueError(f"NativeBatch: evidence counts must lie in [0, n_max={self.n_max}], got "
                             f"[{int(counts_host.min(initial=0))}, {int(counts_host.max(initial=0))}]")
        # slot of every pair inside the (B, n_max, R) padded evidence tensor -- host arithmetic, done once
        offs = np.concatenate([[0], np.cumsum(counts_host)])[:-1]
        p2c = np.repeat(np.arange(self.b), counts_host)
        slot = (p2c * self.n_max + (np.arange(self.b1) - offs[p2c])).astype(np.int64)
        host = [np.ascontiguousarray(counts_host), slot, doc_sources, query_sources, labels,
                (claim_tokens, np.int32), (claim_len, np.int32), (evd_tokens, np.int32), (evd_len, np.int32)]
        if pinned and not any(torch.is_tensor(x[0] if isinstance(x, tuple) else x) for x in host):
            # streaming loaders: every host array goes into ONE pinned staging buffer (8-byte items first, 16-byte
            # aligned pieces) and crosses PCIe as one asynchronous copy; the tensors below are views of its device twin
            arrs = []
            for x in host:
                a, dt = x if isinstance(x, tuple) else (x, np.int64)
                arrs.append(np.ascontiguousarray(np.asarray(a), dtype=dt))
            offs_b, total = [], 0
            for a in arrs:
                offs_b.append(total)
                total += (a.nbytes + 15) // 16 * 16
            stage = torch.empty(max(total, 16), dtype=torch.uint8, pin_memory=True)
            sn = stage.numpy()
            for a, o in zip(arrs, offs_b):
                sn[o:o + a.nbytes] = a.reshape(-1).view(np.uint8)
            devbuf = stage.to(dev, non_blocking=True)
            self._stage = stage            # keep the pinned buffer alive until the copy has been consumed
            views = []
            for a, o in zip(arrs, offs_b):
                tdt = torch.int64 if a.dtype == np.int64 else torch.int32
                views.append(devbuf[o:o + a.nbytes].view(tdt).view(a.shape))
            (self.counts, self._slot, self.doc_sources, self.query_sources, self.labels, self.claim_tokens, self.claim_len,
             self.evd_tokens, self.evd_len) = views
        else:
            t = lambda a, dt=None: torch.as_tensor(np.ascontiguousarray(a) if isinstance(a, np.ndarray) else a).to(dev)
            self.claim_tokens, self.claim_len = t(claim_tokens).int(), t(claim_len).int()
            self.evd_tokens, self.evd_len = t(evd_tokens).int(), t(evd_len).int()
            self.counts = t(counts_host)
            self.doc_sources, self.query_sources, self.labels = t(doc_sources), t(query_sources), t(labels)
            self._slot = t(slot)
        assert self.evd_tokens.shape[0] == self.b1, "evd_tokens must hold sum(evd_counts) rows"
        # every count <= n_max (checked above): the backward needs no zero fill of unmapped rows.  The promise is tied to the
        # tensor's version counter: an in-place edit of `counts` afterwards voids it (fused._prepare compares)
        self.counts._gh_fit = self.counts._version
        # node-compact layout (ops.RaggedPlan): the host has to know the total number of real evidence nodes, i.e.
        # the unique tokens per evidence -- what convert_text returns as `length_` (interactions.py:351) at load time
        if compact is None:
            compact = os.environ.get("GET_AMD_PADDED", "0") != "1"
        self.compact = bool(compact)
        # ... which the device graph build reports (n_nodes).  One launch + one 4-byte read-back when the batch is
        # CONSTRUCTED -- the stage a loader runs one batch ahead of the training step -- instead of a per-row Python
        # set() over the token lists on the host.
        if self.b1 > 0:
            _, _, d_n = ops.graph_build(self.evd_tokens, self.evd_len, self.window)
            self.m_real = int(d_n.sum().item())
        else:
            self.m_real = 0
        self._graphs = None           # per-step graph buffers, allocated by the first inputs()

    def inputs(self):
        """Per-step device work: token ids -> (query node ids, padded document ids, kargs) for
        ``Graph_basedSemantiStructure.forward``.  interactions.py:334-351 for both sides, the node-compact plan and the
        padded ``document`` tensor of basic_fc_model.py:94-121 run as ONE library call (gh_get_prepare: two graph-build
        launches, the plan's scan + fill, one scatter) into buffers this batch owns: they are allocated on the first
        call and rewritten in place on every later one, so a step allocates nothing here."""
        from ._lib import call, ptr, stream
        c = self._graphs
        if c is None:
            dev = self.device
            b, b1 = self.b, self.b1
            l, r = self.claim_tokens.shape[1], self.evd_tokens.shape[1]
            wq, wd = (l + 63) // 64, (r + 63) // 64
            i32 = lambda *sh: torch.empty(sh, device=dev, dtype=torch.int32)
            c = {"q_ids": i32(b, l), "q_n": i32(b), "q_bits": torch.empty((b, l, wq), device=dev, dtype=torch.int64),
                 "q_dinv": torch.empty((b, l), device=dev, dtype=torch.float32),
                 "d_ids": i32(b1, r), "d_n": i32(b1), "d_bits": torch.empty((b1, r, wd), device=dev, dtype=torch.int64),
                 "d_dinv": torch.empty((b1, r), device=dev, dtype=torch.float32),
                 "document": torch.zeros((b * self.n_max, r), device=dev, dtype=torch.int32)}      # unused slots stay zero
            qa = ops.PackedAdj(c["q_bits"], c["q_dinv"], None, None, b, l)
            da = ops.PackedAdj(c["d_bits"], c["d_dinv"], None, None, b1, r)
            plan = None
            if self.compact and b1 > 0:
                plan = ops.RaggedPlan.empty(b1, r, self.m_real, dev)
                da = da.with_plan(plan)
            c["plan"] = plan
            c["kargs"] = {
                K.Query_lens: c["q_n"], K.Doc_lens: None, K.DocLensIndices: None,
                K.DocContentNoPaddingEvidence: c["d_ids"], K.EvidenceCountPerQuery: self.counts,
                K.FIXED_NUM_EVIDENCES: self.n_max, K.Query_Adj: qa, K.Evd_Docs_Adj: da,
                K.DocSources: self.doc_sources, K.QuerySources: self.query_sources,
            }
            c["doc_view"] = c["document"].view(b, self.n_max, r)
            plan_ptrs = (ptr(plan.goff), ptr(plan.rowg), ptr(plan.src), ptr(plan.cids), ptr(plan.maskf)) if plan is not None \
                else (None,) * 5
            c["args"] = (ptr(self.claim_tokens), ptr(self.claim_len), b, l, ptr(self.evd_tokens), ptr(self.evd_len), b1, r, self.window,
                         ptr(c["q_ids"]), ptr(c["q_n"]), ptr(c["q_bits"]), ptr(c["q_dinv"]),
                         ptr(c["d_ids"]), ptr(c["d_n"]), ptr(c["d_bits"]), ptr(c["d_dinv"]),
                         self.m_real if plan is not None else -1, *plan_ptrs, ptr(self._slot), ptr(c["document"]))
            self._graphs = c
        call("gh_get_prepare", *c["args"], stream())
        return c["q_ids"], c["doc_view"], c["kargs"]

    def device_tensors(self):
        """Every device tensor this batch owns (stream hand-over: `t.record_stream(consumer_stream)`)."""
        return [t for t in (self.counts, self._slot, self.doc_sources, self.query_sources, self.labels, self.claim_tokens,
                            self.claim_len, self.evd_tokens, self.evd_len) if torch.is_tensor(t) and t.is_cuda] + \
               ([v for v in self._graphs.values() if torch.is_tensor(v) and v.is_cuda] if self._graphs else [])

    def subset(self, lo: int, hi: int) -> "NativeBatch":
        """Claims [lo, hi) of this batch as a batch of their own (device-side slices; evaluation in chunks)."""
        offs = np.concatenate([[0], np.cumsum(self._counts_host)])
        r0, r1 = int(offs[lo]), int(offs[hi])
        return NativeBatch(self.claim_tokens[lo:hi], self.claim_len[lo:hi], self.evd_tokens[r0:r1], self.evd_len[r0:r1],
                           self._counts_host[lo:hi], self.doc_sources[lo:hi], self.query_sources[lo:hi], self.labels[lo:hi],
                           window=self.window, n_max=self.n_max, device=self.device, compact=self.compact)


class ReferenceDepad:
    """The reference fitter's dense tensors (char_man_fitter_query_repr1.py:196-223: `evd_doc_contents` (B,n,R) ids,
    `evd_docs_adj` (B,n,R,R) float64, `query_adj` (B,L,L), counts, sources) -> forward kargs, in two phases:

    * construction LAUNCHES the de-padding (``gh_ref_depad``: ids narrowed, adjacency packed, node counts, normalisation and
      layout checks, one launch) on `stream` (default: the current one) and an asynchronous copy of its five counters into
      pinned host memory;
    * :meth:`kargs` waits for that copy only (an event), sizes the node-compact plan from the counters and returns the kargs.

    Called back to back (``kargs_from_reference_tensors``) this is one launch + one read-back per step -- but the read-back
    then waits for everything queued before it, i.e. for the whole previous step, and the device idles while the host issues
    the next forward (measured: 154.8 K against 164.4 K pairs/s native, round 4).  With the construction on a side stream ONE
    BATCH AHEAD (:func:`prefetch_reference`), the counters are on the host long before they are needed and no bubble is left.

    An adjacency whose values are the normalised binary graph D^-1/2 A D^-1/2 (always true for `convert_text` output,
    interactions.py:11-18) is handed to the kernels as bit rows + dinv ("normalised" mode, 2 KB per graph) instead of bit
    rows + dense fp32 values ("weighted" mode, 40 KB per graph); any other adjacency keeps the weighted mode."""

    def __init__(self, query_lens, evd_doc_contents, evd_docs_adj, query_adj, evd_counts, doc_sources, query_sources=None,
                 n_max: int = 30, stream=None):
        from ._lib import call, ptr
        b, n, r = evd_doc_contents.shape
        dev = evd_docs_adj.device
        self._meta = (query_lens, query_adj, evd_counts, doc_sources, query_sources, int(n_max), r)
        self._stream = stream if stream is not None else torch.cuda.current_stream(dev)
        if stream is not None:
            # the construction stream must see the tensors the caller produced on ITS stream (prefetch_reference did this itself;
            # a direct ReferenceDepad(stream=side) user had no such ordering)
            self._stream.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(self._stream):
            ids = evd_doc_contents.contiguous()
            counts = evd_counts.to(device=dev, dtype=torch.int64).contiguous()
            w = (r + 63) // 64
            self.d_ids = torch.empty((b * n, r), device=dev, dtype=torch.int32)
            self.bits = torch.empty((b * n, r, w), device=dev, dtype=torch.int64)
            self.vals = torch.empty((b * n, r, r), device=dev, dtype=torch.float32)
            self.dinv = torch.empty((b * n, r), device=dev, dtype=torch.float32)
            self.n_nodes = torch.empty((b * n,), device=dev, dtype=torch.int32)
            stats = torch.empty((5,), device=dev, dtype=torch.int64)
            self._args = (ptr(counts), b, n, r, ptr(ids), 1 if ids.dtype == torch.int64 else 0, ptr(evd_docs_adj),
                          ptr(self.d_ids), ptr(self.bits), ptr(self.vals), ptr(self.dinv), ptr(self.n_nodes), ptr(stats))
            call("gh_ref_depad", *self._args, 0, self._stream.cuda_stream)
            self._host = torch.empty((5,), dtype=torch.int64, pin_memory=True)
            self._host.copy_(stats, non_blocking=True)
            self._event = torch.cuda.Event()
            self._event.record(self._stream)
        self._keep = (ids, counts, stats, evd_docs_adj)

    def kargs(self):
        """Forward kargs (the current stream is ordered behind the de-padding)."""
        query_lens, query_adj, evd_counts, doc_sources, query_sources, n_max, r = self._meta
        self._event.synchronize()                          # the one host wait of this path
        b1, m_real, bad, weighted, _ = self._host.tolist()
        cur = torch.cuda.current_stream(self.d_ids.device)
        if cur != self._stream:
            cur.wait_event(self._event)
            # everything allocated on the construction stream and touched from `cur` (the weighted fallback below relaunches on cur
            # with counts / stats / ids as well)
            for t in (self.d_ids, self.bits, self.vals, self.dinv, self.n_nodes) + tuple(x for x in self._keep if torch.is_tensor(x) and x.is_cuda):
                t.record_stream(cur)
        e_conts = self.d_ids[:b1]
        if weighted:
            # some graph is not D^-1/2 A D^-1/2 of its own pattern (never from convert_text, but legal input): the whole batch
            # takes the weighted mode; the second launch rewrites every graph's dense values
            from ._lib import call
            call("gh_ref_depad", *self._args, 1, cur.cuda_stream)
            adj = ops.PackedAdj(self.bits[:b1], None, self.vals[:b1], None, int(b1), r)
        else:
            adj = ops.PackedAdj(self.bits[:b1], self.dinv[:b1], None, None, int(b1), r)
        if b1 > 0 and not bad and 0 < m_real < b1 * r and os.environ.get("GET_AMD_AUTO_COMPACT", "1") != "0":
            adj = adj.with_plan(ops.RaggedPlan(self.n_nodes[:b1], e_conts, int(m_real)))
        kargs = {
            K.Query_lens: query_lens, K.Doc_lens: None, K.DocLensIndices: None,
            K.DocContentNoPaddingEvidence: e_conts, K.EvidenceCountPerQuery: evd_counts,
            K.FIXED_NUM_EVIDENCES: n_max, K.Query_Adj: query_adj, K.Evd_Docs_Adj: adj, K.DocSources: doc_sources,
        }
        if query_sources is not None:
            kargs[K.QuerySources] = query_sources
        return kargs

    @staticmethod
    def applies(evd_doc_contents, evd_docs_adj) -> bool:
        b, n, r = evd_doc_contents.shape
        return (evd_docs_adj.is_cuda and evd_docs_adj.dtype == torch.float64 and evd_docs_adj.is_contiguous() and r <= 256
                and evd_doc_contents.is_cuda and evd_doc_contents.dtype in (torch.int32, torch.int64) and b > 0)


def prefetch_reference(batches, n_max: int = 30, device=None):
    """Generator over forward kargs for an iterable of the fitter's dense hand-overs, each a tuple
    ``(query_lens, evd_doc_contents, evd_docs_adj, query_adj, evd_counts, doc_sources, query_sources)``: the de-padding of
    batch i+1 is launched on a side stream while the caller works on batch i (what a DataLoader-side prefetcher does for the
    H2D copies), so that no read-back ever waits for a training step."""
    it = iter(batches)
    side = None

    def start(item):
        nonlocal side
        if item is None:
            return None
        if not ReferenceDepad.applies(item[1], item[2]):
            return item
        if side is None:
            side = torch.cuda.Stream(device=item[2].device)
        # the side stream must see the tensors the caller produced on its own stream
        side.wait_stream(torch.cuda.current_stream(item[2].device))
        return ReferenceDepad(*item, n_max=n_max, stream=side)

    nxt = start(next(it, None))
    while nxt is not None:
        cur, nxt = nxt, start(next(it, None))
        yield cur.kargs() if isinstance(cur, ReferenceDepad) else kargs_from_reference_tensors(*cur, n_max=n_max)


def kargs_from_reference_tensors(query_lens, evd_doc_contents, evd_docs_adj, query_adj, evd_counts, doc_sources,
                                 query_sources=None, n_max: int = 30, fused: bool = True):
    """Compatibility shim: the tensors the reference fitter holds before its de-padding loop
    (char_man_fitter_query_repr1.py:196-223) -> forward kargs, with the per-claim ``[:evd_cnt]`` slicing
    done on the device instead of a Python loop with 2 syncs per claim.

    evd_doc_contents (B,n,R) ids, evd_docs_adj (B,n,R,R) dense, query_adj (B,L,L), evd_counts (B,).

    Device tensors with a float64 adjacency (what handlers/mz_sampler.py:146-160 ships) take ONE library launch
    (:class:`ReferenceDepad`) and ONE 40-byte read-back; the evidence adjacency then comes back already packed
    (``ops.PackedAdj``, with the node-compact plan attached when the graphs allow it), which every ``forward`` of this
    package accepts in place of the dense tensor.  ``fused=False`` keeps the plain tensor form (boolean-mask gathers: two
    syncs, three passes over the 77 MB adjacency, then packing inside the forward).  A training loop should prefer
    :func:`prefetch_reference`, which hides the read-back behind the previous step."""
    b, n, r = evd_doc_contents.shape
    if fused and ReferenceDepad.applies(evd_doc_contents, evd_docs_adj):
        return ReferenceDepad(query_lens, evd_doc_contents, evd_docs_adj, query_adj, evd_counts, doc_sources, query_sources,
                              n_max=n_max).kargs()
    valid = torch.arange(n, device=evd_counts.device)[None, :] < evd_counts[:, None]          # (B,n)
    e_conts = evd_doc_contents[valid]                                                         # (B1,R) claim-major
    e_adj = evd_docs_adj[valid]                                                               # (B1,R,R)
    kargs = {
        K.Query_lens: query_lens, K.Doc_lens: None, K.DocLensIndices: None,
        K.DocContentNoPaddingEvidence: e_conts, K.EvidenceCountPerQuery: evd_counts,
        K.FIXED_NUM_EVIDENCES: n_max, K.Query_Adj: query_adj, K.Evd_Docs_Adj: e_adj, K.DocSources: doc_sources,
    }
    if query_sources is not None:
        kargs[K.QuerySources] = query_sources
    return kargs


@torch.no_grad()
def batched_predict(model, batch: "NativeBatch", claims_per_call: int = 0):
    """Evaluation without the reference's one-claim-per-forward loop
    (Fitting/FittingFC/char_man_fitter_query_repr1.py:260-364 runs ~780 B=1 forwards per validation pass):
    all claims of `batch` go through the ragged kernels in one forward, or in chunks of `claims_per_call` claims
    (claims are independent, so chunking changes nothing but the peak memory).

    Returns (phi (B,C), word_att list of (n_b, R, hw) tensors per claim, evd_att (B, n_max, he)) -- the
    observables `_prepare_error_analysis` consumes (:422-472), with each head's weights summing to one."""
    was_training = model.training
    model.train(False)
    try:
        step = batch.b if not claims_per_call or claims_per_call >= batch.b else int(claims_per_call)
        phis, words, evds = [], [], []
        for lo in range(0, batch.b, step):
            part = batch if step == batch.b else batch.subset(lo, min(batch.b, lo + step))
            query, document, kargs = part.inputs()
            kargs = dict(kargs)
            kargs[K.OutputRankingKey] = True
            phi, (word_w, evd_w) = model(query, document, **kargs)
            phis.append(phi)
            evds.append(evd_w)
            words += list(torch.split(word_w, part._counts_host.tolist(), dim=0))
        return torch.cat(phis, 0), words, torch.cat(evds, 0)
    finally:
        model.train(was_training)
