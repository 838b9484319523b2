"""Packed word-graph store for a corpus of texts.

The reference's dataset container (``interactions.py:295-332`` ``ClassificationInteractions.convert_leftright``)
runs the pure-Python ``convert_text`` once per unique text and keeps, per text, a Python list of node ids plus a
dense ``fixed_length x fixed_length`` float64 adjacency (``dict_adj``); the lengths 30/100 are hard-coded
(``:303``).  :class:`GraphCache` is the native replacement: all texts of one side (claims or evidences) go through
``gh_graph_build`` in a few launches and are kept as four flat arrays

    node_ids (N,R) int32 | n_nodes (N,) int32 | bits (N,R,W) int64 | dinv (N,R) float32        (W = ceil(R/64))

i.e. 2.4 KB per 100-token evidence instead of an 80 KB dense matrix, any ``fixed_length <= 256``.  A mini-batch
is a row gather of those arrays (:meth:`gather`), the node counts needed by the node-compact layout are a host
array (:attr:`n_nodes_host`), and the whole store round-trips through one ``.npz`` file.
"""
from __future__ import annotations

from typing import Dict, Hashable, Iterable, Optional, Sequence

import numpy as np
import torch

from . import ops
from .ops import PackedAdj


class GraphCache:
    """Packed graphs of ``N`` texts padded to ``fixed_length`` tokens, built with window ``window``."""

    def __init__(self, keys: Sequence[Hashable], node_ids, n_nodes, bits, dinv, fixed_length: int, window: int):
        self.fixed_length, self.window = int(fixed_length), int(window)
        self.keys = list(keys)
        self.node_ids, self.n_nodes, self.bits, self.dinv = node_ids, n_nodes, bits, dinv
        n = len(self.keys)
        w = (self.fixed_length + 63) // 64
        assert tuple(node_ids.shape) == (n, self.fixed_length) and tuple(n_nodes.shape) == (n,)
        assert tuple(bits.shape) == (n, self.fixed_length, w) and tuple(dinv.shape) == (n, self.fixed_length)
        self.index: Dict[Hashable, int] = {}
        for i, k in enumerate(self.keys):          # interactions.py:324 `assert index not in contents_dict`
            if k in self.index:
                raise ValueError(f"GraphCache: duplicate text id {k!r}")
            self.index[k] = i
        self.n_nodes_host = np.asarray(n_nodes.cpu() if torch.is_tensor(n_nodes) else n_nodes, dtype=np.int64)

    # ------------------------------------------------------------------ construction
    @classmethod
    def build(cls, keys: Sequence[Hashable], tokens, lengths, window: int, device="cuda:0",
              chunk: int = 1 << 16) -> "GraphCache":
        """tokens (N,R) post-padded raw token ids, lengths (N,): ``convert_text`` (interactions.py:334-351) for all
        N texts on the device, ``chunk`` texts per launch.  Texts of length 0 are rejected like the reference's
        ``assert length_ != 0`` (:322)."""
        tokens = np.ascontiguousarray(np.asarray(tokens))
        lengths = np.ascontiguousarray(np.asarray(lengths))
        if tokens.ndim != 2 or lengths.shape != (tokens.shape[0],):
            raise ValueError("GraphCache.build: tokens must be (N,R) and lengths (N,)")
        if len(keys) != tokens.shape[0]:
            raise ValueError("GraphCache.build: one key per text")
        if tokens.shape[0] and int(lengths.min()) <= 0:
            raise ValueError("GraphCache.build: empty text (the reference asserts length != 0)")
        dev = torch.device(device)
        outs = ([], [], [], [])
        for lo in range(0, tokens.shape[0], chunk):
            t = torch.from_numpy(tokens[lo:lo + chunk]).to(dev)
            ln = torch.from_numpy(lengths[lo:lo + chunk]).to(dev)
            adj, ids, nn = ops.graph_build(t, ln, window)
            for o, v in zip(outs, (ids, nn, adj.bits, adj.dinv)):
                o.append(v)
        r = tokens.shape[1]
        w = (r + 63) // 64
        if not outs[0]:
            ids = torch.empty((0, r), device=dev, dtype=torch.int32)
            return cls(keys, ids, torch.empty((0,), device=dev, dtype=torch.int32),
                       torch.empty((0, r, w), device=dev, dtype=torch.int64),
                       torch.empty((0, r), device=dev, dtype=torch.float32), r, window)
        return cls(keys, *[torch.cat(o, 0) for o in outs], r, window)

    @classmethod
    def from_arrays(cls, keys, node_ids, n_nodes, bits, dinv, window: int, device=None) -> "GraphCache":
        """Wrap arrays that already hold packed graphs (e.g. loaded from disk); `device` None keeps them on the host."""
        conv = lambda a, dt: (torch.as_tensor(np.ascontiguousarray(a)).to(dt) if not torch.is_tensor(a) else a.to(dt))
        arrs = [conv(node_ids, torch.int32), conv(n_nodes, torch.int32), conv(bits, torch.int64), conv(dinv, torch.float32)]
        if device is not None:
            arrs = [a.to(device) for a in arrs]
        return cls(keys, *arrs, arrs[0].shape[1], window)

    # ------------------------------------------------------------------ persistence
    def save(self, path: str):
        keys = np.asarray(self.keys)
        if keys.dtype == object:
            raise ValueError("GraphCache.save: keys must be all ints or all strings")
        np.savez_compressed(path, keys=keys, node_ids=self.node_ids.cpu().numpy(), n_nodes=self.n_nodes.cpu().numpy(),
                            bits=self.bits.cpu().numpy(), dinv=self.dinv.cpu().numpy(),
                            meta=np.asarray([self.fixed_length, self.window], dtype=np.int64))

    @classmethod
    def load(cls, path: str, device=None) -> "GraphCache":
        z = np.load(path, allow_pickle=False)
        fixed_length, window = (int(v) for v in z["meta"])
        c = cls.from_arrays(z["keys"].tolist(), z["node_ids"], z["n_nodes"], z["bits"], z["dinv"], window, device)
        assert c.fixed_length == fixed_length
        return c

    def to(self, device) -> "GraphCache":
        return GraphCache(self.keys, self.node_ids.to(device), self.n_nodes.to(device), self.bits.to(device),
                          self.dinv.to(device), self.fixed_length, self.window)

    # ------------------------------------------------------------------ access
    def __len__(self):
        return len(self.keys)

    @property
    def device(self):
        return self.node_ids.device

    def rows(self, keys: Iterable[Hashable]) -> np.ndarray:
        """Row indices of the given text ids (KeyError names the first unknown id)."""
        try:
            return np.fromiter((self.index[k] for k in keys), dtype=np.int64)
        except KeyError as e:
            raise KeyError(f"GraphCache: unknown text id {e.args[0]!r}") from None

    def gather(self, rows, compact: bool = False):
        """Packed graphs of the given rows, in that order: (PackedAdj, node_ids (n,R) int32, n_nodes (n,) int32).
        compact=True attaches the node-compact plan (its row count comes from the host copy of n_nodes: no sync)."""
        rows_host = np.asarray(rows.cpu() if torch.is_tensor(rows) else rows, dtype=np.int64)
        idx = torch.as_tensor(rows_host, device=self.device)
        ids = self.node_ids.index_select(0, idx)
        nn = self.n_nodes.index_select(0, idx)
        adj = PackedAdj(self.bits.index_select(0, idx), self.dinv.index_select(0, idx), None, None, int(idx.numel()),
                        self.fixed_length)
        if compact and idx.numel() > 0:
            adj = adj.with_plan(ops.RaggedPlan(nn, ids, int(self.n_nodes_host[rows_host].sum())))
        return adj, ids, nn

    def dense(self, rows) -> torch.Tensor:
        """The reference's ``dict_adj`` entries for the given rows: (n,R,R) float64 normalised adjacency."""
        adj, _, _ = self.gather(rows)
        return adj.to_dense().double()

    def nbytes(self) -> int:
        return sum(int(t.numel()) * t.element_size() for t in (self.node_ids, self.n_nodes, self.bits, self.dinv))


class CachedBatcher:
    """Mini-batches from two caches (claims, evidences) and the claim -> evidence relation
    (``convert_relations``, interactions.py:353-385): the device work per batch is four row gathers, no graph
    construction.  Produces the same ``(query, document, kargs)`` triple as :class:`get_amd.batch.NativeBatch`."""

    def __init__(self, claims: GraphCache, evidences: GraphCache, relation: Dict[Hashable, Sequence[Hashable]],
                 n_max: int = 30, compact: bool = True):
        self.claims, self.evidences, self.relation = claims, evidences, relation
        self.n_max, self.compact = int(n_max), bool(compact)
        for q, docs in relation.items():
            if len(docs) > self.n_max:
                raise ValueError(f"claim {q!r} has {len(docs)} evidences, more than n_max={self.n_max}")

    def inputs(self, claim_keys: Sequence[Hashable], doc_sources=None, query_sources=None):
        """claim_keys: the claims of this batch.  doc_sources (B,n_max) / query_sources (B,1) optional tensors."""
        from .keywords import KeyWordSettings as K
        dev = self.evidences.device
        counts = np.asarray([len(self.relation[q]) for q in claim_keys], dtype=np.int64)
        c_rows = self.claims.rows(claim_keys)
        e_rows = self.evidences.rows(d for q in claim_keys for d in self.relation[q])
        qa, q_ids, q_n = self.claims.gather(c_rows)
        da, d_ids, _ = self.evidences.gather(e_rows, compact=self.compact)
        b, b1, r = len(claim_keys), int(counts.sum()), self.evidences.fixed_length
        offs = np.concatenate([[0], np.cumsum(counts)])[:-1]
        p2c = np.repeat(np.arange(b), counts)
        slot = torch.as_tensor((p2c * self.n_max + (np.arange(b1) - offs[p2c])).astype(np.int64), device=dev)
        document = torch.zeros((b * self.n_max, r), device=dev, dtype=torch.int32)
        document.index_copy_(0, slot, d_ids)
        kargs = {
            K.Query_lens: q_n, K.Doc_lens: None, K.DocLensIndices: None,
            K.DocContentNoPaddingEvidence: d_ids, K.EvidenceCountPerQuery: torch.as_tensor(counts, device=dev),
            K.FIXED_NUM_EVIDENCES: self.n_max, K.Query_Adj: qa, K.Evd_Docs_Adj: da,
            K.DocSources: doc_sources, K.QuerySources: query_sources,
        }
        return q_ids, document.view(b, self.n_max, r), kargs
