"""Drop-in ``nn.Module`` mirror of the reference's hot-path classes, backed by libget_hip.so.

Same class names, constructor arguments, ``forward()`` signatures, parameter names and shapes
as upstream, so ``MasterFC/master_get.py`` and reference checkpoints
load unchanged:

  Models/BiDAF/wrapper.py                      Linear, GGNN, GSL, GGNN_with_GSL, LSTM
  thirdparty/two_branches_attention.py         ConcatNotEqualSelfAtt, ConcatSelfAtt
  thirdparty/self_attention.py                 MultiHeadSelfAttentionICLR2017Extend
  Models/FCWithEvidences/graph_based_semantic_structure.py   Graph_basedSemantiStructure

Adjacency arguments may be the reference's dense ``(N,R,R)`` tensors (any float dtype; packed once
on the device, values kept exactly) or a native :class:`get_amd.ops.PackedAdj` from
:func:`get_amd.ops.graph_build`.
"""
from __future__ import annotations

from typing import Tuple

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib, ops
from .keywords import KeyWordSettings
from .ops import PackedAdj


def _drop_caches_on_load(module: nn.Module):
    """The forward reads cached derivatives of the weights (transposes, fused bias sums, packed scalar gates; ops.transposed
    / ops.derived) that are validated by tensor identity and in-place version.  load_state_dict copies
    into the parameters in place; every module here drops ALL cached derivatives after a load so that no stale entry can
    survive whatever the copy path did to the version counters.  Any OTHER raw `.data` write (EMA swaps, hand-written
    optimisers) must call ops.bump_weight_epoch() itself (it clears every derived entry, frozen ones included)."""
    module.register_load_state_dict_post_hook(lambda m, incompatible_keys: ops.bump_weight_epoch())


# ------------------------------------------------------------------ Models/BiDAF/wrapper.py:330-347
class Linear(nn.Module):
    def __init__(self, in_features, out_features, bias=True, dropout=0.0):
        super().__init__()
        _drop_caches_on_load(self)
        self.linear = nn.Linear(in_features=in_features, out_features=out_features, bias=bias)
        if dropout > 0:
            self.dropout = nn.Dropout(p=dropout)
        self.reset_params()

    def reset_params(self):
        # kaiming-normal weights; the reference's bias-zeroing branch never fires (wrapper.py:341),
        # so biases keep nn.Linear's default init
        nn.init.kaiming_normal_(self.linear.weight)

    def forward(self, x):
        if hasattr(self, "dropout"):
            x = self.dropout(x)
        return ops.linear(x, self.linear.weight, self.linear.bias)


# ------------------------------------------------------------------ Models/BiDAF/wrapper.py:174-208
class GGNN(nn.Module):
    def __init__(self, in_features, out_features, dropout=0.2):
        super().__init__()
        self.proj = Linear(in_features, out_features, bias=False)
        self.linearz0 = Linear(out_features, out_features)
        self.linearz1 = Linear(out_features, out_features)
        self.linearr0 = Linear(out_features, out_features)
        self.linearr1 = Linear(out_features, out_features)
        self.linearh0 = Linear(out_features, out_features)
        self.linearh1 = Linear(out_features, out_features)
        if dropout > 0:
            self.dropout = nn.Dropout(p=dropout)

    def _params(self):
        g = lambda m: (m.linear.weight, m.linear.bias)
        return (self.proj.linear.weight, *g(self.linearz0), *g(self.linearz1), *g(self.linearr0), *g(self.linearr1),
                *g(self.linearh0), *g(self.linearh1))

    def _drop(self):
        """(p, seed) of the fused input dropout for this call, or None to use the materialised nn.Dropout."""
        if not (hasattr(self, "dropout") and self.training and self.dropout.p > 0):
            return (0.0, 0)
        w = self.proj.linear.weight
        if ops.fused_dropout_ok(w.shape[1], w.shape[0]):
            return (float(self.dropout.p), ops.new_dropout_seed())
        return None

    def forward(self, adj, x, plan=None, rows=0, score=None):
        """adj: dense (N,R,R) or PackedAdj; x: (N,R,Din).  Returns (N,R,Dout).
        Training-mode input dropout (wrapper.py:189-190) runs inside the first GEMM's loader.
        plan (ops.RaggedPlan, internal fast path): x is node-compact (>= rows, Din); returns (rows, Dout).
        score (internal, see ops.ggnn_cell): also return the consuming word scorer's projection of the output."""
        adj = ops.as_packed(adj)
        d = self._drop()
        if d is None:
            return ops.ggnn_cell(adj, self.dropout(x), None, self._params(), plan=plan, rows=rows, score=score)
        return ops.ggnn_cell(adj, x, None, self._params(), d[0], d[1], plan=plan, rows=rows, score=score)

    def forward_ids(self, adj, embedding: nn.Embedding, ids: torch.Tensor, plan=None, rows=0, score=None):
        """Same cell on ``embedding(ids)`` with the row gather fused into the first GEMM
        (graph_based_semantic_structure.py:100,150); training-mode dropout is applied there too.  Falls
        back to an explicit lookup + nn.Dropout only for widths that are not float4-shaped.
        plan: node-compact layout, the ids come from ``plan.cids``; returns (rows, Dout)."""
        adj = ops.as_packed(adj)
        d = self._drop()
        if plan is not None:
            rows = rows or plan.m_real
            ids = plan.cids[:rows]
        if d is None:
            x = self.dropout(embedding(ids.long()))
            return ops.ggnn_cell(adj, x, None, self._params(), plan=plan, rows=rows, score=score)
        return ops.ggnn_cell(adj, embedding.weight, ids.to(torch.int32).reshape(-1), self._params(), d[0], d[1],
                             plan=plan, rows=rows, score=score)


# ------------------------------------------------------------------ Models/BiDAF/wrapper.py:210-227
class GSL(nn.Module):
    def __init__(self, rate):
        super().__init__()
        self.rate = rate

    def forward(self, adj, score):
        """Keep the top int(rate*N) nodes' rows and columns (union), no renormalisation, no gradient.
        Dense adjacency in -> dense refined adjacency out; PackedAdj in -> PackedAdj with keep-set."""
        n_nodes = adj.r if isinstance(adj, PackedAdj) else adj.shape[-1]
        k = int(self.rate * n_nodes)
        keep = ops.gsl_topk(score.reshape(score.shape[0], n_nodes), k)
        if isinstance(adj, PackedAdj):
            return adj.with_keep(keep)
        return PackedAdj.from_dense(adj).with_keep(keep).to_dense().to(adj.dtype)


# ------------------------------------------------------------------ Models/BiDAF/wrapper.py:153-172
class GGNN_with_GSL(nn.Module):
    def __init__(self, input_dim, hidden_dim, output_dim, rate=0.8, dropout=0.2):
        super().__init__()
        self.feat_prop1 = GGNN(input_dim, hidden_dim, dropout)
        self.word_scorer1 = GGNN(hidden_dim, 1, dropout)
        self.gsl1 = GSL(rate)
        self.feat_prop2 = GGNN(hidden_dim, output_dim, dropout)
        self.last_score = None     # observables of the last forward (scores, keep words) for analysis/tests
        self.last_keep = None
        # called (no arguments) during backward as soon as the gradient w.r.t. the first cell's output exists: from
        # then on only the first cell's own gradients are still to come (dist.FlatTrainer.attach_overlap)
        self.grad_milestone_hook = None

    def _milestone(self, feat):
        hook = self.grad_milestone_hook
        if hook is not None and feat.requires_grad:
            def fire(g, _hook=hook):
                _hook()
                return g
            feat.register_hook(fire)

    def _gate12(self):
        s = self.word_scorer1
        srcs = []
        for m in (s.linearz0, s.linearz1, s.linearr0, s.linearr1, s.linearh0, s.linearh1):
            srcs += [m.linear.weight, m.linear.bias]
        # the scorer's parameters never receive a gradient (wrapper.py:219: no gradient through top-k), so a FlatTrainer
        # leaves them out of its bucket and never rewrites them: the packed copy then only depends on tensor identity /
        # in-place version, not on the optimiser's weight epoch (one cat launch + a dozen host ops less per step)
        static = not any(getattr(t, "_gh_direct_grad", False) for t in srcs)
        return ops.derived("gate12", tuple(srcs), lambda: torch.cat([t.detach().reshape(1) for t in srcs]), frozen=static)

    def _scorer_drop(self):
        """(p, seed) of word_scorer1's own input dropout for this call (wrapper.py:189-190)."""
        s = self.word_scorer1
        if hasattr(s, "dropout") and self.training and s.dropout.p > 0:
            return float(s.dropout.p), ops.new_dropout_seed()
        return 0.0, 0

    def _score_arg(self):
        """Argument that makes the first cell's last epilogue also produce the scorer's projection (or None)."""
        w = self.word_scorer1.proj.linear.weight
        if not ops.scorer_fusable(w.shape[1]):
            return None
        return (w,) + self._scorer_drop()

    def _refine(self, adj: PackedAdj, feat, plan=None, collapsed=False, score_x=None):
        s = self.word_scorer1
        drop_p, seed = (0.0, 0) if score_x is not None else self._scorer_drop()
        k = int(self.gsl1.rate * adj.r)
        score, keep = ops.scorer_gsl(adj, feat, s.proj.linear.weight, self._gate12(), k, drop_p, seed, plan=plan,
                                     collapsed=collapsed, score_x=score_x)
        self.last_score, self.last_keep = score, keep
        return adj.with_keep(keep)

    def _first_cell(self, run):
        """Run feat_prop1 through `run(score)`; returns (feat, score_x or None)."""
        sc = self._score_arg()
        res = run(sc)
        return res if sc is not None else (res, None)

    def forward(self, adj, feat):
        adj = ops.as_packed(adj)
        feat, sx = self._first_cell(lambda sc: self.feat_prop1(adj, feat, score=sc))
        self._milestone(feat)
        adj_refined = self._refine(adj, feat, score_x=sx)
        return self.feat_prop2(adj_refined, feat)

    def forward_ids(self, adj, embedding, ids, plan=None):
        """plan (ops.RaggedPlan): node-compact fast path -- the first cell and the scorer run on every row (the
        padding nodes' scores compete in the top-k), the second cell on the real-node rows only; returns
        (plan.m_real, H) instead of (N,R,H)."""
        adj = ops.as_packed(adj)
        if plan is None:
            feat, sx = self._first_cell(lambda sc: self.feat_prop1.forward_ids(adj, embedding, ids, score=sc))
            self._milestone(feat)
            adj_refined = self._refine(adj, feat, score_x=sx)
            return self.feat_prop2(adj_refined, feat)
        # without dropout (evaluation) every padding row of the batch is the same vector: the first cell then runs on
        # the real rows plus ONE representative padding row instead of all n*R rows
        collapsed = not self.training
        rows = min(plan.m_real + 1, plan.m_tot) if collapsed else plan.m_tot
        feat, sx = self._first_cell(lambda sc: self.feat_prop1.forward_ids(adj, embedding, ids, plan=plan, rows=rows, score=sc))
        self._milestone(feat)
        adj_refined = self._refine(adj, feat, plan, collapsed, score_x=sx)
        return self.feat_prop2(adj_refined, feat, plan=plan, rows=plan.m_real)


# ------------------------------------------------------------------ Models/BiDAF/wrapper.py:229-276
class LSTM(nn.Module):
    """Present only for state_dict compatibility: BasicFCModel instantiates two of these
    (basic_fc_model.py:49-52) but GET's forward never runs them."""

    def __init__(self, input_size, hidden_size, batch_first=False, num_layers=1, bidirectional=False, dropout=0.2):
        super().__init__()
        self.rnn = nn.LSTM(input_size=input_size, hidden_size=hidden_size, num_layers=num_layers,
                           bidirectional=bidirectional, batch_first=batch_first)
        self.dropout = nn.Dropout(p=dropout)

    def forward(self, *a, **k):
        raise RuntimeError("get_amd: the LSTM encoders are dead code in GET and are not part of the HIP hot path")


# ------------------------------------------------------------------ thirdparty/two_branches_attention.py:112-148
class ConcatNotEqualSelfAtt(nn.Module):
    def __init__(self, inp_dim: int, out_dim: int, num_heads: int = 1):
        super().__init__()
        self.inp_dim, self.out_dim, self.num_heads = inp_dim, out_dim, num_heads
        _drop_caches_on_load(self)
        self.linear1 = nn.Linear(inp_dim, out_dim, bias=False)
        self.linear2 = nn.Linear(out_dim, num_heads, bias=False)

    def forward(self, left: torch.Tensor, right: torch.Tensor, mask: torch.Tensor, plan=None) -> Tuple[torch.Tensor, torch.Tensor]:
        """plan (ops.RaggedPlan): `right` / `mask` are node-compact (m_real, D) / (m_real,), weights come back compact."""
        if plan is None:
            assert left.size(0) == right.size(0), "Must same dimensions"
            assert len(left.size()) == 2 and len(right.size()) == 3
        assert self.inp_dim == (left.size(-1) + right.size(-1))
        return ops.concat_att(left, right, mask, self.linear1.weight, self.linear2.weight, plan)


class ConcatSelfAtt(ConcatNotEqualSelfAtt):
    """two_branches_attention.py:73-109 -- identical arithmetic to ConcatNotEqualSelfAtt."""


# ------------------------------------------------------------------ thirdparty/self_attention.py:51-100
class MultiHeadSelfAttentionICLR2017Extend(nn.Module):
    def __init__(self, inp_dim: int, out_dim: int, num_heads: int):
        super().__init__()
        self.inp_dim, self.out_dim, self.num_heads = inp_dim, out_dim, num_heads
        _drop_caches_on_load(self)
        self.linear1 = nn.Linear(inp_dim, out_dim, bias=False)
        self.linear2 = nn.Linear(out_dim, num_heads, bias=False)

    def forward(self, tsr: torch.Tensor, mask: torch.Tensor, return_att_weights=False):
        assert len(tsr.size()) == 3
        assert tsr.size(-1) == self.inp_dim
        attended, weights = ops.concat_att(None, tsr, mask, self.linear1.weight, self.linear2.weight)
        attended = attended.permute(0, 2, 1)       # (B, C, D)
        if return_att_weights:
            return attended, weights
        return attended


def init_weights(m):
    """torch_utils.py:379-388 (Linear branch): xavier-uniform weight, zero bias."""
    if type(m) == nn.Linear:
        nn.init.xavier_uniform_(m.weight)
        if hasattr(m.bias, "data"):
            m.bias.data.fill_(0)


class _HeadLinear(nn.Linear):
    """nn.Linear whose forward runs the library GEMM (keeps the reference's `out.0.weight` names)."""

    def forward(self, x):
        return ops.linear(x, self.weight, self.bias)


# ------------------------------------------------------------------ Models/FCWithEvidences/graph_based_semantic_structure.py:15-274
class Graph_basedSemantiStructure(nn.Module):
    """GET: claim GGNN + evidence GGNN with GSL + word- and evidence-level concat attention + head."""

    def __init__(self, params):
        super().__init__()
        _drop_caches_on_load(self)
        self._params = params
        self.embedding = self._make_default_embedding_layer(params)
        self.num_classes = params["num_classes"]
        self.fixed_length_right = params["fixed_length_right"]
        self.fixed_length_left = params["fixed_length_left"]
        self.use_claim_source = params["use_claim_source"]
        self.use_article_source = params["use_article_source"]
        self._use_cuda = params["cuda"]
        self.num_att_heads_for_words = params["num_att_heads_for_words"]
        self.num_att_heads_for_evds = params["num_att_heads_for_evds"]
        self.dropout_gnn = params["dropout_gnn"]
        self.dropout_left = params["dropout_left"]
        self.dropout_right = params["dropout_right"]
        self.hidden_size = params["hidden_size"]
        self.output_size = params["output_size"]
        self.gsl_rate = params["gsl_rate"]
        self.num_heads = 1
        H = self.hidden_size
        if self.use_claim_source:
            self.claim_source_embs = self._make_entity_embedding_layer(params["claim_source_embeddings"], freeze=False)
            self.claim_emb_size = params["claim_source_embeddings"].shape[1]
        if self.use_article_source:
            self.article_source_embs = self._make_entity_embedding_layer(params["article_source_embeddings"], freeze=False)
            self.article_emb_size = params["article_source_embeddings"].shape[1]
        D = params["embedding_output_dim"]
        # dead-but-present parameters of BasicFCModel.__init__ (basic_fc_model.py:49-52)
        self.bilstm = LSTM(input_size=D, hidden_size=H, num_layers=1, bidirectional=True, batch_first=True,
                           dropout=self.dropout_left)
        self.query_bilstm = LSTM(input_size=D, hidden_size=H, num_layers=1, bidirectional=True, batch_first=True,
                                 dropout=self.dropout_right)
        # live graph encoders (:52-55)
        self.ggnn4claim_1 = GGNN(in_features=D, out_features=H)
        self.ggnn_with_gsl = GGNN_with_GSL(input_dim=D, hidden_dim=H, output_dim=H, rate=self.gsl_rate,
                                           dropout=self.dropout_gnn)
        self.trans = Linear(2 * H, H)          # constructed, never used (:55)
        # attention (:223-249)
        self.self_att_word = ConcatNotEqualSelfAtt(inp_dim=2 * H, out_dim=H, num_heads=self.num_att_heads_for_words)
        evd_inp = H + self.num_att_heads_for_words * H
        if self.use_claim_source:
            evd_inp += self.claim_emb_size
        if self.use_article_source:
            evd_inp += self.article_emb_size
        self.self_att_evd = ConcatNotEqualSelfAtt(inp_dim=evd_inp, out_dim=H, num_heads=self.num_att_heads_for_evds)
        # head (:62-74): Linear -> Linear, no activation
        evd_input_size = H
        if self.use_claim_source:
            evd_input_size += self.claim_emb_size
        evd_input_size += H * self.num_att_heads_for_words * self.num_att_heads_for_evds
        if self.use_article_source:
            evd_input_size += self.article_emb_size * self.num_att_heads_for_evds
        self.out = nn.Sequential(_HeadLinear(evd_input_size, H), _HeadLinear(H, self.output_size))
        for m in self.out:
            nn.init.xavier_uniform_(m.weight)
            m.bias.data.fill_(0)

    # -- Models/base_model.py:144-162,184-188
    @staticmethod
    def _make_default_embedding_layer(_params) -> nn.Module:
        if isinstance(_params["embedding"], np.ndarray):
            _params["embedding_input_dim"] = _params["embedding"].shape[0]
            _params["embedding_output_dim"] = _params["embedding"].shape[1]
            return nn.Embedding.from_pretrained(embeddings=torch.Tensor(_params["embedding"]),
                                                freeze=_params["embedding_freeze"])
        return nn.Embedding(num_embeddings=_params["embedding_input_dim"],
                            embedding_dim=_params["embedding_output_dim"])

    @staticmethod
    def _make_entity_embedding_layer(matrix: np.ndarray, freeze: bool) -> nn.Module:
        return nn.Embedding.from_pretrained(embeddings=torch.Tensor(matrix), freeze=freeze)

    # -- forward (:76-125)
    def forward(self, query: torch.Tensor, document: torch.Tensor, verbose=False, **kargs):
        K = KeyWordSettings
        assert K.Query_lens in kargs and K.Doc_lens in kargs
        B, L = query.size()
        assert query.size(0) == document.size(0)
        batch_size, n, R = document.size()
        assert n == 30
        assert K.DocContentNoPaddingEvidence in kargs
        doc = kargs[K.DocContentNoPaddingEvidence]               # (B1, R) de-padded evidence node ids
        if K.DocLensIndices in kargs and kargs[K.DocLensIndices] is not None:
            d_lens = kargs[K.DocLensIndices][2]
            assert d_lens.shape[0] == doc.size(0)
        b1 = doc.size(0)
        n_max = kargs[K.FIXED_NUM_EVIDENCES]
        # The whole forward (and, through autograd, the whole backward) as ONE library call each when the model and
        # the batch qualify (get_amd/fused.py: fp32, frozen word table, float4-shaped widths with d <= h <= 320); the
        # module-by-module path below is the general one.
        from . import fused
        if _lib.gemm_mode() == "fp32x3p":
            ops.fp32x3p_guard(self)
        if fused.eligible(self, query, kargs):
            phi, word_w, evd_att_weight, plan = fused.forward(self, query, document, kargs)
            if kargs.get(K.OutputRankingKey, False):
                hw = self.num_att_heads_for_words
                word_att_weights = plan.to_padded(word_w) if plan is not None else word_w.view(b1, R, hw)
                return phi, (word_att_weights, evd_att_weight)
            return phi
        seg = ops.Segments(kargs[K.EvidenceCountPerQuery], b1, n_max)

        # claim branch (:144-155): GGNN -> masked mean over the unique claim nodes -> one row per pair
        def claim_branch():
            q_hid = self.ggnn4claim_1.forward_ids(kargs[K.Query_Adj], self.embedding, query)
            q = ops.masked_mean(q_hid, query, kargs[K.Query_lens])             # (B, H)
            return q, ops.seg_broadcast(q, seg)                                # (B1, H)

        # The claim branch is a chain of few-row launches (B x L rows) that is independent of the evidence branch until
        # the word attention: on a ROCm device it runs on a side stream underneath the evidence cells' GEMMs.  It is
        # issued AFTER the evidence branch so that autograd (which replays nodes newest-first, each on its forward
        # stream) also starts the claim backward before the evidence cells' backward.
        side = ops.side_stream(query.device) if ops.CLAIM_SIDE_STREAM and query.is_cuda else None
        if side is None:
            q_repr, query_repr = claim_branch()
        else:
            main = torch.cuda.current_stream(query.device)
            inputs_ready = torch.cuda.Event()
            inputs_ready.record(main)

        # evidence branch (:107): GGNN -> scorer + GSL -> GGNN on the refined graph.  A PackedAdj that carries a
        # node-compact plan (NativeBatch) takes the fast path that skips the padding nodes wherever they cannot
        # influence a result (ops.RaggedPlan); anything else runs the reference's padded layout.
        d_adj = kargs[K.Evd_Docs_Adj]
        plan = d_adj.plan if isinstance(d_adj, PackedAdj) else None
        doc_out = self.ggnn_with_gsl.forward_ids(d_adj, self.embedding, doc, plan=plan)

        if side is not None:
            side.wait_event(inputs_ready)
            with torch.cuda.stream(side):
                q_repr, query_repr = claim_branch()
            main.wait_stream(side)
            q_repr.record_stream(main)
            query_repr.record_stream(main)
            if query_repr.requires_grad:
                # the claim branch's backward will run on the side stream: join it at the end of the backward pass
                # whether or not any other side-stream work (ops._side_wgrad) happens in that pass
                dev_ = query.device
                query_repr.register_hook(lambda g, _d=dev_: ops.side_mark_backward(_d))

        # word-level attention (:173-193); the claim vector WITHOUT its source embedding (:110)
        if plan is None:
            att, word_att_weights = self.self_att_word(query_repr, doc_out, doc >= 1)
        else:
            att, word_w = self.self_att_word(query_repr, doc_out, plan.maskf[:plan.m_real], plan=plan)
            word_att_weights = None
        avg = torch.flatten(att, start_dim=1)                                   # (B1, H*hw), head fastest

        # evidence-level attention (:195-221).  Its left input is row 0 of pad_right([claim source | query_repr]), i.e. the
        # claim's own vector (zeros for a claim without evidences) -- taken per claim instead of broadcasting to B1 pairs,
        # padding to (B, n, X) and slicing slot 0
        left_claim = q_repr
        if self.use_claim_source:
            claim_embs = self.claim_source_embs(kargs[K.QuerySources].long()).squeeze(1)
            left_claim = torch.cat([claim_embs, q_repr], dim=-1)                # source first (:116)
        new_left = left_claim * seg.has                                         # (B, X)
        # pad_right(avg) ++ article_source_embs(src with -1 -> 0) and the slot mask, one launch (:157-170, :195-215)
        padded_avg, mask = ops.evd_assemble(avg, self.article_source_embs.weight if self.use_article_source else None, seg,
                                            kargs[K.DocSources] if self.use_article_source else None, document)
        attended_avg, evd_att_weight = self.self_att_evd(new_left.contiguous(), padded_avg, mask)
        output = torch.cat([new_left, torch.flatten(attended_avg, start_dim=1)], dim=-1)   # (:251-267)
        phi = self.out(output)
        if kargs.get(K.OutputRankingKey, False):
            if word_att_weights is None:
                word_att_weights = plan.to_padded(word_w)          # (B1, R, hw), zeros at the padding nodes
            return phi, (word_att_weights, evd_att_weight)
        return phi

    def predict(self, query: torch.Tensor, doc: torch.Tensor, verbose: bool = False, **kargs):
        self.train(False)
        assert query.size(0) == doc.size(0)
        return self(query, doc, **kargs)

    # ragged helpers kept under the reference's names (basic_fc_model.py:80-121)
    def _pad_left_tensor(self, left_tsr: torch.Tensor, **kargs):
        cnt = kargs[KeyWordSettings.EvidenceCountPerQuery]
        b1 = int(cnt.sum().item())
        return ops.seg_broadcast(left_tsr, ops.Segments(cnt, b1, kargs.get(KeyWordSettings.FIXED_NUM_EVIDENCES, 30)))

    @classmethod
    def _pad_right_tensor(cls, tsr: torch.Tensor, **kargs):
        cnt = kargs[KeyWordSettings.EvidenceCountPerQuery]
        return ops.seg_pad(tsr, ops.Segments(cnt, tsr.size(0), kargs[KeyWordSettings.FIXED_NUM_EVIDENCES]))
