"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the GET hot path.

A plain restatement of the reference algorithm in torch-CPU / numpy ops, written
from the behaviour of the reference (file:line cited per function, paths
relative to the upstream CRIPAC-DIG/GET tree).  It is differentiable through
``torch.autograd`` so the same code yields oracle gradients.  It is pinned
against golden vectors captured from the imported reference by
``oracle/make_golden.py`` (see ``tests/test_oracle_golden.py``).

Parity status: PINNED by fixtures generated from the reference itself in the
build container (the reference holds no tests of its own for this path).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch


# ---------------------------------------------------------------------------
# a1  graph build -- interactions.py:334-351 (convert_text) and :11-18
#     (_laplacian_normalize)
# ---------------------------------------------------------------------------
def convert_text(raw_text: Sequence[int], fixed_length: int, length: int, window: int):
    """Sliding-window word graph of one text.

    Follows interactions.py:334-351: nodes are the distinct tokens of
    ``raw_text[:length]`` in first-occurrence order; for every position i every
    position j with ``i-window+1 <= j <= i+window-1`` (clipped to [0,length))
    links node(tok_i) -> node(tok_j); the binary matrix is normalised as
    D^-1/2 A D^-1/2 with zero-degree rows mapped to 0 (interactions.py:11-18).

    Returns (node_ids[fixed_length] int64, adj[fixed_length,fixed_length] f64, n_nodes).
    """
    toks = [int(t) for t in raw_text[:length]]
    node_of: Dict[int, int] = {}
    nodes: List[int] = []
    for t in toks:
        if t not in node_of:
            node_of[t] = len(nodes)
            nodes.append(t)
    n = len(nodes)
    a = np.zeros((fixed_length, fixed_length), dtype=np.float64)
    for i, t in enumerate(toks):
        lo, hi = max(i - window + 1, 0), min(i + window, length)
        for j in range(lo, hi):
            a[node_of[t], node_of[toks[j]]] = 1.0
    deg = a.sum(axis=1)
    with np.errstate(divide="ignore"):
        dinv = np.where(deg > 0, deg ** -0.5, 0.0)
    # (A . D^-1/2)^T . D^-1/2  ==  D^-1/2 A^T D^-1/2   (interactions.py:18)
    adj = (a * dinv[None, :]).T * dinv[None, :]
    ids = np.zeros((fixed_length,), dtype=np.int64)
    ids[:n] = nodes
    return ids, adj, n


def pack_adjacency(adj: np.ndarray):
    """Dense (R,R) -> (bits[R, ceil(R/64)] uint64, nonzero mask).  Helper for tests."""
    r = adj.shape[-1]
    w = (r + 63) // 64
    bits = np.zeros(adj.shape[:-1] + (w,), dtype=np.uint64)
    nz = adj != 0
    for j in range(r):
        bits[..., j // 64] |= nz[..., j].astype(np.uint64) << np.uint64(j % 64)
    return bits


# ---------------------------------------------------------------------------
# a2  GGNN cell -- Models/BiDAF/wrapper.py:174-208
# ---------------------------------------------------------------------------
def ggnn_cell(adj: torch.Tensor, x: torch.Tensor, p: Dict[str, torch.Tensor], prefix: str = "", drop_p: float = 0.0, keep=None):
    """Gated graph cell.  wrapper.py:188-208.  drop_p > 0 = training mode: the cell's input dropout (:189-190) with
    torch's own mask (used only to TIME a training step on the CPU).  keep = (mask, p): the same dropout with a GIVEN keep
    mask (x * mask / (1 - p), what nn.Dropout computes for that draw) -- training-mode parity checks replay the product's
    stateless mask through it.

    ``p`` holds the reference state_dict names below ``prefix``:
    proj.linear.weight, linear{z,r,h}{0,1}.linear.{weight,bias}.
    """
    g = lambda n: p[prefix + n]
    if keep is not None:
        x = x * keep[0].to(x.dtype) / (1.0 - float(keep[1]))
    elif drop_p > 0.0:
        x = torch.nn.functional.dropout(x, drop_p, training=True)
    xp = x @ g("proj.linear.weight").t()
    a = adj @ xp
    lin = lambda name, t: t @ g(name + ".linear.weight").t() + g(name + ".linear.bias")
    z = torch.sigmoid(lin("linearz0", a) + lin("linearz1", xp))
    r = torch.sigmoid(lin("linearr0", a) + lin("linearr1", xp))
    h = torch.tanh(lin("linearh0", a) + lin("linearh1", r * xp))
    return h * z + xp * (1 - z)


# ---------------------------------------------------------------------------
# a3  GSL -- Models/BiDAF/wrapper.py:215-227
# ---------------------------------------------------------------------------
def gsl_keep(score: torch.Tensor, rate: float, n_padded: int) -> torch.Tensor:
    """Keep-set of the top-k nodes, k = int(rate * padded length) (wrapper.py:216-219).

    score (B,R) -> bool (B,R).  Ties are broken towards the LOWER index (our
    convention; the reference's torch.topk tie order is unspecified).
    """
    k = int(rate * n_padded)
    b, r = score.shape
    s = score.detach()
    # rank_i = #{j : s_j > s_i or (s_j == s_i and j < i)}
    gt = s[:, None, :] > s[:, :, None]
    eq = (s[:, None, :] == s[:, :, None]) & (torch.arange(r)[None, None, :] < torch.arange(r)[None, :, None])
    rank = (gt | eq).sum(-1)
    return rank < k


def gsl_refine(adj: torch.Tensor, score: torch.Tensor, rate: float, keep: Optional[torch.Tensor] = None):
    """adj * mask with mask = 1 on kept rows UNION kept columns (wrapper.py:221-225).

    No renormalisation, no gradient through the selection.
    """
    if keep is None:
        keep = gsl_keep(score.squeeze(-1), rate, adj.shape[-1])
    mask = (keep[:, :, None] | keep[:, None, :]).to(adj.dtype)
    return adj * mask, keep


# ---------------------------------------------------------------------------
# a4  GGNN_with_GSL -- Models/BiDAF/wrapper.py:165-172
# ---------------------------------------------------------------------------
def ggnn_with_gsl(adj, feat, p, prefix, rate, keep_override=None, return_aux=False, drop_p: float = 0.0, drop_keep=None):
    dk = drop_keep or {}
    f1 = ggnn_cell(adj, feat, p, prefix + "feat_prop1.", drop_p, dk.get("cell1"))
    score = ggnn_cell(adj, f1, p, prefix + "word_scorer1.", drop_p, dk.get("scorer"))
    adj_r, keep = gsl_refine(adj, score, rate, keep_override)
    out = ggnn_cell(adj_r, f1, p, prefix + "feat_prop2.", drop_p, dk.get("cell2"))
    if return_aux:
        return out, dict(feat1=f1, score=score.squeeze(-1), keep=keep)
    return out


# ---------------------------------------------------------------------------
# a5  ConcatNotEqualSelfAtt -- thirdparty/two_branches_attention.py:121-148
# ---------------------------------------------------------------------------
def concat_att(left, right, mask, w1, w2):
    """softmax over the sequence axis of W2.tanh(W1.[left || right_t]); attended = right^T w."""
    b, l, _ = right.shape
    tsr = torch.cat([left[:, None, :].expand(b, l, left.shape[-1]), right], dim=-1)
    e = torch.tanh(tsr @ w1.t()) @ w2.t()                      # (B,L,C)
    e = e.masked_fill((mask == 0)[:, :, None], float("-inf"))
    w = torch.softmax(e, dim=1)
    return right.transpose(1, 2) @ w, w                        # (B,D,C), (B,L,C)


# ---------------------------------------------------------------------------
# a6  MultiHeadSelfAttentionICLR2017Extend -- thirdparty/self_attention.py:75-100
# ---------------------------------------------------------------------------
def self_att_extend(tsr, mask, w1, w2):
    e = torch.tanh(tsr @ w1.t()) @ w2.t()
    e = e.masked_fill((mask == 0)[:, :, None], float("-inf"))
    w = torch.softmax(e, dim=1)
    return (tsr.transpose(1, 2) @ w).transpose(1, 2), w        # (B,C,D), (B,L,C)


# ---------------------------------------------------------------------------
# a8  ragged helpers -- Models/FCWithEvidences/basic_fc_model.py:80-121
# ---------------------------------------------------------------------------
def pad_left(t: torch.Tensor, counts: Sequence[int]):
    """(B,H) -> (sum counts, H): row b repeated counts[b] times."""
    return torch.cat([t[b:b + 1].expand(int(c), t.shape[1]) for b, c in enumerate(counts)], dim=0)


def pad_right(t: torch.Tensor, counts: Sequence[int], n_max: int):
    """(sum counts, X) -> (B, n_max, X), zero padded per claim."""
    out, last = [], 0
    for c in counts:
        c = int(c)
        seg = t[last:last + c]
        out.append(torch.cat([seg, seg.new_zeros(n_max - c, t.shape[1])], dim=0))
        last += c
    return torch.stack(out, dim=0)


# ---------------------------------------------------------------------------
# a7  Graph_basedSemantiStructure.forward -- graph_based_semantic_structure.py:76-125
# ---------------------------------------------------------------------------
def model_forward(p: Dict[str, torch.Tensor], cfg: dict, query, document, query_adj, doc_ids, doc_adj,
                  query_lens, evd_counts, doc_sources, query_sources=None, keep_override=None,
                  return_aux=False, drop_p: float = 0.0, drop_keep=None):
    """Forward of the GET model from a reference-named state dict ``p`` (evaluation mode; drop_p > 0 switches the
    four cells' input dropout on -- dropout_gnn = 0.2 in the reference -- for TIMING a training step only;
    drop_keep = {"claim" | "cell1" | "scorer" | "cell2": (keep mask shaped like that cell's input, p)} replays GIVEN masks).

    query (B,L) node ids; document (B,n,R) ids (only its evidence-slot mask is
    used, :215); doc_ids (B1,R) de-padded evidence node ids; adjacencies dense;
    evd_counts python ints.  Returns phi (B,out), word weights (B1,R,hw),
    evidence weights (B,n,he).
    """
    emb = p["embedding.weight"]
    counts = [int(c) for c in evd_counts]
    n_max = document.shape[1]
    # claim branch (:144-155)
    q_mask = (query > 0).to(emb.dtype)[:, :, None]
    q_h = ggnn_cell(query_adj.to(emb.dtype), emb[query.long()], p, "ggnn4claim_1.", drop_p, (drop_keep or {}).get("claim"))
    q_repr = (q_h * q_mask).sum(1) / query_lens.to(emb.dtype)[:, None]
    q_rep_pairs = pad_left(q_repr, counts)
    # evidence branch (:107)
    aux = {}
    doc_out = ggnn_with_gsl(doc_adj.to(emb.dtype), emb[doc_ids.long()], p, "ggnn_with_gsl.",
                            cfg["gsl_rate"], keep_override, return_aux=return_aux, drop_p=drop_p, drop_keep=drop_keep)
    if return_aux:
        doc_out, aux = doc_out
    # word-level attention (:173-193); claim vector WITHOUT source embedding (:110)
    att, word_w = concat_att(q_rep_pairs, doc_out, doc_ids >= 1,
                             p["self_att_word.linear1.weight"], p["self_att_word.linear2.weight"])
    avg = att.flatten(1)                                        # head index fastest (:191)
    left = q_rep_pairs
    if cfg.get("use_claim_source"):
        ce = p["claim_source_embs.weight"][query_sources.long()].squeeze(1)
        left = torch.cat([pad_left(ce, counts), q_rep_pairs], dim=-1)   # source first (:116)
    # evidence-level attention (:195-221)
    new_left = pad_right(left, counts, n_max)[:, 0, :]
    padded = pad_right(avg, counts, n_max)
    evd_mask = (document.sum(-1) >= 1).to(emb.dtype)
    if cfg.get("use_article_source"):
        src = doc_sources.clone()
        src[src == -1] = 0                                      # (:166-168)
        padded = torch.cat([padded, p["article_source_embs.weight"][src.long()]], dim=-1)
    att2, evd_w = concat_att(new_left, padded, evd_mask,
                             p["self_att_evd.linear1.weight"], p["self_att_evd.linear2.weight"])
    final = torch.cat([new_left, att2.flatten(1)], dim=-1)      # (:251-267)
    hid = final @ p["out.0.weight"].t() + p["out.0.bias"]
    phi = hid @ p["out.1.weight"].t() + p["out.1.bias"]         # no activation between (:69-72)
    if return_aux:
        aux.update(doc_out=doc_out, q_repr=q_repr, avg=avg, final=final)
        return phi, word_w, evd_w, aux
    return phi, word_w, evd_w


def cross_entropy(phi, labels):
    """losses.py:29-32 -- nn.CrossEntropyLoss()(pred, labels.long()), mean reduction."""
    return torch.nn.functional.cross_entropy(phi, labels.long())


def adam_step(params: Dict[str, torch.Tensor], grads: Dict[str, Optional[torch.Tensor]], state: dict,
              lr=1e-4, weight_decay=1e-3, betas=(0.9, 0.999), eps=1e-8):
    """One torch.optim.Adam step (L2-style weight decay added to the gradient),
    Fitting/FittingFC/declare_fitter.py:58-61.  Parameters whose grad is None are
    skipped entirely (no decay), as torch.optim.Adam does."""
    state["t"] = state.get("t", 0) + 1
    t = state["t"]
    out = {}
    for k, w in params.items():
        g = grads.get(k)
        if g is None:
            out[k] = w
            continue
        g = g + weight_decay * w
        m = state.setdefault("m", {}).get(k, torch.zeros_like(w))
        v = state.setdefault("v", {}).get(k, torch.zeros_like(w))
        m = betas[0] * m + (1 - betas[0]) * g
        v = betas[1] * v + (1 - betas[1]) * g * g
        state["m"][k], state["v"][k] = m, v
        mh = m / (1 - betas[0] ** t)
        vh = v / (1 - betas[1] ** t)
        out[k] = w - lr * mh / (vh.sqrt() + eps)
    return out
