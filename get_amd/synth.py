"""Seeded synthetic Snopes/PolitiFact-shaped batches and reference-named parameter sets.

Pure numpy (``default_rng`` streams are platform independent), no device work:
``bench.py`` uses it to make the workload BASELINE.json names, the tests and the
golden-vector script use it so that the reference, the oracle and the HIP path
all see byte-identical inputs and parameters.  Shapes and distributions follow
SURVEY.md section 8(d); tensor names follow the reference's ``state_dict``
(SURVEY.md section 8(b)).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, Optional

import numpy as np


@dataclass
class SynthConfig:
    batch: int = 32                 # claims per batch (B)
    n_evd: int = 30                 # evidences per claim; <=0 -> ragged U[1, fixed_num_evidences]
    fixed_num_evidences: int = 30   # hard-wired n of graph_based_semantic_structure.py:93
    len_left: int = 30              # L  (interactions.py:303)
    len_right: int = 100            # R
    emb_dim: int = 300              # D
    hidden: int = 300               # H
    word_heads: int = 5
    evd_heads: int = 2
    window: int = 3
    gsl_rate: float = 0.6
    vocab: int = 20000
    n_article_src: int = 3000
    n_claim_src: int = 500
    src_dim: int = 128
    use_claim_source: bool = False
    use_article_source: bool = True
    num_classes: int = 2
    evd_counts: Optional[list] = None   # explicit per-claim counts (overrides n_evd)

    def model_params(self, embedding: np.ndarray, article_src: np.ndarray, claim_src: np.ndarray) -> dict:
        """The ``params`` dict MasterFC/master_get.py:118-144 hands to the model ctor."""
        return {
            "embedding": embedding, "embedding_freeze": True, "num_classes": self.num_classes,
            "fixed_length_left": self.len_left, "fixed_length_right": self.len_right,
            "use_claim_source": self.use_claim_source, "claim_source_embeddings": claim_src,
            "use_article_source": self.use_article_source, "article_source_embeddings": article_src,
            "cuda": 0, "num_att_heads_for_words": self.word_heads, "num_att_heads_for_evds": self.evd_heads,
            "dropout_gnn": 0.2, "dropout_left": 0.2, "dropout_right": 0.2, "hidden_size": self.hidden,
            "output_size": self.num_classes, "gsl_rate": self.gsl_rate,
        }


# Evidences per claim on Snopes: histogram over counts 1..26 of the 782 claims of the reference's
# formatted_data/declare/Snopes/mapped_data/5fold/test_0.tsv (mean 6.92, max 26 -- SURVEY.md 8(d) "realistic series").
SNOPES_EVD_HIST = (107, 94, 79, 69, 60, 44, 57, 32, 38, 28, 23, 23, 18, 15, 14, 14, 14, 6, 13, 4, 9, 5, 6, 6, 2, 2)


def snopes_evidence_counts(rng: np.random.Generator, n_claims: int) -> np.ndarray:
    """Evidence counts drawn from the empirical Snopes histogram."""
    h = np.asarray(SNOPES_EVD_HIST, dtype=np.float64)
    return (rng.choice(len(h), size=n_claims, p=h / h.sum()) + 1).astype(np.int64)


def _zipf_probs(n: int, s: float = 1.1) -> np.ndarray:
    p = 1.0 / np.arange(1, n + 1, dtype=np.float64) ** s
    return p / p.sum()


def make_tokens(rng: np.random.Generator, n_texts: int, fixed_length: int, vocab: int,
                min_len: int, max_len: int) -> tuple:
    """Raw (not yet de-duplicated) token sequences, post-padded with 0.

    ids ~ Zipf(1.1) over [2, vocab) with 3 % replaced by OOV id 1.
    Returns (tokens[n_texts, fixed_length] int32, lengths[n_texts] int32).
    """
    lens = rng.integers(min_len, max_len + 1, size=n_texts).astype(np.int32)
    ranks = rng.choice(vocab - 2, size=(n_texts, fixed_length), p=_zipf_probs(vocab - 2))
    toks = (ranks + 2).astype(np.int32)
    toks[rng.random((n_texts, fixed_length)) < 0.03] = 1
    toks[np.arange(fixed_length)[None, :] >= lens[:, None]] = 0
    return toks, lens


def make_raw_batch(cfg: SynthConfig, seed: int) -> Dict[str, np.ndarray]:
    """One synthetic mini-batch in raw-token form (input of the graph build).

    Keys: claim_tokens (B,L) i32, claim_len (B), evd_tokens (B1,R) i32, evd_len (B1),
    evd_counts (B) i64, doc_sources (B,n) i64 with -1 padding, query_sources (B,1) i64,
    labels (B) i64.
    """
    rng = np.random.default_rng(seed)
    B, n = cfg.batch, cfg.fixed_num_evidences
    if cfg.evd_counts is not None:
        counts = np.asarray(cfg.evd_counts, dtype=np.int64)
        assert counts.shape == (B,)
    elif cfg.n_evd > 0:
        counts = np.full((B,), cfg.n_evd, dtype=np.int64)
    else:
        counts = rng.integers(1, n + 1, size=B).astype(np.int64)
    b1 = int(counts.sum())
    claim_tokens, claim_len = make_tokens(rng, B, cfg.len_left, cfg.vocab, min(5, cfg.len_left), cfg.len_left)
    evd_tokens, evd_len = make_tokens(rng, b1, cfg.len_right, cfg.vocab, cfg.len_right, cfg.len_right)
    doc_sources = np.full((B, n), -1, dtype=np.int64)
    for b in range(B):
        doc_sources[b, :counts[b]] = rng.integers(0, cfg.n_article_src, size=counts[b])
    query_sources = rng.integers(0, cfg.n_claim_src, size=(B, 1)).astype(np.int64)
    labels = (rng.random(B) < 0.27).astype(np.int64)
    return dict(claim_tokens=claim_tokens, claim_len=claim_len, evd_tokens=evd_tokens, evd_len=evd_len,
                evd_counts=counts, doc_sources=doc_sources, query_sources=query_sources, labels=labels)


def make_embeddings(cfg: SynthConfig, seed: int):
    """GloVe-like word table (row 0 = constant 0.1) and source tables."""
    rng = np.random.default_rng(seed + 7919)
    emb = (rng.standard_normal((cfg.vocab, cfg.emb_dim)) * 0.4).astype(np.float32)
    emb[0, :] = 0.1
    art = rng.uniform(-0.2, 0.2, size=(cfg.n_article_src, cfg.src_dim)).astype(np.float32)
    clm = rng.uniform(-0.2, 0.2, size=(cfg.n_claim_src, cfg.src_dim)).astype(np.float32)
    return emb, art, clm


def state_dict_shapes(cfg: SynthConfig) -> Dict[str, tuple]:
    """Name -> shape of every LIVE float parameter (reference state_dict names)."""
    D, H, hw, he = cfg.emb_dim, cfg.hidden, cfg.word_heads, cfg.evd_heads
    sh: Dict[str, tuple] = {}

    def cell(prefix, din, dout):
        sh[prefix + "proj.linear.weight"] = (dout, din)
        for g in ("z0", "z1", "r0", "r1", "h0", "h1"):
            sh[prefix + f"linear{g}.linear.weight"] = (dout, dout)
            sh[prefix + f"linear{g}.linear.bias"] = (dout,)

    cell("ggnn4claim_1.", D, H)
    cell("ggnn_with_gsl.feat_prop1.", D, H)
    cell("ggnn_with_gsl.word_scorer1.", H, 1)
    cell("ggnn_with_gsl.feat_prop2.", H, H)
    sh["self_att_word.linear1.weight"] = (H, 2 * H)
    sh["self_att_word.linear2.weight"] = (hw, H)
    left = H + (cfg.src_dim if cfg.use_claim_source else 0)
    right = H * hw + (cfg.src_dim if cfg.use_article_source else 0)
    sh["self_att_evd.linear1.weight"] = (H, left + right)
    sh["self_att_evd.linear2.weight"] = (he, H)
    sh["out.0.weight"] = (H, left + right * he)
    sh["out.0.bias"] = (H,)
    sh["out.1.weight"] = (cfg.num_classes, H)
    sh["out.1.bias"] = (cfg.num_classes,)
    return sh


def make_state_dict(cfg: SynthConfig, seed: int) -> Dict[str, np.ndarray]:
    """Seeded numpy values for every live parameter (fan-in scaled normals, small biases).

    Not the reference's init scheme -- a deterministic, platform-independent
    parameter set that the reference, oracle and HIP modules all load, so that
    fixtures need not store 15 MB of weights.
    """
    rng = np.random.default_rng(seed + 104729)
    out = {}
    for name, shape in state_dict_shapes(cfg).items():
        if name.endswith("bias"):
            out[name] = rng.uniform(-0.05, 0.05, size=shape).astype(np.float32)
        else:
            scale = 1.0 if shape[-1] == 1 else (1.0 / np.sqrt(shape[-1]))
            out[name] = (rng.standard_normal(shape) * scale).astype(np.float32)
    return out
