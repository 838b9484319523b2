"""Host logic of get_amd.graph_cache on CPU tensors: container invariants, key lookup,
persistence and batch assembly.  The packed arrays come from the oracle's convert_text (no device work here);
the device build is checked against the same oracle in tests/test_gpu_model.py."""
import numpy as np
import pytest
import torch

from get_amd.graph_cache import CachedBatcher, GraphCache
from get_amd.keywords import KeyWordSettings as K
from oracle import get_oracle as O


def _pack(texts, lengths, r, window):
    n = len(texts)
    w = (r + 63) // 64
    ids = np.zeros((n, r), np.int32)
    nn = np.zeros((n,), np.int32)
    bits = np.zeros((n, r, w), np.uint64)
    dinv = np.zeros((n, r), np.float32)
    dense = np.zeros((n, r, r), np.float64)
    for i, (t, ln) in enumerate(zip(texts, lengths)):
        words, adj, k = O.convert_text([int(x) for x in t], r, int(ln), window)
        ids[i], nn[i], dense[i] = np.asarray(words), k, np.asarray(adj)
        nz = dense[i] != 0
        for a in range(r):
            for b in np.nonzero(nz[a])[0]:
                bits[i, a, b // 64] |= np.uint64(1) << np.uint64(b % 64)
        deg = nz.sum(1)
        dinv[i] = np.where(deg > 0, 1.0 / np.sqrt(np.maximum(deg, 1)), 0.0)
    return ids, nn, bits.view(np.int64), dinv, dense


def _corpus(seed, n, r, vocab=50):
    rng = np.random.default_rng(seed)
    lengths = rng.integers(1, r + 1, size=n)
    toks = np.zeros((n, r), np.int64)
    for i in range(n):
        toks[i, :lengths[i]] = rng.integers(1, vocab, size=lengths[i])
    return toks, lengths


def test_container_roundtrip_and_lookup(tmp_path):
    toks, lens = _corpus(1, 9, 100)
    keys = [f"doc{i}" for i in range(9)]
    ids, nn, bits, dinv, _ = _pack(toks, lens, 100, 3)
    c = GraphCache.from_arrays(keys, ids, nn, bits, dinv, window=3)
    assert len(c) == 9 and c.fixed_length == 100 and c.window == 3
    assert c.nbytes() == 9 * (100 * 4 + 4 + 100 * 2 * 8 + 100 * 4)          # 2.4 KB per text
    assert np.array_equal(c.rows(["doc3", "doc0", "doc3"]), [3, 0, 3])
    with pytest.raises(KeyError, match="doc99"):
        c.rows(["doc1", "doc99"])
    with pytest.raises(ValueError, match="duplicate"):
        GraphCache.from_arrays(["a", "a"], ids[:2], nn[:2], bits[:2], dinv[:2], window=3)
    p = str(tmp_path / "evd.npz")
    c.save(p)
    d = GraphCache.load(p)
    assert d.keys == keys and d.window == 3 and d.fixed_length == 100
    for a, b in ((c.node_ids, d.node_ids), (c.n_nodes, d.n_nodes), (c.bits, d.bits), (c.dinv, d.dinv)):
        assert torch.equal(a, b)
    assert np.array_equal(d.n_nodes_host, nn)
    adj, g_ids, g_nn = d.gather([4, 2])
    assert adj.n == 2 and adj.r == 100 and torch.equal(g_ids, c.node_ids[[4, 2]]) and torch.equal(g_nn, c.n_nodes[[4, 2]])
    assert torch.equal(adj.bits, c.bits[[4, 2]]) and adj.plan is None


def test_int_keys_and_shape_checks():
    toks, lens = _corpus(2, 4, 30)
    ids, nn, bits, dinv, _ = _pack(toks, lens, 30, 3)
    c = GraphCache.from_arrays([11, 7, 5, 3], ids, nn, bits, dinv, window=3)
    assert np.array_equal(c.rows([5, 11]), [2, 0])
    with pytest.raises(AssertionError):
        GraphCache.from_arrays([1, 2, 3], ids, nn, bits, dinv, window=3)          # one key short
    with pytest.raises(ValueError, match="empty text"):
        GraphCache.build([0], np.zeros((1, 30), np.int64), np.zeros((1,), np.int64), 3, device="cpu")


def test_cached_batcher_assembles_reference_layout():
    """Same de-padded, claim-major layout as the fitter builds (char_man_fitter_query_repr1.py:204-250)."""
    ctoks, clens = _corpus(3, 3, 30)
    etoks, elens = _corpus(4, 8, 100)
    cc = GraphCache.from_arrays(["q0", "q1", "q2"], *_pack(ctoks, clens, 30, 3)[:4], window=3)
    ec = GraphCache.from_arrays(list(range(100, 108)), *_pack(etoks, elens, 100, 3)[:4], window=3)
    rel = {"q0": [103, 100], "q1": [107], "q2": [101, 102, 104, 105]}
    bt = CachedBatcher(cc, ec, rel, n_max=5, compact=False)
    q_ids, document, kargs = bt.inputs(["q2", "q0"])
    assert q_ids.shape == (2, 30) and torch.equal(q_ids, cc.node_ids[[2, 0]])
    order = [1, 2, 4, 5, 3, 0]                                                   # q2's evidences, then q0's
    assert torch.equal(kargs[K.DocContentNoPaddingEvidence], ec.node_ids[order])
    assert kargs[K.EvidenceCountPerQuery].tolist() == [4, 2] and kargs[K.FIXED_NUM_EVIDENCES] == 5
    assert document.shape == (2, 5, 100)
    assert torch.equal(document[0, :4], ec.node_ids[[1, 2, 4, 5]]) and int(document[0, 4].abs().sum()) == 0
    assert torch.equal(document[1, :2], ec.node_ids[[3, 0]]) and int(document[1, 2:].abs().sum()) == 0
    assert torch.equal(kargs[K.Query_lens], cc.n_nodes[[2, 0]])
    assert kargs[K.Evd_Docs_Adj].n == 6 and kargs[K.Query_Adj].n == 2
    with pytest.raises(ValueError, match="more than n_max"):
        CachedBatcher(cc, ec, rel, n_max=3)
