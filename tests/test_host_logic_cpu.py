"""Host-side logic that feeds reported numbers (no device work): bench.py's FLOP accounting against SURVEY.md 8(d)
and the synthetic evidence-count distributions."""
import numpy as np

from get_amd.synth import SNOPES_EVD_HIST, SynthConfig, snopes_evidence_counts


def test_bench_flop_accounting_matches_survey():
    import bench
    cfg = SynthConfig()
    f = bench.flops_per_pair(cfg, nnz_per_graph=427.0, real_nodes=64.6)
    assert abs(f["fwd"] / 1e6 - 271.0) < 2.0            # ~271 MFLOP forward, minimal formulation
    assert abs(f["fwd_bwd"] / 1e9 - 0.80) < 0.01        # ~0.80 GFLOP forward + backward
    assert 0.5 < f["executed"] / 1e9 < 0.6              # node-compact layout at 64.6 real nodes of 100
    padded = bench.flops_per_pair(cfg, 427.0, real_nodes=100.0)
    assert abs(padded["executed"] - padded["fwd_bwd"]) < 1e-3 * padded["fwd_bwd"]     # no padding: nothing skipped (the scorer has no backward)


def test_snopes_evidence_histogram():
    """SURVEY 8(d) realistic series: mean 6.9 evidences per claim, max 26 (the reference's Snopes test_0 fold)."""
    h = np.asarray(SNOPES_EVD_HIST, dtype=np.float64)
    assert len(h) == 26 and h.sum() == 782
    assert abs((h * np.arange(1, 27)).sum() / h.sum() - 6.92) < 0.01
    c = snopes_evidence_counts(np.random.default_rng(0), 20000)
    assert c.min() >= 1 and c.max() <= 26 and abs(c.mean() - 6.92) < 0.15


def test_reference_tensor_shim_depads_like_the_fitter_on_cpu():
    """get_amd.batch.kargs_from_reference_tensors in its plain-tensor form (what runs without the HIP library, and what
    gh_ref_depad is compared with on the GPU): the padded (B, n, R) ids and (B, n, R, R) float64 adjacency come back de-padded
    claim-major exactly as the fitter's per-claim loop leaves them (char_man_fitter_query_repr1.py:204-223, restated in
    oracle/assemble.py), including a claim whose count is below n and a claim with a single evidence."""
    import torch

    from get_amd.batch import kargs_from_reference_tensors
    from get_amd.synth import make_raw_batch
    from oracle import get_oracle as O
    from oracle.assemble import assemble_inputs
    cfg = SynthConfig(batch=4, emb_dim=32, hidden=32, vocab=200, n_article_src=10, n_claim_src=5, src_dim=8, evd_counts=[3, 30, 1, 9])
    raw = make_raw_batch(cfg, 17)
    inp = assemble_inputs(raw, cfg, O.convert_text)
    B, n, R = inp["document"].shape
    adj_padded = np.zeros((B, n, R, R))
    last = 0
    for b, c in enumerate(inp["evd_counts"]):
        adj_padded[b, :c] = inp["doc_adj"][last:last + c]
        last += c
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    k = kargs_from_reference_tensors(T(inp["query_lens"]), T(inp["document"]), T(adj_padded), T(inp["query_adj"]),
                                     T(np.asarray(inp["evd_counts"])), T(inp["doc_sources"]), T(inp["query_sources"]), n_max=n)
    assert np.array_equal(k["doc_content_without_padding_evidences"].numpy(), inp["doc_ids"])
    assert np.array_equal(k["docs_adj"].numpy(), inp["doc_adj"])
    assert k["fixed_num_evidences"] == n and k["docs_adj"].shape[0] == int(np.sum(inp["evd_counts"]))


def test_arena_pool_converges_to_one_buffer_per_kind():
    """get_amd.fused._ArenaPool (persistent activation arenas of the composite path): a returned buffer is re-used for requests
    it holds with at most 30 % slack, a larger request supersedes (drops) the free buffers it makes redundant, and two kinds of
    arenas whose sizes are far apart (forward / backward) never take each other's buffers."""
    import torch
    from get_amd.fused import _ArenaPool, _arena_floats
    pool, dev = _ArenaPool(), torch.device("cpu")
    fwd, bwd = 40 << 20, 24 << 20            # floats
    a = pool.take(fwd, dev)
    b = pool.take(bwd, dev)
    assert a.numel() == _arena_floats(fwd) and b.numel() == _arena_floats(bwd) and a.data_ptr() != b.data_ptr()
    pool.give(a); pool.give(b)
    a2, b2 = pool.take(fwd - 1000, dev), pool.take(bwd, dev)
    assert a2.data_ptr() == a.data_ptr() and b2.data_ptr() == b.data_ptr()       # re-used, each by its own kind
    pool.give(a2); pool.give(b2)
    big = pool.take(int(fwd * 1.1), dev)      # a batch with more nodes: new buffer, the superseded forward arena is dropped
    assert big.numel() >= int(fwd * 1.1) and big.data_ptr() != a.data_ptr()
    sizes = sorted(t.numel() for t in pool._list(dev))
    assert sizes == [b.numel()], sizes
    pool.give(big)
    again = pool.take(fwd, dev)               # the smaller batch now fits the larger arena (within the slack): no allocation
    assert again.data_ptr() == big.data_ptr()
    for _ in range(10):                       # never more than MAX_FREE buffers held
        pool.give(torch.empty(1 << 22))
    assert len(pool._list(dev)) <= _ArenaPool.MAX_FREE
    pool.clear()
    assert not pool.free


def test_prefetch_reference_yields_in_order_on_plain_tensors():
    """get_amd.batch.prefetch_reference on CPU tensors (no HIP library: every item takes kargs_from_reference_tensors' plain form):
    one kargs dict per hand-over, in order, equal to the direct call -- the generator's look-ahead must not reorder or drop."""
    import numpy as np
    import torch
    from get_amd.batch import kargs_from_reference_tensors, prefetch_reference
    rng = np.random.default_rng(3)
    items = []
    for i in range(3):
        b, n, r, l = 2 + i, 4, 6, 5
        counts = torch.tensor(rng.integers(0, n + 1, size=b))
        items.append((torch.ones(b), torch.tensor(rng.integers(0, 9, size=(b, n, r))), torch.tensor(rng.random((b, n, r, r))),
                      torch.tensor(rng.random((b, l, l))), counts, torch.zeros((b, n), dtype=torch.int64), None))
    got = list(prefetch_reference(items, n_max=4))
    assert len(got) == 3
    for item, k in zip(items, got):
        ref = kargs_from_reference_tensors(*item, n_max=4)
        assert set(k) == set(ref)
        assert torch.equal(k["doc_content_without_padding_evidences"], ref["doc_content_without_padding_evidences"])
        assert torch.equal(k["docs_adj"], ref["docs_adj"])
        assert k["doc_content_without_padding_evidences"].shape[0] == int(item[4].sum())


def test_bench_block_statistics_tolerate_one_hiccup_block():
    """bench.summarize_blocks: the value is the median block.  Two indicators: `unstable` = the blocks spread more than
    UNSTABLE_SPREAD (round 4's rule: "look at the blocks"); `unstable_without_outliers` = the median itself is not to be trusted --
    several deviating blocks, or a single one among fewer than 8.  ONE hiccup among >= 8 blocks is counted (`outlier_blocks`, both
    spreads printed) and sets only the first."""
    import bench
    one = bench.summarize_blocks({"blocks_s": [0.063] * 7 + [0.095], "settle_s": [0.063, 0.0631]}, 10, [9600] * 8)
    assert one["timed"]["unstable"] and "unstable_without_outliers" not in one["timed"] and one["timed"]["outlier_blocks"] == 1
    assert one["timed"]["spread_rel"] > 0.3 and one["timed"]["spread_rel_without_outliers"] == 0.0
    assert abs(one["value"] - 9600 / 0.063) < 1e-6
    few = bench.summarize_blocks({"blocks_s": [0.063] * 4 + [0.095], "settle_s": []}, 10, [9600] * 5)
    assert few["timed"]["unstable"] and few["timed"]["unstable_without_outliers"]          # 1 of 5 blocks = 20 % of the sample
    many = bench.summarize_blocks({"blocks_s": [0.063] * 5 + [0.095] * 3, "settle_s": []}, 10, [9600] * 8)
    assert many["timed"]["unstable"] and many["timed"]["unstable_without_outliers"] and many["timed"]["outlier_blocks"] == 3
    calm = bench.summarize_blocks({"blocks_s": [0.063, 0.0632, 0.0629, 0.0631, 0.063], "settle_s": []}, 10, [9600] * 5)
    assert "unstable" not in calm["timed"] and "unstable_without_outliers" not in calm["timed"]
    assert calm["timed"]["outlier_blocks"] == 0 and "spread_rel_without_outliers" not in calm["timed"]


def test_unit_seed_backward_is_plain_backward_for_any_loss():
    """get_amd.fused.backward(loss) = torch.autograd.backward with a cached constant 1 as the root gradient: for a loss that does not
    know the trick (plain torch cross-entropy, here on the CPU) the gradients are those of loss.backward()."""
    import torch
    from get_amd import fused
    g = torch.Generator().manual_seed(5)
    x0 = torch.randn(16, 3, generator=g)
    y = torch.randint(0, 3, (16,), generator=g)
    grads = []
    for unit in (False, True):
        x = x0.clone().requires_grad_(True)
        loss = torch.nn.functional.cross_entropy(x, y)
        fused.backward(loss) if unit else loss.backward()
        grads.append(x.grad.clone())
    assert torch.equal(grads[0], grads[1])
    one = fused.unit_seed(torch.device("cpu"))
    assert one.shape == () and float(one) == 1.0 and fused.unit_seed(torch.device("cpu")) is one


def test_stateless_dropout_mask_is_unbiased_and_uncorrelated():
    """Mask hygiene of the kernels' stateless dropout (gemm.hip.h drop_hash: murmur3-finalised idx * golden-ratio + seed;
    ops.dropout_mask_reference is its host replica, wrapper.py:185-190 is the op it implements): over 1.2e7 elements the keep
    rate, the per-column and per-row keep rates, the lag-1 correlations along rows and along columns, and the correlation of
    two seeds' masks all sit within 4 sigma of an i.i.d. Bernoulli(1 - p) field."""
    import numpy as np
    from get_amd.ops import dropout_mask_reference
    rows, cols = 40000, 300
    n = rows * cols
    for p, seed in ((0.2, 12345), (0.5, 7)):
        k = dropout_mask_reference(seed, rows, cols, p).astype(np.float64)
        q = 1.0 - p
        sd = np.sqrt(p * q)
        assert abs(k.mean() - q) <= 4 * sd / np.sqrt(n)
        assert np.abs(k.mean(0) - q).max() <= 4.9 * sd / np.sqrt(rows)          # 300 columns: 4.9 sigma ~ 1e-4 two-sided family-wise
        assert np.abs(k.mean(1) - q).max() <= 5.5 * sd / np.sqrt(cols)          # 40 000 rows
        c = k - q
        for a, b in ((c[:, 1:], c[:, :-1]), (c[1:], c[:-1]), (c[2:], c[:-2]), (c[:, 4:], c[:, :-4])):
            r = float((a * b).mean()) / (p * q)
            assert abs(r) <= 4.0 / np.sqrt(a.size), r
        k2 = dropout_mask_reference(seed + 1, rows, cols, p).astype(np.float64) - q
        assert abs(float((c * k2).mean()) / (p * q)) <= 4.0 / np.sqrt(n)
        k3 = dropout_mask_reference(seed ^ 0x5bd1e995, rows, cols, p).astype(np.float64) - q
        assert abs(float((c * k3).mean()) / (p * q)) <= 4.0 / np.sqrt(n)
