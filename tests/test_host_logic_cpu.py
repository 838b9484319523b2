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
