"""Host-side logic that feeds launch sizes and reported numbers (no device work): the node counts behind the
node-compact layout, and bench.py's FLOP accounting against SURVEY.md 8(d)."""
import numpy as np

from get_amd.batch import NativeBatch
from get_amd.synth import SynthConfig, make_raw_batch
from oracle import get_oracle as O


def test_native_batch_counts_nodes_like_convert_text():
    """m_real sizes every launch of the compact layout: it must equal the sum of convert_text's `length_`
    (interactions.py:351), also for ragged evidence counts, short texts and repeated tokens."""
    cfg = SynthConfig(batch=5, n_evd=0, vocab=60, evd_counts=[1, 7, 30, 2, 11])
    raw = make_raw_batch(cfg, 3)
    raw["evd_len"][0] = 1
    raw["evd_tokens"][1, :] = 5                       # one token repeated: a single node
    nb = NativeBatch(raw["claim_tokens"], raw["claim_len"], raw["evd_tokens"], raw["evd_len"], raw["evd_counts"],
                     raw["doc_sources"], raw["query_sources"], raw["labels"], window=cfg.window,
                     n_max=cfg.fixed_num_evidences, device="cpu")
    want = [O.convert_text([int(t) for t in row], cfg.len_right, int(n), cfg.window)[2]
            for row, n in zip(raw["evd_tokens"], raw["evd_len"])]
    assert nb.evd_nodes_host.tolist() == want and nb.m_real == sum(want)
    assert want[0] == 1 and want[1] == 1
    assert nb.b1 == 51 and nb.compact in (True, False)


def test_bench_flop_accounting_matches_survey():
    import bench
    cfg = SynthConfig()
    f = bench.flops_per_pair(cfg, nnz_per_graph=427.0, real_nodes=64.6)
    assert abs(f["fwd"] / 1e6 - 271.0) < 2.0            # ~271 MFLOP forward, minimal formulation
    assert abs(f["fwd_bwd"] / 1e9 - 0.80) < 0.01        # ~0.80 GFLOP forward + backward
    assert 0.5 < f["executed"] / 1e9 < 0.6              # node-compact layout at 64.6 real nodes of 100
    padded = bench.flops_per_pair(cfg, 427.0, real_nodes=100.0)
    assert abs(padded["executed"] - padded["fwd_bwd"]) < 1e-3 * padded["fwd_bwd"]     # no padding: nothing skipped (the scorer has no backward)
