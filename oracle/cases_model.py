"""TEST INFRASTRUCTURE ONLY -- the full-model golden cases (config, seed)."""
from get_amd.synth import SynthConfig

MODEL_CASES = {
    "small": (SynthConfig(batch=4, emb_dim=32, hidden=32, vocab=200, n_article_src=20, n_claim_src=10,
                          src_dim=16, evd_counts=[1, 30, 7, 12]), 700),
    "small_claimsrc": (SynthConfig(batch=3, emb_dim=32, hidden=32, vocab=200, n_article_src=20, n_claim_src=10,
                                   src_dim=16, use_claim_source=True, word_heads=3, evd_heads=1,
                                   evd_counts=[5, 1, 30]), 701),
    "full": (SynthConfig(batch=4, vocab=2000, n_article_src=50, n_claim_src=10, evd_counts=[3, 30, 1, 9]), 702),
}
