"""TEST INFRASTRUCTURE ONLY -- seeded input generators shared by the golden-vector
script (which feeds them to the reference) and the tests (which feed them to the
oracle and to the HIP path), so every side sees byte-identical inputs."""
import numpy as np

from get_amd.synth import make_tokens

G2_CASES = [(3, 30, 48, 64, 3), (2, 100, 300, 300, 3), (3, 100, 300, 1, 3), (2, 100, 64, 64, 5)]
G4_CASES = [(3, 30, 48, 64, 3, 0.6), (2, 100, 300, 300, 3, 0.6), (2, 100, 64, 64, 5, 0.8)]
G5_CASES = [(4, 30, 40, 48, 32, 3, "bool"), (3, 100, 300, 300, 300, 5, "bool"), (4, 30, 300, 1628, 300, 2, "float"),
            (5, 17, 24, 56, 40, 1, "float")]
G6_CASES = [(3, 30, 64, 32, 4), (2, 100, 300, 300, 1)]


def cell_params(rng, din, dout, prefix=""):
    p = {}
    p[prefix + "proj.linear.weight"] = (rng.standard_normal((dout, din)) / np.sqrt(din)).astype(np.float32)
    for g in ("z0", "z1", "r0", "r1", "h0", "h1"):
        sc = 1.0 if dout == 1 else 1.0 / np.sqrt(dout)
        p[prefix + f"linear{g}.linear.weight"] = (rng.standard_normal((dout, dout)) * sc).astype(np.float32)
        p[prefix + f"linear{g}.linear.bias"] = rng.uniform(-0.1, 0.1, size=(dout,)).astype(np.float32)
    return p


def graphs(rng, n, r, window, convert_text_fn, vocab=300):
    toks, lens = make_tokens(rng, n, r, vocab, max(1, r // 3), r)
    ids = np.zeros((n, r), np.int64)
    adj = np.zeros((n, r, r), np.float64)
    for i in range(n):
        w, a, k = convert_text_fn([int(t) for t in toks[i]], r, int(lens[i]), window)
        ids[i], adj[i] = np.asarray(w), np.asarray(a)
    return toks, lens, ids, adj


def g2_inputs(ci, convert_text_fn):
    n, r, din, dout, window = G2_CASES[ci]
    rng = np.random.default_rng(200 + ci)
    toks, lens, ids, adj = graphs(rng, n, r, window, convert_text_fn)
    x = rng.standard_normal((n, r, din)).astype(np.float32)
    gw = rng.standard_normal((n, r, dout)).astype(np.float32)
    p = cell_params(rng, din, dout)
    return dict(toks=toks, lens=lens, ids=ids, adj=adj, x=x, gw=gw, p=p, window=window)


def g4_inputs(ci, convert_text_fn):
    n, r, d, h, window, rate = G4_CASES[ci]
    rng = np.random.default_rng(400 + ci)
    toks, lens, ids, adj = graphs(rng, n, r, window, convert_text_fn)
    x = rng.standard_normal((n, r, d)).astype(np.float32) * 0.4
    gw = rng.standard_normal((n, r, h)).astype(np.float32)
    p = {}
    p.update(cell_params(rng, d, h, "feat_prop1."))
    p.update(cell_params(rng, h, 1, "word_scorer1."))
    p.update(cell_params(rng, h, h, "feat_prop2."))
    return dict(toks=toks, lens=lens, ids=ids, adj=adj, x=x, gw=gw, p=p, rate=rate, window=window)


def g5_inputs(ci):
    b, l, xl, dr, ha, heads, mkind = G5_CASES[ci]
    rng = np.random.default_rng(500 + ci)
    left = rng.standard_normal((b, xl)).astype(np.float32)
    right = rng.standard_normal((b, l, dr)).astype(np.float32)
    valid = rng.integers(1, l + 1, size=b)
    valid[0] = l
    mask = (np.arange(l)[None, :] < valid[:, None])
    w1 = (rng.standard_normal((ha, xl + dr)) / np.sqrt(xl + dr)).astype(np.float32)
    w2 = (rng.standard_normal((heads, ha)) / np.sqrt(ha)).astype(np.float32)
    g_att = rng.standard_normal((b, dr, heads)).astype(np.float32)
    g_w = rng.standard_normal((b, l, heads)).astype(np.float32)
    return dict(left=left, right=right, valid=valid, mask=mask, w1=w1, w2=w2, g_att=g_att, g_w=g_w, mkind=mkind)


def g6_inputs(ci):
    b, l, d, ha, heads = G6_CASES[ci]
    rng = np.random.default_rng(600 + ci)
    tsr = rng.standard_normal((b, l, d)).astype(np.float32)
    valid = rng.integers(1, l + 1, size=b)
    mask = (np.arange(l)[None, :] < valid[:, None]).astype(np.float32)
    w1 = (rng.standard_normal((ha, d)) / np.sqrt(d)).astype(np.float32)
    w2 = (rng.standard_normal((heads, ha)) / np.sqrt(ha)).astype(np.float32)
    # upstream gradients for the backward fixture (drawn AFTER everything else, so the forward inputs keep their values)
    g_att = rng.standard_normal((b, heads, d)).astype(np.float32)      # attended is (B, C, D) in self_attention.py:98
    g_w = rng.standard_normal((b, l, heads)).astype(np.float32)
    return dict(tsr=tsr, valid=valid, mask=mask, w1=w1, w2=w2, g_att=g_att, g_w=g_w)


def g3_scores(ci, r, ties, b=3):
    rng = np.random.default_rng(300 + ci)
    score = rng.standard_normal((b, r, 1)).astype(np.float32)
    if ties:
        score = np.round(score * 2) / 2
        score[0, r // 2:] = score[0, r // 2]
    return score
