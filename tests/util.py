"""Shared helpers for the parity tests (fixtures, gradient summaries, tolerances)."""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def load(name):
    z = np.load(os.path.join(GOLDEN, name))
    meta = json.loads(bytes(z["meta"]).decode())
    return z, meta


def summary(g):
    """Same summary oracle/make_golden.py stores for large gradients."""
    g = np.asarray(g, dtype=np.float64)
    flat = g.reshape(g.shape[0], -1) if g.ndim > 1 else g.reshape(1, -1)
    return dict(head=flat[:4, :64], sum=g.sum(), abssum=np.abs(g).sum(), sqsum=(g * g).sum())


def check_grad(z, key, got, rtol=1e-3, atol=1e-5, what=""):
    """Compare ``got`` with the fixture entry ``key`` (full tensor or summary form)."""
    got = np.asarray(got, dtype=np.float64)
    if key in z.files:
        exp = z[key].astype(np.float64)
        scale = max(np.abs(exp).max(), 1e-6)
        err = np.abs(got.reshape(exp.shape) - exp).max()
        assert err <= atol + rtol * scale, f"{what}{key}: max err {err:.3e} vs scale {scale:.3e}"
        return
    s = summary(got)
    exp_head = z[key + "::head"]
    scale = max(np.sqrt(float(z[key + "::sqsum"]) / max(got.size, 1)), 1e-6)   # rms of the reference gradient
    err = np.abs(s["head"] - exp_head).max()
    assert err <= atol + rtol * max(scale, np.abs(exp_head).max()), f"{what}{key} head: {err:.3e} (rms {scale:.3e})"
    for k in ("abssum", "sqsum"):
        e = float(z[f"{key}::{k}"])
        assert abs(s[k] - e) <= 2e-3 * abs(e) + atol, f"{what}{key} {k}: {s[k]} vs {e}"
    e = float(z[key + "::sum"])
    assert abs(s["sum"] - e) <= 1e-3 * float(z[key + "::abssum"]) + atol, f"{what}{key} sum: {s['sum']} vs {e}"


def dense_from_coo(z, idx, fixed_length):
    a = np.zeros((fixed_length, fixed_length), np.float64)
    a[z[f"c{idx}_rows"].astype(int), z[f"c{idx}_cols"].astype(int)] = z[f"c{idx}_vals"]
    return a
