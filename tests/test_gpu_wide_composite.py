"""Round-5 additions of the composite path (ABI 9): wide hidden layers (BASELINE configs[4], h = 768) and the bf16 storage
pipeline go through gh_get_forward / gh_get_backward; gh_weights_refresh makes the fp32 transposes and the bf16 twins in one
launch; clamped out-of-range ids / labels are counted (gh_clamp_events)."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _cfg4(batch, n_evd=30):
    from get_amd.synth import SynthConfig
    return SynthConfig(batch=batch, n_evd=n_evd, emb_dim=768, hidden=768, word_heads=8, evd_heads=2, window=5, gsl_rate=0.8,
                       vocab=900, n_article_src=40, n_claim_src=10)


def _run(wl, fused_on, mode, monkeypatch):
    from get_amd import _lib, fused, ops
    monkeypatch.setattr(fused, "ENABLED", fused_on)
    model = wl["model"].train(False)
    model.__dict__.pop("_gh_binding", None)
    model.zero_grad(set_to_none=True)
    _lib.set_gemm_mode(mode)
    ops.bump_weight_epoch()
    try:
        _lib.gemm_path_counters(reset=True)
        phi, (ww, ew) = model(wl["query"], wl["document"], **dict(wl["kargs"], output_ranking=True))
        torch.nn.functional.cross_entropy(phi, wl["labels"]).backward()
        torch.cuda.synchronize()
        assert _lib.gemm_path_counters()["generic_large"] == 0
    finally:
        _lib.set_gemm_mode("fp32")
        ops.bump_weight_epoch()
    assert (model.__dict__.get("_gh_binding") is not None) == fused_on, "the wrong path ran"
    return dict(phi=phi.detach().clone(), ww=ww.detach().clone(), ew=ew.detach().clone(),
                score=model.ggnn_with_gsl.last_score.clone(), keep=model.ggnn_with_gsl.last_keep.clone(),
                grads={k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None})


@pytest.mark.parametrize("compact", [True, False])
@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_h768_composite_equals_the_module_by_module_path(mode, compact, monkeypatch):
    """configs[4] (h = 768, 8 word heads, window 5, rate 0.8) at a batch that takes the big-tile / bf16-storage kernels
    (>= 8192 real node rows): the composite entry points against the per-module path, same kernels underneath.  fp32: 2e-6 /
    2e-5 as at h = 300.  bf16 storage: both paths round the same fp32 values to bf16, but their fp32 sums differ in order by
    an ulp here and there, which a bf16 rounding turns into 4e-3 of that value -- stated bounds logits 2e-3, gradients 2e-2 of
    their largest entry, keep-sets equal except for score ties within 1e-3."""
    from bench import build_workload
    wl = build_workload(seed=20240305, device=DEV, cfg=_cfg4(6), compact=compact)
    assert (wl["m_real"] if compact else wl["b1"] * 100) >= 8192
    a = _run(wl, False, mode, monkeypatch)
    b = _run(wl, True, mode, monkeypatch)
    if mode == "fp32":
        assert torch.equal(a["keep"], b["keep"])
        tol_v, tol_g = 2e-6, 2e-5
    else:
        flips = int((a["keep"] != b["keep"]).any(1).sum())
        assert flips <= 0.05 * a["keep"].shape[0], flips
        tol_v, tol_g = 2e-3, 2e-2
    for k in ("phi", "ww", "ew", "score"):
        assert float((a[k] - b[k]).abs().max()) <= tol_v * max(1.0, float(a[k].abs().max())), k
    assert set(a["grads"]) == set(b["grads"]) and len(a["grads"]) >= 40
    for k, g in a["grads"].items():
        scale = max(float(g.abs().max()), 1e-8)
        assert float((g - b["grads"][k]).abs().max()) <= tol_g * scale, k


def test_bf16_composite_training_steps_with_the_flat_trainer(monkeypatch):
    """Three optimiser steps (dropout off, so that the trajectories are comparable) in bf16 storage mode through the composite
    path with a FlatTrainer, beside the same three steps in fp32: the bf16 twins live in persistent buffers that
    gh_weights_refresh rewrites after every step (same launch as the fp32 transposes) -- the cached descriptor stays valid
    (pointer-stable), the twins track the updated parameters exactly, and the loss follows the fp32 run's."""
    from bench import build_workload
    from get_amd import _lib, fused, ops
    from get_amd.dist import FlatTrainer
    traj = {}
    for mode in ("fp32", "bf16"):
        wl = build_workload(seed=20240306, device=DEV, cfg=_cfg4(6), compact=True)
        model = wl["model"].train(False)
        trainer = FlatTrainer(model, lr=1e-4, weight_decay=1e-3)
        _lib.set_gemm_mode(mode)
        ops.bump_weight_epoch()
        try:
            losses, structs = [], []
            for _ in range(3):
                trainer.zero_grad()
                q, d, k = wl["batches"][0].inputs()
                loss = ops.cross_entropy(model(q, d, **k), wl["labels"])
                loss.backward()
                trainer.step()
                losses.append(float(loss.detach()))
                structs.append(id(fused._binding(model).struct[True]))
            torch.cuda.synchronize()
            traj[mode] = losses
            assert all(np.isfinite(losses)) and losses[2] < losses[0], (mode, losses)
            assert structs[1] == structs[2], "the model descriptor was rebuilt although no pointer moved"
            if mode == "bf16":
                w = model.ggnn_with_gsl.feat_prop2.linearz0.linear.weight
                w16, wt16 = ops.bf16_twins(w)
                assert torch.equal(w16, w.detach().to(torch.bfloat16))
                assert torch.equal(wt16, w.detach().t().contiguous().to(torch.bfloat16))
        finally:
            _lib.set_gemm_mode("fp32")
            ops.bump_weight_epoch()
    assert max(abs(a - b) for a, b in zip(traj["fp32"], traj["bf16"])) <= 5e-3, traj


def test_weights_refresh_makes_transposes_and_bf16_twins_in_one_launch():
    from get_amd import _lib
    g = torch.Generator().manual_seed(5)
    mats = [torch.randn(r, c, generator=g).to(DEV) for r, c in ((768, 768), (300, 3556), (5, 300), (33, 65))]
    mats[0][0, 0] = float("nan")
    mats[0][0, 1] = float("inf")
    n = len(mats)
    t32 = [torch.empty(m.shape[1], m.shape[0], device=DEV) for m in mats]
    w16 = [torch.empty(m.shape, device=DEV, dtype=torch.bfloat16) for m in mats]
    t16 = [torch.empty(m.shape[1], m.shape[0], device=DEV, dtype=torch.bfloat16) for m in mats]
    t16[2] = None                                                         # any entry may be NULL
    arr = lambda ts: ctypes.cast((ctypes.c_void_p * n)(*[(t.data_ptr() if t is not None else None) for t in ts]), ctypes.c_void_p)
    ints = lambda v: ctypes.cast((ctypes.c_int * n)(*v), ctypes.c_void_p)
    _lib.call("gh_weights_refresh", n, arr(mats), arr(t32), arr(w16), arr(t16), ints([m.shape[0] for m in mats]),
              ints([m.shape[1] for m in mats]), _lib.stream())
    torch.cuda.synchronize()
    for i, m in enumerate(mats):
        assert torch.equal(t32[i].nan_to_num(7.0), m.t().contiguous().nan_to_num(7.0))
        assert torch.equal(w16[i].float().nan_to_num(7.0), m.to(torch.bfloat16).float().nan_to_num(7.0))      # round-to-nearest-even
        if t16[i] is not None:
            assert torch.equal(t16[i].float().nan_to_num(7.0), m.t().contiguous().to(torch.bfloat16).float().nan_to_num(7.0))


def test_clamped_labels_and_source_ids_are_counted():
    """the reference raises for a label outside [0, C) (nn.CrossEntropyLoss) and for a source id outside its table
    (nn.Embedding); the kernels clamp for memory safety and COUNT the event (gh_clamp_events) so that corrupt data is visible."""
    from get_amd import _lib, ops
    from get_amd.synth import SynthConfig, make_embeddings, make_raw_batch, make_state_dict  # noqa: F401
    from oracle import get_oracle as O
    from oracle.assemble import assemble_inputs, reference_kargs
    from tests.test_gpu_model import build_model, to_dev

    def events(reset=False):
        out = (ctypes.c_int64 * 4)()
        _lib.call("gh_clamp_events", ctypes.cast(out, ctypes.c_void_p), 1 if reset else 0)
        return list(out)

    events(reset=True)
    phi = torch.randn(6, 2, device=DEV)
    ok = torch.tensor([0, 1, 1, 0, 1, 0], device=DEV)
    ops.cross_entropy(phi, ok)
    assert events() == [0, 0, 0, 0]
    bad = torch.tensor([0, -100, 1, 2, 1, 7], device=DEV)
    loss = ops.cross_entropy(phi, bad)
    assert torch.isfinite(loss)
    assert events(reset=True) == [3, 0, 0, 0] and events() == [0, 0, 0, 0]
    # source ids beyond their tables, through the whole model (composite path): article ids hit evd_assemble, claim ids left_assemble
    cfg = SynthConfig(batch=3, emb_dim=32, hidden=32, vocab=200, n_article_src=20, n_claim_src=10, src_dim=16,
                      use_claim_source=True, word_heads=3, evd_heads=1, evd_counts=[5, 1, 30])
    model = build_model(cfg, 701)
    raw = make_raw_batch(cfg, 701)
    inp = assemble_inputs(raw, cfg, O.convert_text)
    kargs = to_dev(reference_kargs(inp, torch))
    q, d = torch.from_numpy(inp["query"]).to(DEV), torch.from_numpy(inp["document"]).to(DEV)
    phi0 = model(q, d, **kargs)
    assert events() == [0, 0, 0, 0]
    kb = dict(kargs)
    src = kargs["doc_sources"].clone()
    src.view(-1)[0] = 20            # == table rows: one past the end
    src.view(-1)[1] = 10 ** 6
    kb["doc_sources"] = src
    qs = kargs["query_sources"].clone()
    qs.view(-1)[2] = 10
    kb["query_sources"] = qs
    phi1 = model(q, d, **kb)
    phi1.sum().backward()           # the backward clamps the same way (no out-of-bounds write into the table gradient)
    torch.cuda.synchronize()
    assert torch.isfinite(phi1).all() and torch.isfinite(model.article_source_embs.weight.grad).all()
    assert events(reset=True) == [0, 1, 2, 0]
    assert phi0.shape == phi1.shape


def test_grad_mode_observables_release_the_arena_when_detached():
    """in grad mode the attention weights carry the fused forward's grad_fn, whose context owns the multi-GB
    activation arena until a backward runs.  A caller that keeps observables past the step without running a backward
    (error analysis in grad mode) must detach() them: the detached copies pin nothing, while an un-detached weight tensor
    keeps the arena alive (documented in INTEGRATION.md).  After a backward the arena goes back to the model's pool, which
    a second model-less reference does not leak."""
    import gc
    from bench import build_workload
    wl = build_workload(batch=16, n_evd=30, seed=92, device=DEV)
    model, nb = wl["model"].train(False), wl["batches"][0]
    q, d, k = nb.inputs()
    torch.cuda.synchronize()
    gc.collect()
    torch.cuda.empty_cache()
    base = torch.cuda.memory_allocated()
    phi, (ww, ew) = model(q, d, **dict(k, output_ranking=True))
    torch.cuda.synchronize()
    held_graph = torch.cuda.memory_allocated() - base
    assert held_graph > 256 << 20, "the grad-mode forward should hold its activation arena"
    kept = (phi.detach().clone(), ww.detach().clone(), ew.detach().clone())
    del phi, ww, ew
    gc.collect()
    torch.cuda.synchronize()
    held = torch.cuda.memory_allocated() - base
    assert held < 64 << 20, f"detached observables still pin {held / 2**20:.0f} MiB"
    # with a backward the arena returns to the model's pool (persistent across steps) and is reused by the next step
    phi, (ww, ew) = model(q, d, **dict(k, output_ranking=True))
    torch.nn.functional.cross_entropy(phi, nb.labels).backward()
    del phi, ww, ew
    gc.collect()
    torch.cuda.synchronize()
    pooled = torch.cuda.memory_allocated() - base
    phi2 = model(q, d, **k)
    torch.nn.functional.cross_entropy(phi2, nb.labels).backward()
    del phi2
    gc.collect()
    torch.cuda.synchronize()
    assert torch.cuda.memory_allocated() - base <= pooled + (8 << 20), "the second step allocated new arenas instead of reusing the pool"
    assert len(kept) == 3


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_h1024_widest_eligible_hidden_size_composite_vs_per_module(mode, monkeypatch):
    """The composite path's eligibility reaches h <= 1024 (ABI 9); the suite otherwise covers h = 300 and h = 768 only.
    h = 1024 (four 256-wide column blocks: four scorer / head-score partials per row), a batch of >= 8192 real node rows: composite
    against the per-module path, same bounds as at h = 768."""
    from bench import build_workload
    from get_amd.synth import SynthConfig
    cfg = SynthConfig(batch=6, n_evd=30, emb_dim=1024, hidden=1024, word_heads=8, evd_heads=2, window=5, gsl_rate=0.8,
                      vocab=900, n_article_src=40, n_claim_src=10)
    wl = build_workload(seed=20240306, device=DEV, cfg=cfg, compact=True)
    assert wl["m_real"] >= 8192
    a = _run(wl, False, mode, monkeypatch)
    b = _run(wl, True, mode, monkeypatch)
    if mode == "fp32":
        assert torch.equal(a["keep"], b["keep"])
        tol_v, tol_g = 2e-6, 2e-5
    else:
        assert int((a["keep"] != b["keep"]).any(1).sum()) <= 0.05 * a["keep"].shape[0]
        tol_v, tol_g = 2e-3, 2e-2
    for k in ("phi", "ww", "ew", "score"):
        assert float((a[k] - b[k]).abs().max()) <= tol_v * max(1.0, float(a[k].abs().max())), k
    for k, g in a["grads"].items():
        assert float((g - b["grads"][k]).abs().max()) <= tol_g * max(float(g.abs().max()), 1e-8), k
