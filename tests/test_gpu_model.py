"""GPU parity of the full drop-in model against the golden vectors
captured from the reference (G7/G8) and against the CPU oracle.  Tolerances: logits 1e-4
(north_star), attention weights / scorer outputs 1e-5, gradients 1e-3 relative."""
import numpy as np
import pytest
import torch

from get_amd.synth import make_embeddings, make_raw_batch, make_state_dict
from oracle import get_oracle as O
from oracle.assemble import assemble_inputs, reference_kargs
from oracle.cases_model import MODEL_CASES
from tests.util import check_grad, load

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def build_model(cfg, seed):
    from get_amd import modules
    emb, art, clm = make_embeddings(cfg, seed)
    model = modules.Graph_basedSemantiStructure(cfg.model_params(emb, art, clm))
    full = model.state_dict()
    full.update({k: torch.from_numpy(v) for k, v in make_state_dict(cfg, seed).items()})
    model.load_state_dict(full, strict=True)
    return model.to(DEV).train(False)


def to_dev(kargs):
    out = {}
    for k, v in kargs.items():
        if torch.is_tensor(v):
            out[k] = v.to(DEV)
        elif isinstance(v, tuple):
            out[k] = tuple(t.to(DEV) if torch.is_tensor(t) else t for t in v)
        else:
            out[k] = v
    return out


def run_case(name, native_graphs=False, int32_inputs=False):
    cfg, seed = MODEL_CASES[name]
    model = build_model(cfg, seed)
    raw = make_raw_batch(cfg, seed)
    inp = assemble_inputs(raw, cfg, O.convert_text)
    kargs = to_dev(reference_kargs(inp, torch, output_ranking=True))
    query = torch.from_numpy(inp["query"]).to(DEV)
    document = torch.from_numpy(inp["document"]).to(DEV)
    if native_graphs:
        from get_amd import ops
        qa, q_ids, q_n = ops.graph_build(torch.from_numpy(raw["claim_tokens"]).to(DEV),
                                         torch.from_numpy(raw["claim_len"]).to(DEV), cfg.window)
        da, d_ids, d_n = ops.graph_build(torch.from_numpy(raw["evd_tokens"]).to(DEV),
                                         torch.from_numpy(raw["evd_len"]).to(DEV), cfg.window)
        assert np.array_equal(q_ids.cpu().numpy(), inp["query"]) and np.array_equal(d_ids.cpu().numpy(), inp["doc_ids"])
        assert np.array_equal(q_n.cpu().numpy(), inp["query_lens"])
        if native_graphs == "compact":     # node-compact fast path (ops.RaggedPlan): padding nodes skipped where inert
            da = da.with_plan(ops.RaggedPlan(d_n, d_ids, int(inp["doc_lens"].sum()) if "doc_lens" in inp else int(d_n.sum().item())))
        kargs["query_adj"], kargs["docs_adj"] = qa, da
    if int32_inputs:       # the evaluation path hands int32 ids/lens (char_man_fitter_query_repr1.py:298-316)
        query, document = query.int(), document.int()
        for k in ("query_lens", "doc_content_without_padding_evidences", "doc_sources", "query_sources"):
            kargs[k] = kargs[k].int()
    phi, (ww, ew) = model(query, document, **kargs)
    loss = torch.nn.functional.cross_entropy(phi, torch.from_numpy(inp["labels"]).to(DEV))
    return cfg, model, inp, phi, ww, ew, loss


@pytest.mark.parametrize("name", list(MODEL_CASES))
@pytest.mark.parametrize("native", [False, True, "compact"])
def test_model_vs_golden(name, native, arith_mode):
    z, meta = load(f"g7_model_{name}.npz")
    cfg, model, inp, phi, ww, ew, loss = run_case(name, native_graphs=native)
    assert phi.shape == (cfg.batch, cfg.num_classes)
    assert np.abs(phi.detach().cpu().numpy() - z["phi"]).max() <= 1e-4
    assert np.abs(ww.detach().cpu().numpy() - z["word_w"]).max() <= 1e-5
    assert np.abs(ew.detach().cpu().numpy() - z["evd_w"]).max() <= 1e-5
    assert np.abs(model.ggnn_with_gsl.last_score.cpu().numpy() - z["score"]).max() <= 1e-5
    assert abs(loss.item() - float(z["loss"])) <= 1e-5
    loss.backward()
    none = set(meta["none_grads"])
    n_live = 0
    for k, prm in model.named_parameters():
        if k in none:
            assert prm.grad is None, f"{k} must not receive a gradient"
            continue
        assert prm.grad is not None, k
        check_grad(z, f"g::{k}", prm.grad.cpu().numpy(), what=f"{name} native={native} ")
        n_live += prm.numel()
    assert n_live == meta["n_live"]


@pytest.mark.parametrize("native", [False, "compact"])
def test_claim_side_stream_is_bit_identical_to_single_stream(native, monkeypatch):
    """The forward issues the claim branch on an auxiliary stream (modules.Graph_basedSemantiStructure.forward): logits,
    attention weights and every gradient must be bit-identical to the single-stream schedule, several times in a row
    (a missing join would show up as stale or torn claim vectors)."""
    from get_amd import ops
    runs = {}
    for side in (False, True, True, False):
        monkeypatch.setattr(ops, "CLAIM_SIDE_STREAM", side)
        cfg, model, inp, phi, ww, ew, loss = run_case("small", native_graphs=native)
        loss.backward()
        torch.cuda.synchronize()
        res = [phi.detach().clone(), ww.detach().clone(), ew.detach().clone()] + \
              [p.grad.clone() for _, p in sorted(model.named_parameters()) if p.grad is not None]
        runs.setdefault(side, []).append(res)
    ref = runs[False][0]
    for side in (False, True):
        for res in runs[side]:
            assert len(res) == len(ref)
            for a, b in zip(res, ref):
                assert torch.equal(a, b)


def test_side_stream_weight_gradients_are_bit_identical(monkeypatch):
    """With a FlatTrainer the weight gradients of the evidence-level attention and the head are issued on the auxiliary
    stream and joined when backward ends (ops._side_wgrad): the flat gradient bucket must equal the in-line schedule bit
    for bit, read right after backward() on the caller's stream, several times in a row."""
    from get_amd import ops
    from get_amd.dist import FlatTrainer
    cfg, model, inp, phi, ww, ew, loss = run_case("small", native_graphs="compact")
    trainer = FlatTrainer(model, lr=1e-4, weight_decay=1e-3)
    ops.bump_weight_epoch()
    model.train(False)
    raw = make_raw_batch(cfg, MODEL_CASES["small"][1])
    buckets = {}
    for side in (False, True, True, False, True):
        monkeypatch.setattr(ops, "WGRAD_SIDE_STREAM", side)
        trainer.zero_grad()
        cfg2, model2, inp2, phi2, ww2, ew2, loss2 = None, None, None, None, None, None, None
        query = torch.from_numpy(inp["query"]).to(DEV)
        document = torch.from_numpy(inp["document"]).to(DEV)
        kargs = to_dev(reference_kargs(inp, torch, output_ranking=False))
        out = model(query, document, **kargs)
        torch.nn.functional.cross_entropy(out, torch.from_numpy(inp["labels"]).to(DEV)).backward()
        buckets.setdefault(side, []).append(trainer.flat_g.clone())       # no synchronize: stream order must suffice
    torch.cuda.synchronize()
    ref = buckets[False][0]
    assert float(ref.abs().max()) > 0
    for side in (False, True):
        for b in buckets[side]:
            assert torch.equal(b, ref)


def test_model_eval_path_int32_and_predict():
    z, _ = load("g7_model_small.npz")
    cfg, model, inp, phi, ww, ew, loss = run_case("small", int32_inputs=True)
    assert np.abs(phi.detach().cpu().numpy() - z["phi"]).max() <= 1e-4
    assert np.allclose(ww.detach().sum(1).cpu().numpy(), 1.0, atol=1e-5)
    # evidence slots beyond a claim's count get exactly zero weight
    cnt = inp["evd_counts"]
    e = ew.detach().cpu().numpy()
    for b, c in enumerate(cnt):
        assert np.all(e[b, c:] == 0) and abs(e[b, :c].sum(0) - 1).max() <= 1e-5


def test_model_adam_step_matches_golden(arith_mode):
    """G8: one Adam(lr=1e-4, weight_decay=1e-3) step through the flat bucket the DP wrapper uses."""
    from get_amd.dist import FlatTrainer
    z, meta = load("g7_model_small.npz")
    cfg, seed = MODEL_CASES["small"]
    model = build_model(cfg, seed)
    trainer = FlatTrainer(model, lr=1e-4, weight_decay=1e-3)
    raw = make_raw_batch(cfg, seed)
    inp = assemble_inputs(raw, cfg, O.convert_text)
    kargs = to_dev(reference_kargs(inp, torch))
    before = {k: p.detach().clone() for k, p in model.named_parameters()}
    trainer.zero_grad()
    phi = model(torch.from_numpy(inp["query"]).to(DEV), torch.from_numpy(inp["document"]).to(DEV), **kargs)
    torch.nn.functional.cross_entropy(phi, torch.from_numpy(inp["labels"]).to(DEV)).backward()
    trainer.step()
    none = set(meta["none_grads"])
    for k, prm in model.named_parameters():
        if k in none:
            # grad-None parameters are skipped by the reference's Adam (no weight decay either);
            # their values are module-specific random init, so the check is "unchanged"
            assert k not in trainer.live_names and torch.equal(prm.detach(), before[k]), k
            continue
        exp = z[f"adam::{k}"]
        assert np.abs(prm.detach().cpu().numpy() - exp).max() <= 2e-6, k


def test_bench_shape_forward_backward_properties():
    """BASELINE config 2 at full size (B=32, 30 evidences, R=100, H=300): size-independent properties."""
    from bench import build_workload
    wl = build_workload(batch=32, n_evd=30, seed=20240229, device=DEV)
    model = wl["model"].train(False)
    phi, (ww, ew) = model(wl["query"], wl["document"], **dict(wl["kargs"], output_ranking=True))
    assert phi.shape == (32, 2) and torch.isfinite(phi).all()
    assert ww.shape == (960, 100, 5) and ew.shape == (32, 30, 2)
    assert torch.allclose(ww.sum(1), torch.ones(960, 5, device=DEV), atol=1e-5)
    assert torch.allclose(ew.sum(1), torch.ones(32, 2, device=DEV), atol=1e-5)
    # padded evidence-graph nodes get zero word attention
    pad = wl["kargs"]["doc_content_without_padding_evidences"] < 1
    assert float(ww.detach()[pad].abs().max()) == 0.0
    loss = torch.nn.functional.cross_entropy(phi, wl["labels"])
    loss.backward()
    for k, p in model.named_parameters():
        if p.grad is not None:
            assert torch.isfinite(p.grad).all(), k
    # a 4-claim slice of the same batch agrees with the oracle on the CPU (logits 1e-4)
    sub = wl["oracle_slice"](4)
    phi_o = sub["phi"]
    assert float((phi[:4].detach().cpu() - phi_o).abs().max()) <= 1e-4


def _full_size_properties(cfg, seed, compact, slice_claims, min_rows):
    """Size-independent properties of one forward + backward at a BASELINE shape, plus an oracle slice (logits 1e-4,
    word-attention weights 1e-5) of the first `slice_claims` claims of the same batch."""
    from bench import build_workload
    from get_amd import _lib
    wl = build_workload(seed=seed, device=DEV, cfg=cfg, compact=compact)
    assert wl["compact"] == compact
    model = wl["model"].train(False)
    b, n, r, hw, he = cfg.batch, cfg.fixed_num_evidences, cfg.len_right, cfg.word_heads, cfg.evd_heads
    b1 = int(wl["b1"])
    rows = wl["m_real"] if compact else b1 * r
    assert rows >= min_rows, "the shape must take the big-tile GEMM path"
    _lib.gemm_path_counters(reset=True)
    phi, (ww, ew) = model(wl["query"], wl["document"], **dict(wl["kargs"], output_ranking=True))
    assert phi.shape == (b, 2) and torch.isfinite(phi).all()
    assert ww.shape == (b1, r, hw) and ew.shape == (b, n, he)
    assert torch.allclose(ww.sum(1), torch.ones(b1, hw, device=DEV), atol=1e-5)
    counts = torch.as_tensor(wl["raw"]["evd_counts"])
    assert torch.allclose(ew.sum(1), torch.ones(b, he, device=DEV), atol=1e-5)
    slot_pad = (torch.arange(n)[None, :] >= counts[:, None]).to(DEV)
    assert float(ew.detach()[slot_pad].abs().max() if slot_pad.any() else 0.0) == 0.0      # padded evidence slots
    pad = wl["kargs"]["doc_content_without_padding_evidences"] < 1
    assert float(ww.detach()[pad].abs().max()) == 0.0                                          # padded graph nodes
    torch.nn.functional.cross_entropy(phi, wl["labels"]).backward()
    for k, p in model.named_parameters():
        if p.grad is not None:
            assert torch.isfinite(p.grad).all(), k
    assert _lib.gemm_path_counters()["generic_large"] == 0, "a large GEMM fell off the MFMA fast path"
    sub = wl["oracle_slice"](slice_claims)
    nb1 = int(wl["raw"]["evd_counts"][:slice_claims].sum())
    assert float((phi[:slice_claims].detach().cpu() - sub["phi"]).abs().max()) <= 1e-4
    assert float((ww[:nb1].detach().cpu() - sub["word_w"]).abs().max()) <= 1e-5


@pytest.mark.parametrize("compact", [False, True])
def test_config2_politifact_full_size(compact):
    """BASELINE configs[2] at FULL size: B=64 claims x 10 evidences, L_right=200 (4 bit words per adjacency row, 8-slab
    aggregation), D=H=300 -- properties + a 2-claim oracle slice, both row layouts."""
    from get_amd.synth import SynthConfig
    _full_size_properties(SynthConfig(batch=64, n_evd=10, len_right=200), 20240301, compact, 2, 8192)


@pytest.mark.parametrize("compact", [False, True])
def test_config4_h768_full_width_fp32(compact):
    """BASELINE configs[4] shape in fp32 at a batch that takes the big-tile path (>= 8192 real node rows): h=768 (three
    column blocks per GEMM), 8 word heads, gnn_window=5, gsl_rate=0.8 -- properties + a 2-claim oracle slice."""
    from get_amd.synth import SynthConfig
    cfg = SynthConfig(batch=5, n_evd=30, emb_dim=768, hidden=768, word_heads=8, window=5, gsl_rate=0.8)
    _full_size_properties(cfg, 20240302, compact, 2, 8192)


@pytest.mark.parametrize("compact", [False, True])
def test_politifact_shaped_long_evidence_vs_oracle(compact):
    """BASELINE configs[2] shape (L_right=200, 10 evidences/claim) at reduced width: full model, native
    graphs, HIP vs the CPU oracle (logits 1e-4, attention weights 1e-5, live gradients 1e-3)."""
    from get_amd import ops
    from get_amd.synth import SynthConfig
    cfg = SynthConfig(batch=4, n_evd=10, len_right=200, emb_dim=64, hidden=64, vocab=500, n_article_src=30,
                      n_claim_src=10, src_dim=16, word_heads=3, evd_heads=1, window=5, gsl_rate=0.8)
    seed = 4242
    model = build_model(cfg, seed)
    raw = make_raw_batch(cfg, seed)
    inp = assemble_inputs(raw, cfg, O.convert_text)
    kargs = to_dev(reference_kargs(inp, torch, output_ranking=True))
    da, d_ids, d_n = ops.graph_build(torch.from_numpy(raw["evd_tokens"]).to(DEV), torch.from_numpy(raw["evd_len"]).to(DEV), cfg.window)
    assert da.words == 4 and np.array_equal(d_ids.cpu().numpy(), inp["doc_ids"])
    if compact:
        da = da.with_plan(ops.RaggedPlan(d_n, d_ids, int(d_n.sum().item())))
    kargs["docs_adj"] = da
    phi, (ww, ew) = model(torch.from_numpy(inp["query"]).to(DEV), torch.from_numpy(inp["document"]).to(DEV), **kargs)
    torch.nn.functional.cross_entropy(phi, torch.from_numpy(inp["labels"]).to(DEV)).backward()
    emb, art, clm = make_embeddings(cfg, seed)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    p = {k: T(v).requires_grad_(True) for k, v in make_state_dict(cfg, seed).items()}
    p["embedding.weight"] = T(emb)
    p["article_source_embs.weight"] = T(art).requires_grad_(True)
    phi_o, ww_o, ew_o = O.model_forward(p, cfg.__dict__, T(inp["query"]), T(inp["document"]), T(inp["query_adj"]),
                                        T(inp["doc_ids"]), T(inp["doc_adj"]), T(inp["query_lens"]), inp["evd_counts"],
                                        T(inp["doc_sources"]), T(inp["query_sources"]))
    O.cross_entropy(phi_o, T(inp["labels"])).backward()
    assert float((phi.detach().cpu() - phi_o.detach()).abs().max()) <= 1e-4
    assert float((ww.detach().cpu() - ww_o.detach()).abs().max()) <= 1e-5
    assert float((ew.detach().cpu() - ew_o.detach()).abs().max()) <= 1e-5
    for k, prm in model.named_parameters():
        if k in p and p[k].grad is not None and prm.grad is not None:
            go = p[k].grad
            err = float((prm.grad.cpu() - go).abs().max())
            assert err <= 1e-3 * float(go.abs().max()) + 1e-6, (k, err)


@pytest.mark.parametrize("compact", [False, True])
def test_h768_wide_hidden_vs_oracle(compact):
    """BASELINE configs[4] shape in fp32 (h=768, 8 word heads, gnn_window=5, gsl_rate=0.8) at a reduced batch:
    hidden layers wider than one GEMM column block (cells split into column blocks, the attention's tanh + head
    scores run as a separate row pass).  HIP vs the CPU oracle: logits 1e-4, attention weights 1e-5, gradients 1e-3."""
    from get_amd import ops
    from get_amd.synth import SynthConfig
    cfg = SynthConfig(batch=2, n_evd=4, hidden=768, word_heads=8, evd_heads=2, window=5, gsl_rate=0.8, vocab=900,
                      n_article_src=40, n_claim_src=10)
    seed = 768
    model = build_model(cfg, seed)
    raw = make_raw_batch(cfg, seed)
    inp = assemble_inputs(raw, cfg, O.convert_text)
    kargs = to_dev(reference_kargs(inp, torch, output_ranking=True))
    da, d_ids, d_n = ops.graph_build(torch.from_numpy(raw["evd_tokens"]).to(DEV), torch.from_numpy(raw["evd_len"]).to(DEV), cfg.window)
    if compact:
        da = da.with_plan(ops.RaggedPlan(d_n, d_ids, int(d_n.sum().item())))
    kargs["docs_adj"] = da
    phi, (ww, ew) = model(torch.from_numpy(inp["query"]).to(DEV), torch.from_numpy(inp["document"]).to(DEV), **kargs)
    torch.nn.functional.cross_entropy(phi, torch.from_numpy(inp["labels"]).to(DEV)).backward()
    emb, art, clm = make_embeddings(cfg, seed)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    p = {k: T(v).requires_grad_(True) for k, v in make_state_dict(cfg, seed).items()}
    p["embedding.weight"] = T(emb)
    p["article_source_embs.weight"] = T(art).requires_grad_(True)
    phi_o, ww_o, ew_o = O.model_forward(p, cfg.__dict__, T(inp["query"]), T(inp["document"]), T(inp["query_adj"]),
                                        T(inp["doc_ids"]), T(inp["doc_adj"]), T(inp["query_lens"]), inp["evd_counts"],
                                        T(inp["doc_sources"]), T(inp["query_sources"]))
    O.cross_entropy(phi_o, T(inp["labels"])).backward()
    assert float((phi.detach().cpu() - phi_o.detach()).abs().max()) <= 1e-4
    assert float((ww.detach().cpu() - ww_o.detach()).abs().max()) <= 1e-5
    assert float((ew.detach().cpu() - ew_o.detach()).abs().max()) <= 1e-5
    n_checked = 0
    for k, prm in model.named_parameters():
        if k in p and p[k].grad is not None and prm.grad is not None:
            go = p[k].grad
            err = float((prm.grad.cpu() - go).abs().max())
            assert err <= 1e-3 * float(go.abs().max()) + 1e-6, (k, err)
            n_checked += 1
    assert n_checked >= 40


def test_h768_bf16_gemm_mode_tracks_the_fp32_oracle():
    """BASELINE configs[4] ("h=768 bf16"): with gh_set_gemm_mode(1) the big projections run on bf16 MFMA.  At a batch
    large enough to take that path the logits must track the fp32 oracle to bf16 accuracy, the attention weights
    still sum to one, and switching back restores the fp32 result bit for bit."""
    from get_amd import _lib, ops
    from get_amd.synth import SynthConfig
    cfg = SynthConfig(batch=6, n_evd=30, emb_dim=768, hidden=768, word_heads=8, evd_heads=2, window=5, gsl_rate=0.8, vocab=900,
                      n_article_src=40, n_claim_src=10)
    seed = 769
    model = build_model(cfg, seed)
    raw = make_raw_batch(cfg, seed)
    inp = assemble_inputs(raw, cfg, O.convert_text)
    kargs = to_dev(reference_kargs(inp, torch, output_ranking=True))
    da, d_ids, d_n = ops.graph_build(torch.from_numpy(raw["evd_tokens"]).to(DEV), torch.from_numpy(raw["evd_len"]).to(DEV), cfg.window)
    kargs["docs_adj"] = da.with_plan(ops.RaggedPlan(d_n, d_ids, int(d_n.sum().item())))
    assert int(d_n.sum().item()) >= 8192                     # real-node rows: the big-tile GEMM configuration is in play
    q, d = torch.from_numpy(inp["query"]).to(DEV), torch.from_numpy(inp["document"]).to(DEV)
    with torch.no_grad():
        phi32, (ww32, _) = model(q, d, **kargs)
        keep32, score32 = model.ggnn_with_gsl.last_keep.clone(), model.ggnn_with_gsl.last_score.clone()
        _lib.set_gemm_mode("bf16")
        try:
            _lib.gemm_path_counters(reset=True)
            phi16, (ww16, ew16) = model(q, d, **kargs)
            keep16, score16 = model.ggnn_with_gsl.last_keep.clone(), model.ggnn_with_gsl.last_score.clone()
        finally:
            _lib.set_gemm_mode("fp32")
        phi32b, _ = model(q, d, **kargs)
    assert torch.equal(phi32, phi32b)
    # stated bf16 bounds per quantity (measured on MI355X: logits 9.5e-5 at scale 0.07, word weights 1.0e-3, scorer scores
    # 1.9e-3 at scale 0.45, 12 of 180 graphs / 16 of 9690 real nodes with a different GSL keep decision)
    diff = float((phi16 - phi32).abs().max())
    assert 1e-6 < diff <= 2e-2 * max(1.0, float(phi32.abs().max())), diff
    assert diff <= 2e-3, diff
    assert float((ww16 - ww32).abs().max()) <= 5e-3
    assert float((score16 - score32).abs().max()) <= 1e-2
    R = cfg.len_right
    unpack = lambda k: ((k.cpu().numpy().astype(np.uint64)[:, :, None] >> np.arange(64, dtype=np.uint64)[None, None, :])
                        & np.uint64(1)).astype(bool).reshape(k.shape[0], -1)[:, :R]
    real = d_ids.cpu().numpy() > 0
    mism = (unpack(keep32) != unpack(keep16)) & real
    graphs_off, nodes_off = int(mism.any(1).sum()), int(mism.sum())
    print(f"bf16 vs fp32: logits {diff:.2e}, GSL keep decisions differ in {graphs_off} of {mism.shape[0]} graphs "
          f"({nodes_off} of {int(real.sum())} real nodes)")
    assert nodes_off <= 0.005 * real.sum() and graphs_off <= 0.15 * mism.shape[0]
    assert torch.allclose(ww16.sum(1), torch.ones_like(ww16.sum(1)), atol=1e-5)
    assert torch.allclose(ew16.sum(1), torch.ones_like(ew16.sum(1)), atol=1e-5)
    # training step in both modes: every live gradient of the bf16 storage pipeline stays within 6e-2 of the fp32 one
    # (relative to that gradient's largest entry) -- the stated tolerance of the configs[4] path
    labels = torch.from_numpy(inp["labels"]).to(DEV)
    grads = {}
    for mode in ("fp32", "bf16"):
        model.zero_grad(set_to_none=True)
        _lib.set_gemm_mode(mode)
        try:
            torch.nn.functional.cross_entropy(model(q, d, **{k: v for k, v in kargs.items() if k != "output_ranking"}), labels).backward()
        finally:
            _lib.set_gemm_mode("fp32")
        grads[mode] = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
    assert set(grads["fp32"]) == set(grads["bf16"]) and len(grads["fp32"]) >= 40
    for k, g32 in grads["fp32"].items():
        err = float((grads["bf16"][k] - g32).abs().max())
        assert err <= 6e-2 * float(g32.abs().max()) + 1e-7, (k, err, float(g32.abs().max()))


@pytest.mark.parametrize("train", [False, True])
def test_h768_bf16_storage_vs_the_cpu_oracle(train):
    """(train = True: the mode its bench line is measured in -- the four cells' input dropout on, the oracle replaying the product's
    stateless masks, test_training_mode_model_vs_oracle_replaying_the_dropout_masks; in bf16 storage the first cell's masked operand is
    materialised once by gather_rows and re-used by its weight gradient.)
    BASELINE configs[4], DIRECT oracle comparison: the bf16 storage pipeline at h = 768, 8 word heads,
    window 5, rate 0.8, at a batch that takes the big-tile path, against O.model_forward (wrapper.py:188-206 arithmetic in
    fp32 on the CPU) on the SAME batch -- not against the HIP fp32 path.  Stated bf16 bounds per quantity: logits 2e-3,
    word / evidence attention weights 5e-3, scorer scores 1e-2, every live gradient 6e-2 of its largest entry; GSL keep
    decisions may flip for nodes whose score sits within bf16 noise of the k-th: at most 0.5 % of the real nodes and 15 %
    of the graphs (counts printed)."""
    from get_amd import _lib, ops
    from get_amd.synth import SynthConfig
    cfg = SynthConfig(batch=6, n_evd=30, emb_dim=768, hidden=768, word_heads=8, evd_heads=2, window=5, gsl_rate=0.8, vocab=900,
                      n_article_src=40, n_claim_src=10)
    seed = 769
    model = build_model(cfg, seed).train(train)
    raw = make_raw_batch(cfg, seed)
    inp = assemble_inputs(raw, cfg, O.convert_text)
    kargs = to_dev(reference_kargs(inp, torch, output_ranking=True))
    da, d_ids, d_n = ops.graph_build(torch.from_numpy(raw["evd_tokens"]).to(DEV), torch.from_numpy(raw["evd_len"]).to(DEV), cfg.window)
    plan = ops.RaggedPlan(d_n, d_ids, int(d_n.sum().item()))
    kargs["docs_adj"] = da.with_plan(plan)
    assert int(d_n.sum().item()) >= 8192
    drop_keep = None
    if train:
        torch.manual_seed(seed)
        seeds = torch.randint(0, 2 ** 31 - 1, (4,)).tolist()
        torch.manual_seed(seed)
        p_c, p_g = float(model.ggnn4claim_1.dropout.p), float(model.ggnn_with_gsl.feat_prop1.dropout.p)
        B_, L_ = inp["query"].shape
        B1_, R_ = inp["doc_ids"].shape
        src = plan.src.cpu().numpy()

        def rows_mask(sd, width):
            kc = ops.dropout_mask_reference(sd, B1_ * R_, width, p_g)
            k = np.zeros_like(kc)
            k[src] = kc
            return torch.from_numpy(k.reshape(B1_, R_, width))
        drop_keep = {"claim": (torch.from_numpy(ops.dropout_mask_reference(seeds[0], B_ * L_, cfg.emb_dim, p_c).reshape(B_, L_, cfg.emb_dim)), p_c),
                     "cell1": (rows_mask(seeds[1], cfg.emb_dim), p_g), "scorer": (rows_mask(seeds[2], cfg.hidden), p_g),
                     "cell2": (rows_mask(seeds[3], cfg.hidden), p_g)}
    q, d = torch.from_numpy(inp["query"]).to(DEV), torch.from_numpy(inp["document"]).to(DEV)
    labels = torch.from_numpy(inp["labels"]).to(DEV)
    _lib.set_gemm_mode("bf16")
    try:
        _lib.gemm_path_counters(reset=True)
        phi, (ww, ew) = model(q, d, **kargs)
        assert not train or getattr(model, "_gh_binding", None) is not None, "the masks are replayed for the composite path's seeds"
        score = model.ggnn_with_gsl.last_score.detach().cpu()
        keep = model.ggnn_with_gsl.last_keep.cpu().numpy().astype(np.uint64)
        torch.nn.functional.cross_entropy(phi, labels).backward()
        assert _lib.gemm_path_counters()["generic_large"] == 0
    finally:
        _lib.set_gemm_mode("fp32")
    # the oracle on the same batch (fp32, CPU)
    emb, art, clm = make_embeddings(cfg, seed)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    p = {k: T(v).requires_grad_(True) for k, v in make_state_dict(cfg, seed).items()}
    p["embedding.weight"] = T(emb)
    p["article_source_embs.weight"] = T(art).requires_grad_(True)
    phi_o, ww_o, ew_o, aux = O.model_forward(p, cfg.__dict__, T(inp["query"]), T(inp["document"]), T(inp["query_adj"]),
                                             T(inp["doc_ids"]), T(inp["doc_adj"]), T(inp["query_lens"]), inp["evd_counts"],
                                             T(inp["doc_sources"]), T(inp["query_sources"]), return_aux=True, drop_keep=drop_keep)
    O.cross_entropy(phi_o, T(inp["labels"])).backward()
    d_phi = float((phi.detach().cpu() - phi_o.detach()).abs().max())
    d_ww = float((ww.detach().cpu() - ww_o.detach()).abs().max())
    d_ew = float((ew.detach().cpu() - ew_o.detach()).abs().max())
    d_sc = float((score - aux["score"].detach()).abs().max())
    R = cfg.len_right
    bits = ((keep[:, :, None] >> np.arange(64, dtype=np.uint64)[None, None, :]) & np.uint64(1)).astype(bool).reshape(keep.shape[0], -1)[:, :R]
    real = inp["doc_ids"] > 0
    mism = (bits != aux["keep"].numpy().astype(bool)) & real
    graphs_off, nodes_off = int(mism.any(1).sum()), int(mism.sum())
    worst_g, worst_k = 0.0, None
    n_checked = 0
    for k, prm in model.named_parameters():
        if k in p and p[k].grad is not None and prm.grad is not None:
            go = p[k].grad
            rel = float((prm.grad.cpu() - go).abs().max()) / (float(go.abs().max()) + 1e-12)
            if rel > worst_g:
                worst_g, worst_k = rel, k
            n_checked += 1
    print(f"bf16 storage vs the CPU oracle: logits {d_phi:.2e}, word weights {d_ww:.2e}, evidence weights {d_ew:.2e}, scores "
          f"{d_sc:.2e}, worst gradient {worst_g:.2e} ({worst_k}), keep decisions differ in {graphs_off} of {mism.shape[0]} graphs "
          f"({nodes_off} of {int(real.sum())} real nodes)")
    assert 1e-6 < d_phi <= 2e-3, d_phi
    assert d_ww <= 5e-3 and d_ew <= 5e-3, (d_ww, d_ew)
    assert d_sc <= 1e-2, d_sc
    assert nodes_off <= 0.005 * real.sum() and graphs_off <= 0.15 * mism.shape[0]
    assert n_checked >= 40 and worst_g <= 6e-2, (worst_k, worst_g)


def test_ragged_realistic_batch_properties():
    """Evidence counts drawn U[1,30] (B1 not a multiple of any tile): weights sum to one, padded slots and
    padded nodes get exactly zero attention, gradients finite, logits equal the oracle on a slice."""
    from bench import build_workload
    wl = build_workload(batch=16, n_evd=0, seed=77, device=DEV)
    model = wl["model"].train(False)
    phi, (ww, ew) = model(wl["query"], wl["document"], **dict(wl["kargs"], output_ranking=True))
    counts = wl["raw"]["evd_counts"]
    assert ww.shape[0] == int(counts.sum())
    assert torch.allclose(ww.sum(1), torch.ones_like(ww.sum(1)), atol=1e-5)
    assert torch.allclose(ew.sum(1), torch.ones_like(ew.sum(1)), atol=1e-5)
    e = ew.detach().cpu().numpy()
    for b, c in enumerate(counts):
        assert np.all(e[b, c:] == 0)
    torch.nn.functional.cross_entropy(phi, wl["labels"]).backward()
    assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)
    sub = wl["oracle_slice"](3)
    assert float((phi[:3].detach().cpu() - sub["phi"]).abs().max()) <= 1e-4


def test_batch_shim_matches_fitter_depadding():
    """get_amd.batch: the dense compatibility shim reproduces the fitter's per-claim de-padding
    (char_man_fitter_query_repr1.py:204-223) and the native batch yields the same ids/graphs."""
    from get_amd.batch import NativeBatch, kargs_from_reference_tensors
    cfg, seed = MODEL_CASES["small"]
    raw = make_raw_batch(cfg, seed)
    inp = assemble_inputs(raw, cfg, O.convert_text)
    B, n, R = inp["document"].shape
    adj_padded = np.zeros((B, n, R, R))
    last = 0
    for b, c in enumerate(inp["evd_counts"]):
        adj_padded[b, :c] = inp["doc_adj"][last:last + c]
        last += c
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    kargs = kargs_from_reference_tensors(T(inp["query_lens"]), T(inp["document"]), T(adj_padded), T(inp["query_adj"]),
                                         T(inp["evd_counts"]), T(inp["doc_sources"]), T(inp["query_sources"]), fused=False)
    assert np.array_equal(kargs["doc_content_without_padding_evidences"].cpu().numpy(), inp["doc_ids"])
    assert np.array_equal(kargs["docs_adj"].cpu().numpy(), inp["doc_adj"])
    # the one-launch form (gh_ref_depad): same ids, the same adjacency in packed form, the node-compact plan attached
    from get_amd.ops import PackedAdj
    kf = kargs_from_reference_tensors(T(inp["query_lens"]), T(inp["document"]), T(adj_padded), T(inp["query_adj"]),
                                      T(inp["evd_counts"]), T(inp["doc_sources"]), T(inp["query_sources"]))
    assert isinstance(kf["docs_adj"], PackedAdj) and kf["docs_adj"].plan is not None
    assert np.array_equal(kf["doc_content_without_padding_evidences"].cpu().numpy(), inp["doc_ids"])
    # convert_text's D^-1/2 A D^-1/2 is recognised: bit rows + dinv (the normalised mode of the native batches), no dense values
    assert kf["docs_adj"].vals is None and kf["docs_adj"].dinv is not None
    assert np.abs(kf["docs_adj"].to_dense().cpu().numpy() - inp["doc_adj"]).max() <= 2e-7
    assert np.array_equal(kf["docs_adj"].to_dense().cpu().numpy() != 0, inp["doc_adj"] != 0)
    n_nodes = (inp["doc_ids"] >= 1).sum(1)
    assert kf["docs_adj"].plan.m_real == int(n_nodes.sum())
    assert np.array_equal(kf["docs_adj"].plan.goff.cpu().numpy(), np.concatenate([[0], np.cumsum(n_nodes)]))
    # int32 ids; a claim without evidences; an edge on a padding node (-> packed, but no node-compact plan)
    doc0, adj0, cnt0 = inp["document"].copy(), adj_padded.copy(), np.asarray(inp["evd_counts"]).copy()
    cnt0[1] = 0
    k0 = kargs_from_reference_tensors(T(inp["query_lens"]), T(doc0.astype(np.int32)), T(adj0), T(inp["query_adj"]), T(cnt0),
                                      T(inp["doc_sources"]), T(inp["query_sources"]))
    valid = np.arange(n)[None, :] < cnt0[:, None]
    assert np.array_equal(k0["doc_content_without_padding_evidences"].cpu().numpy(), doc0[valid])
    assert np.abs(k0["docs_adj"].to_dense().cpu().numpy() - adj0[valid]).max() <= 2e-7
    # a hand-crafted adjacency that is NOT the normalised graph of its pattern (one real edge scaled): the whole batch takes the
    # weighted mode with the exact fp32 values of every graph (second launch with force_vals), the node-compact plan stays
    adjw = adj0.copy()
    i1, j1 = np.argwhere(adjw[0, 0] != 0)[1]
    adjw[0, 0, i1, j1] *= 1.01
    kw = kargs_from_reference_tensors(T(inp["query_lens"]), T(doc0), T(adjw), T(inp["query_adj"]), T(cnt0),
                                      T(inp["doc_sources"]), T(inp["query_sources"]))
    assert kw["docs_adj"].vals is not None and kw["docs_adj"].plan is not None
    assert np.array_equal(kw["docs_adj"].to_dense().cpu().numpy(), adjw[valid].astype(np.float32))
    adj0[0, 0, R - 1, 0] = 0.5      # (position R - 1 of this evidence is a padding node)
    assert doc0[0, 0, R - 1] == 0
    k1 = kargs_from_reference_tensors(T(inp["query_lens"]), T(doc0), T(adj0), T(inp["query_adj"]), T(cnt0),
                                      T(inp["doc_sources"]), T(inp["query_sources"]))
    assert k1["docs_adj"].plan is None and float(k1["docs_adj"].to_dense()[0, R - 1, 0]) == 0.5
    assert k1["docs_adj"].vals is not None and np.array_equal(k1["docs_adj"].to_dense().cpu().numpy(), adj0[valid].astype(np.float32))
    # one batch ahead on a side stream (batch.prefetch_reference): same kargs, in order
    from get_amd.batch import prefetch_reference
    items = [(T(inp["query_lens"]), T(inp["document"]), T(adj_padded), T(inp["query_adj"]), T(inp["evd_counts"]), T(inp["doc_sources"]),
              T(inp["query_sources"])), (T(inp["query_lens"]), T(doc0), T(adjw), T(inp["query_adj"]), T(cnt0), T(inp["doc_sources"]),
                                         T(inp["query_sources"]))] * 2
    got = list(prefetch_reference(items))
    assert len(got) == 4
    for kk, ref_k in zip(got, [kf, kw, kf, kw]):
        assert torch.equal(kk["doc_content_without_padding_evidences"], ref_k["doc_content_without_padding_evidences"])
        assert torch.equal(kk["docs_adj"].to_dense(), ref_k["docs_adj"].to_dense())
        assert (kk["docs_adj"].plan is None) == (ref_k["docs_adj"].plan is None)
    nb = NativeBatch(raw["claim_tokens"], raw["claim_len"], raw["evd_tokens"], raw["evd_len"], raw["evd_counts"],
                     raw["doc_sources"], raw["query_sources"], raw["labels"], window=cfg.window, device=DEV)
    q_ids, document, k2 = nb.inputs()
    assert np.array_equal(q_ids.cpu().numpy(), inp["query"]) and np.array_equal(document.cpu().numpy(), inp["document"])
    assert np.abs(k2["docs_adj"].to_dense().cpu().numpy() - inp["doc_adj"]).max() <= 2e-7
    model = build_model(cfg, seed)
    z, _ = load("g7_model_small.npz")
    phi_a = model(T(inp["query"]), T(inp["document"]), **kargs)
    phi_b = model(q_ids, document, **k2)
    phi_f = model(T(inp["query"]), T(inp["document"]), **kf)
    assert np.abs(phi_f.detach().cpu().numpy() - z["phi"]).max() <= 1e-4
    assert np.abs(phi_a.detach().cpu().numpy() - z["phi"]).max() <= 1e-4
    assert np.abs(phi_b.detach().cpu().numpy() - z["phi"]).max() <= 1e-4


def test_chunked_predict_and_kept_observables_do_not_pin_the_activation_arena():
    """the logits / attention weights / scores / keep-sets a caller keeps are views of a small
    observables buffer of their own (ABI 7), not of the multi-GB activation arena: after a no-grad forward the arena is
    free, the model's last_score / last_keep hold a few MB, and batched_predict in chunks keeps one chunk's observables per
    chunk -- "chunking changes nothing but the peak memory" is true again.  A second backward through one fused forward
    raises a clear error instead of an AttributeError."""
    from bench import build_workload
    from get_amd.batch import batched_predict
    wl = build_workload(batch=16, n_evd=30, seed=91, device=DEV)
    model, nb = wl["model"].train(False), wl["batches"][0]
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    base = torch.cuda.memory_allocated()
    with torch.no_grad():
        q, d, k = nb.inputs()
        phi = model(q, d, **k)
    torch.cuda.synchronize()
    held = torch.cuda.memory_allocated() - base
    assert held < 32 << 20, f"a no-grad forward left {held / 2**20:.0f} MiB allocated: the arena is pinned by a kept view"
    del phi
    base = torch.cuda.memory_allocated()
    peak0 = torch.cuda.max_memory_allocated()
    torch.cuda.reset_peak_memory_stats()
    phi_c, words_c, evd_c = batched_predict(model, nb, claims_per_call=4)
    torch.cuda.synchronize()
    kept = torch.cuda.memory_allocated() - base
    assert kept < 64 << 20, f"chunked predict keeps {kept / 2**20:.0f} MiB: per-chunk arenas are held until the final cat"
    peak_chunked = torch.cuda.max_memory_allocated() - base
    torch.cuda.reset_peak_memory_stats()
    phi_f, words_f, evd_f = batched_predict(model, nb)
    torch.cuda.synchronize()
    peak_full = torch.cuda.max_memory_allocated() - base
    assert peak_chunked < 0.6 * peak_full, (peak_chunked, peak_full)
    assert float((phi_c - phi_f).abs().max()) <= 1e-5 and float((evd_c - evd_f).abs().max()) <= 1e-6
    # second backward through one fused forward: explicit error
    model.train(False)
    q, d, k = nb.inputs()
    loss = torch.nn.functional.cross_entropy(model(q, d, **k), nb.labels)
    loss.backward(retain_graph=True)
    with pytest.raises(RuntimeError, match="ran twice"):
        loss.backward()


def test_attention_weight_gradients_flow_through_the_fused_path(monkeypatch):
    """A loss term on the returned attention weights (they were marked non-differentiable, the term silently
    got zero gradient): gh_get_backward's g_word_w / g_evd_w inputs against the module-by-module path on the `small` case."""
    from get_amd import fused
    grads = {}
    for use_fused in (True, False):
        monkeypatch.setattr(fused, "ENABLED", use_fused)
        cfg, seed = MODEL_CASES["small"]
        model = build_model(cfg, seed)
        raw = make_raw_batch(cfg, seed)
        inp = assemble_inputs(raw, cfg, O.convert_text)
        kargs = to_dev(reference_kargs(inp, torch, output_ranking=True))
        phi, (ww, ew) = model(torch.from_numpy(inp["query"]).to(DEV), torch.from_numpy(inp["document"]).to(DEV), **kargs)
        gw = torch.linspace(-1, 1, ww.numel(), device=DEV).view_as(ww)
        ge = torch.linspace(1, -1, ew.numel(), device=DEV).view_as(ew)
        ((ww * gw).sum() + (ew * ge).sum() + 0.0 * phi.sum()).backward()
        grads[use_fused] = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
    assert set(grads[True]) == set(grads[False]) and len(grads[True]) >= 30
    moved = 0
    for k, g in grads[False].items():
        err = float((grads[True][k] - g).abs().max())
        assert err <= 2e-5 * float(g.abs().max()) + 1e-7, (k, err)
        moved += int(float(g.abs().max()) > 0)
    assert moved >= 25, "the attention-weight loss produced no gradient"


def test_batched_predict_equals_per_claim_predict():
    """Row (f)3: one ragged forward over all claims == the reference-style B=1 evaluation loop."""
    from get_amd.batch import NativeBatch, batched_predict
    cfg, seed = MODEL_CASES["small"]
    model = build_model(cfg, seed)
    raw = make_raw_batch(cfg, seed)
    mk = lambda sl_c, sl_e, cnt, b: NativeBatch(raw["claim_tokens"][sl_c], raw["claim_len"][sl_c], raw["evd_tokens"][sl_e],
                                                raw["evd_len"][sl_e], cnt, raw["doc_sources"][sl_c], raw["query_sources"][sl_c],
                                                raw["labels"][sl_c], window=cfg.window, device=DEV)
    full = mk(slice(None), slice(None), raw["evd_counts"], cfg.batch)
    phi, word_w, evd_w = batched_predict(model, full)
    offs = np.concatenate([[0], np.cumsum(raw["evd_counts"])])
    for b in range(cfg.batch):
        one = mk(slice(b, b + 1), slice(offs[b], offs[b + 1]), raw["evd_counts"][b:b + 1], 1)
        q, d, k = one.inputs()
        p1, (w1, e1) = model.predict(q, d, **dict(k, output_ranking=True))
        assert float((p1[0] - phi[b]).detach().abs().max()) <= 1e-5
        assert float((w1 - word_w[b]).abs().max()) <= 1e-6 and float((e1[0] - evd_w[b]).abs().max()) <= 1e-6
        assert torch.allclose(word_w[b].sum(1), torch.ones_like(word_w[b].sum(1)), atol=1e-5)


@pytest.mark.parametrize("name", ["small", "small_claimsrc"])
def test_batched_predict_vs_reference_fixture_and_eval_protocol(name):
    """Row (f)3 against the REFERENCE: `batched_predict` (one ragged forward, and in chunks of 2 claims) must reproduce
    G7's logits / word weights / evidence weights, and so must the evaluation protocol of
    char_man_fitter_query_repr1.py:298-349 -- one claim per forward, int32 ids and lengths, output_ranking, weights
    summing to one within 1e-5 (:433, :448)."""
    from get_amd.batch import NativeBatch, batched_predict
    z, meta = load(f"g7_model_{name}.npz")
    cfg, seed = MODEL_CASES[name]
    model = build_model(cfg, seed)
    raw = make_raw_batch(cfg, seed)
    nb = NativeBatch(raw["claim_tokens"], raw["claim_len"], raw["evd_tokens"], raw["evd_len"], raw["evd_counts"],
                     raw["doc_sources"], raw["query_sources"], raw["labels"], window=cfg.window, device=DEV)
    counts = raw["evd_counts"]
    offs = np.concatenate([[0], np.cumsum(counts)])
    for chunk in (0, 2):
        phi, word_w, evd_w = batched_predict(model, nb, claims_per_call=chunk)
        assert np.abs(phi.cpu().numpy() - z["phi"]).max() <= 1e-4
        assert np.abs(evd_w.cpu().numpy() - z["evd_w"]).max() <= 1e-5
        assert len(word_w) == cfg.batch
        for b in range(cfg.batch):
            assert np.abs(word_w[b].cpu().numpy() - z["word_w"][offs[b]:offs[b + 1]]).max() <= 1e-5
    # the reference's own loop: B = 1, int32 tensors, dense adjacency from the host-side assembly
    inp = assemble_inputs(raw, cfg, O.convert_text)
    for b in range(cfg.batch):
        e = slice(int(offs[b]), int(offs[b + 1]))
        k = {"query_lens": torch.from_numpy(inp["query_lens"][b:b + 1]).int().to(DEV), "docs_lens": None,
             "doc_lens_indices": None,
             "doc_content_without_padding_evidences": torch.from_numpy(inp["doc_ids"][e]).int().to(DEV),
             "evd_cnt_each_query": torch.from_numpy(counts[b:b + 1]).to(DEV), "fixed_num_evidences": cfg.fixed_num_evidences,
             "query_adj": torch.from_numpy(inp["query_adj"][b:b + 1]).to(DEV),
             "docs_adj": torch.from_numpy(inp["doc_adj"][e]).to(DEV),
             "doc_sources": torch.from_numpy(inp["doc_sources"][b:b + 1]).int().to(DEV),
             "query_sources": torch.from_numpy(inp["query_sources"][b:b + 1]).int().to(DEV), "output_ranking": True}
        with torch.no_grad():
            p1, (w1, e1) = model.predict(torch.from_numpy(inp["query"][b:b + 1]).int().to(DEV),
                                         torch.from_numpy(inp["document"][b:b + 1]).int().to(DEV), **k)
        assert np.abs(p1.cpu().numpy() - z["phi"][b:b + 1]).max() <= 1e-4
        assert np.abs(w1.cpu().numpy() - z["word_w"][e]).max() <= 1e-5 and np.abs(e1.cpu().numpy() - z["evd_w"][b:b + 1]).max() <= 1e-5
        assert abs(float(w1.sum(1).mean()) - 1.0) <= 1e-5 and abs(float(e1.sum(1).mean()) - 1.0) <= 1e-5


def test_trainer_checkpoint_resume_is_bit_identical():
    """Row (f)4: model state_dict + FlatTrainer.state_dict() resume reproduces the next step exactly (eval mode)."""
    from get_amd.batch import NativeBatch
    from get_amd.dist import FlatTrainer
    cfg, seed = MODEL_CASES["small"]
    raw = make_raw_batch(cfg, seed)
    nb = NativeBatch(raw["claim_tokens"], raw["claim_len"], raw["evd_tokens"], raw["evd_len"], raw["evd_counts"],
                     raw["doc_sources"], raw["query_sources"], raw["labels"], window=cfg.window, device=DEV)

    def step(model, tr):
        tr.zero_grad()
        q, d, k = nb.inputs()
        torch.nn.functional.cross_entropy(model(q, d, **k), nb.labels).backward()
        tr.step()

    m1 = build_model(cfg, seed)
    t1 = FlatTrainer(m1)
    step(m1, t1)
    ck_model = {k: v.detach().cpu().clone() for k, v in m1.state_dict().items()}
    ck_opt = t1.state_dict()
    step(m1, t1)
    m2 = build_model(cfg, seed + 1)                 # different init, then restored
    m2.load_state_dict(ck_model, strict=True)
    t2 = FlatTrainer(m2)
    t2.load_state_dict(ck_opt)
    step(m2, t2)
    # split-K partial sums are order-deterministic (no atomics), so the resumed step matches bit for bit
    for (k1, p1), (k2, p2) in zip(m1.named_parameters(), m2.named_parameters()):
        if k1 in t1.live_names:
            assert torch.equal(p1.detach(), p2.detach()), k1


def test_graph_cache_build_and_cached_batches_match_native_batch(tmp_path):
    """the packed graph store built on the device equals the oracle's convert_text for every text
    (ids / node counts bit-exact, adjacency values 1e-7), survives a save/load round trip, and a batch gathered
    from it runs the model to the same logits as the per-step graph build of NativeBatch (both layouts)."""
    from get_amd.batch import NativeBatch
    from get_amd.graph_cache import CachedBatcher, GraphCache
    from get_amd.synth import SynthConfig
    cfg = SynthConfig(batch=5, n_evd=0, vocab=700, n_article_src=40, n_claim_src=10, evd_counts=[3, 30, 1, 7, 12])
    seed = 99
    model = build_model(cfg, seed)
    raw = make_raw_batch(cfg, seed)
    inp = assemble_inputs(raw, cfg, O.convert_text)
    b1 = int(raw["evd_counts"].sum())
    ckeys = [f"c{i}" for i in range(cfg.batch)]
    ekeys = list(range(1000, 1000 + b1))
    cc = GraphCache.build(ckeys, raw["claim_tokens"], raw["claim_len"], cfg.window, device=DEV)
    ec = GraphCache.build(ekeys, raw["evd_tokens"], raw["evd_len"], cfg.window, device=DEV, chunk=16)   # several launches
    assert np.array_equal(ec.node_ids.cpu().numpy(), inp["doc_ids"]) and np.array_equal(ec.n_nodes_host, inp["doc_lens"])
    assert np.array_equal(cc.node_ids.cpu().numpy(), inp["query"]) and np.array_equal(cc.n_nodes_host, inp["query_lens"])
    assert float((ec.dense(np.arange(b1)).cpu() - torch.from_numpy(inp["doc_adj"])).abs().max()) <= 1e-7
    p = str(tmp_path / "evd_cache.npz")
    ec.save(p)
    ec = GraphCache.load(p, device=DEV)
    rel, last = {}, 0
    for q, c in zip(ckeys, raw["evd_counts"]):
        rel[q] = ekeys[last:last + int(c)]
        last += int(c)
    src = torch.from_numpy(raw["doc_sources"]).to(DEV)
    qsrc = torch.from_numpy(raw["query_sources"]).to(DEV)
    nb = NativeBatch(raw["claim_tokens"], raw["claim_len"], raw["evd_tokens"], raw["evd_len"], raw["evd_counts"],
                     raw["doc_sources"], raw["query_sources"], raw["labels"], window=cfg.window, n_max=cfg.fixed_num_evidences,
                     device=DEV, compact=True)
    q0, d0, k0 = nb.inputs()
    phi0 = model(q0, d0, **k0)
    for compact in (True, False):
        bt = CachedBatcher(cc, ec, rel, n_max=cfg.fixed_num_evidences, compact=compact)
        q1, d1, k1 = bt.inputs(ckeys, doc_sources=src, query_sources=qsrc)
        assert torch.equal(q1, q0) and torch.equal(d1, d0)
        phi1 = model(q1, d1, **k1)
        assert float((phi1 - phi0).abs().max()) <= (0.0 if compact else 2e-6)
    # a permuted sub-batch of claims equals the corresponding rows
    bt = CachedBatcher(cc, ec, rel, n_max=cfg.fixed_num_evidences)
    sel = [3, 1]
    q2, d2, k2 = bt.inputs([ckeys[i] for i in sel], doc_sources=src[sel], query_sources=qsrc[sel])
    phi2 = model(q2, d2, **k2)
    assert float((phi2 - phi0[sel]).abs().max()) <= 2e-6


@pytest.mark.parametrize("name", ["small", "small_claimsrc"])
def test_early_gradients_are_final_at_the_milestone(name):
    """dist.FlatTrainer overlaps the all-reduce of the 'early' bucket range with the rest of the backward pass; that is
    only legal if every early gradient is complete when the milestone hook fires (autograd's execution order).
    Snapshot the range inside the hook and compare it with the final bucket, in both row layouts."""
    from get_amd import ops
    from get_amd.dist import FlatTrainer, LATE_PREFIXES
    cfg, seed = MODEL_CASES[name]
    model = build_model(cfg, seed).train(True)
    tr = FlatTrainer(model)
    assert 0 < tr.n_early < tr.numel
    assert all(n.startswith(LATE_PREFIXES) for n in tr.live_names[len([n for n in tr.live_names if not n.startswith(LATE_PREFIXES)]):])
    raw = make_raw_batch(cfg, seed)
    inp = assemble_inputs(raw, cfg, O.convert_text)
    labels = torch.from_numpy(inp["labels"]).to(DEV)
    for compact in (False, True):
        kargs = to_dev(reference_kargs(inp, torch))
        da, d_ids, d_n = ops.graph_build(torch.from_numpy(raw["evd_tokens"]).to(DEV), torch.from_numpy(raw["evd_len"]).to(DEV), cfg.window)
        if compact:
            da = da.with_plan(ops.RaggedPlan(d_n, d_ids, int(d_n.sum().item())))
        kargs["docs_adj"] = da
        snaps = []
        # (the early range is final in stream order on BOTH streams of the backward: join the auxiliary one, as the
        # trainer's own hook does before it starts the collective)
        model.ggnn_with_gsl.grad_milestone_hook = lambda: (ops.side_join(), snaps.append(tr.flat_g[:tr.n_early].clone()))
        tr.zero_grad()
        phi = model(torch.from_numpy(inp["query"]).to(DEV), torch.from_numpy(inp["document"]).to(DEV), **kargs)
        torch.nn.functional.cross_entropy(phi, labels).backward()
        assert len(snaps) == 1
        assert torch.equal(snaps[0], tr.flat_g[:tr.n_early]), f"an early gradient changed after the milestone (compact={compact})"
        assert float(tr.flat_g[:tr.n_early].abs().sum()) > 0 and float(tr.flat_g[tr.n_early:].abs().sum()) > 0
    model.ggnn_with_gsl.grad_milestone_hook = None


def test_training_step_keeps_large_gemms_on_the_fast_kernel():
    """Performance guard: with the flat-bucket trainer (parameters and gradients are views into flat buffers) no GEMM
    with >= 1 GFLOP of work may fall back to the generic scalar-load kernel -- that only happens when an operand view
    is misaligned (a 2-element bias once shifted a whole parameter group off its 16-byte alignment)."""
    from bench import build_workload
    from get_amd import _lib
    from get_amd.dist import FlatTrainer
    wl = build_workload(batch=8, n_evd=30, seed=5, device=DEV)
    model = wl["model"].train(True)
    tr = FlatTrainer(model)
    assert all(p.data_ptr() % 16 == 0 and p.grad.data_ptr() % 16 == 0 for p in tr.params)
    for _ in range(2):
        tr.zero_grad()
        q, d, k = wl["make_inputs"]()
        _lib.gemm_path_counters(reset=True)
        torch.nn.functional.cross_entropy(model(q, d, **k), wl["labels"]).backward()
        c = _lib.gemm_path_counters()
        tr.step()
    assert c["fast"] >= 30, c
    assert c["generic_large"] == 0, c


def test_training_loop_overfits_a_small_batch():
    """End-to-end sanity of the training path beyond single-step parity: native batches, node-compact layout, training
    mode with the fused dropout, flat-bucket Adam -- 60 steps on one small batch must drive the loss down."""
    from bench import build_workload
    from get_amd.dist import FlatTrainer
    torch.manual_seed(7)
    wl = build_workload(batch=6, n_evd=0, seed=13, device=DEV)
    model = wl["model"].train(True)
    tr = FlatTrainer(model, lr=1e-3, weight_decay=0.0)
    losses = []
    for _ in range(60):
        tr.zero_grad()
        q, d, k = wl["make_inputs"]()
        loss = torch.nn.functional.cross_entropy(model(q, d, **k), wl["labels"])
        loss.backward()
        tr.step()
        losses.append(float(loss.item()))
    assert all(np.isfinite(losses))
    assert np.mean(losses[-5:]) < 0.35 * np.mean(losses[:3]), losses[::6]


def test_native_batch_counts_nodes_like_convert_text():
    """m_real sizes every launch of the compact layout: it must equal the sum of convert_text's `length_`
    (interactions.py:351), also for ragged evidence counts, short texts and repeated tokens."""
    from get_amd.batch import NativeBatch
    from get_amd.synth import SynthConfig
    cfg = SynthConfig(batch=5, n_evd=0, vocab=60, evd_counts=[1, 7, 30, 2, 11])
    raw = make_raw_batch(cfg, 3)
    raw["evd_len"][0] = 1
    raw["evd_tokens"][1, :] = 5                       # one token repeated: a single node
    nb = NativeBatch(raw["claim_tokens"], raw["claim_len"], raw["evd_tokens"], raw["evd_len"], raw["evd_counts"],
                     raw["doc_sources"], raw["query_sources"], raw["labels"], window=cfg.window,
                     n_max=cfg.fixed_num_evidences, device=DEV)
    want = [O.convert_text([int(t) for t in row], cfg.len_right, int(n), cfg.window)[2]
            for row, n in zip(raw["evd_tokens"], raw["evd_len"])]
    assert nb.m_real == sum(want) and want[0] == 1 and want[1] == 1
    assert nb.b1 == 51 and nb.compact in (True, False)


def test_claim_branch_backward_on_the_side_stream_is_joined_without_side_wgrad(monkeypatch):
    """with a FlatTrainer the claim cell's backward writes straight into the flat bucket and returns nothing to
    autograd, so the engine never joins the auxiliary stream it ran on; the only join used to come from ops._side_wgrad.
    With GET_AMD_WGRAD_STREAM off nothing ordered those writes against the caller's stream.  Made deterministic here: the
    side stream is kept busy for ~100 ms before backward(), so the claim backward is still pending when backward()
    returns -- the bucket read on the caller's stream right after must nevertheless be complete."""
    from get_amd import ops
    from get_amd.dist import FlatTrainer
    cfg, model, inp, phi, ww, ew, loss = run_case("small", native_graphs="compact")
    trainer = FlatTrainer(model, lr=1e-4, weight_decay=1e-3)
    ops.bump_weight_epoch()
    model.train(False)
    monkeypatch.setattr(ops, "WGRAD_SIDE_STREAM", False)
    query = torch.from_numpy(inp["query"]).to(DEV)
    document = torch.from_numpy(inp["document"]).to(DEV)
    labels = torch.from_numpy(inp["labels"]).to(DEV)
    big = torch.randn(6144, 6144, device=DEV)
    buckets = []
    for busy in (False, True, True):
        trainer.zero_grad()
        kargs = to_dev(reference_kargs(inp, torch, output_ranking=False))
        out = model(query, document, **kargs)
        lossv = torch.nn.functional.cross_entropy(out, labels)
        if busy:
            torch.cuda.synchronize()
            with torch.cuda.stream(ops.side_stream(DEV)):
                for _ in range(40):
                    big = (big @ big).clamp_(-1, 1)
        lossv.backward()
        buckets.append(trainer.flat_g.clone())          # caller's stream, no synchronize
    torch.cuda.synchronize()
    assert float(buckets[0].abs().max()) > 0
    for b in buckets[1:]:
        assert torch.equal(b, buckets[0])


@pytest.mark.parametrize("trainer_bucket", [False, True])
def test_frozen_article_source_table_keeps_every_other_gradient(trainer_bucket):
    """evd_assemble_bwd takes the table width as the row pitch of its incoming gradient; with a frozen
    article_source_embs (no table gradient wanted) the pitch used to collapse to the avg width and every slot after the
    first read the wrong row.  Freezing the table must leave every other gradient exactly as it was."""
    from get_amd import ops
    from get_amd.dist import FlatTrainer
    grads = {}
    for frozen in (False, True):
        cfg, model, inp, phi, ww, ew, loss = run_case("small", native_graphs="compact")
        model.article_source_embs.weight.requires_grad_(not frozen)
        if trainer_bucket:
            FlatTrainer(model)
            ops.bump_weight_epoch()
        query = torch.from_numpy(inp["query"]).to(DEV)
        document = torch.from_numpy(inp["document"]).to(DEV)
        kargs = to_dev(reference_kargs(inp, torch, output_ranking=False))
        out = model(query, document, **kargs)
        torch.nn.functional.cross_entropy(out, torch.from_numpy(inp["labels"]).to(DEV)).backward()
        torch.cuda.synchronize()
        grads[frozen] = {k: p.grad.clone() for k, p in model.named_parameters()
                         if p.grad is not None and p.requires_grad and float(p.grad.abs().max()) > 0}
    assert "article_source_embs.weight" in grads[False] and "article_source_embs.weight" not in grads[True]
    for k, g in grads[True].items():
        assert torch.equal(g, grads[False][k]), k
    assert len(grads[True]) == len(grads[False]) - 1


def test_bump_weight_epoch_drops_frozen_derived_entries():
    """a raw `.data` write followed by ops.bump_weight_epoch() must also refresh entries derived from tensors
    the trainer never touches (the scorer's packed gates): the keep-sets have to follow the new scorer weights."""
    from get_amd import ops
    cfg, model, inp, phi, ww, ew, loss = run_case("small", native_graphs="compact")
    s0 = model.ggnn_with_gsl.last_score.clone()
    with torch.no_grad():
        model.ggnn_with_gsl.word_scorer1.linearz0.linear.bias.data += 3.0        # raw write: no version bump
    ops.bump_weight_epoch()
    query = torch.from_numpy(inp["query"]).to(DEV)
    document = torch.from_numpy(inp["document"]).to(DEV)
    kargs = to_dev(reference_kargs(inp, torch, output_ranking=False))
    with torch.no_grad():
        model(query, document, **kargs)
    assert not torch.equal(model.ggnn_with_gsl.last_score, s0), "the scorer still ran on its stale packed gates"


@pytest.mark.parametrize("name", list(MODEL_CASES))
@pytest.mark.parametrize("native", [False, True, "compact"])
def test_composite_entry_points_equal_the_module_by_module_path(name, native, monkeypatch):
    """get_amd/fused.py (gh_get_forward / gh_get_backward: the whole model in one library call each) against the
    module-by-module path on the same inputs: same kernels underneath, so logits, attention weights, scores, keep-sets
    and every gradient agree to fp32 summation-order noise (the composite sums the claim vector's gradient in a different
    order and projects the word attention's left input once per claim instead of once per pair)."""
    from get_amd import fused
    res = {}
    for on in (False, True):
        monkeypatch.setattr(fused, "ENABLED", on)
        cfg, model, inp, phi, ww, ew, loss = run_case(name, native_graphs=native)
        assert (getattr(model, "_gh_binding", None) is not None) == on, "the wrong path ran"
        loss.backward()
        torch.cuda.synchronize()
        res[on] = dict(phi=phi.detach().clone(), ww=ww.detach().clone(), ew=ew.detach().clone(),
                       score=model.ggnn_with_gsl.last_score.clone(), keep=model.ggnn_with_gsl.last_keep.clone(),
                       grads={k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None})
    a, b = res[False], res[True]
    assert torch.equal(a["keep"], b["keep"])
    for k in ("phi", "ww", "ew", "score"):
        assert float((a[k] - b[k]).abs().max()) <= 2e-6 * max(1.0, float(a[k].abs().max())), k
    assert set(a["grads"]) == set(b["grads"])
    for k, g in a["grads"].items():
        scale = max(float(g.abs().max()), 1e-8)
        assert float((g - b["grads"][k]).abs().max()) <= 2e-5 * scale, k


def test_composite_path_with_flat_trainer_matches_autograd_gradients():
    """The composite backward accumulates straight into the FlatTrainer bucket (no gradient returned to autograd): the
    bucket must equal the gradients plain autograd collects from the same call, for three steps in a row (persistent
    transposes, cached descriptor)."""
    from get_amd import ops
    from get_amd.dist import FlatTrainer
    cfg, model, inp, phi, ww, ew, loss = run_case("small", native_graphs="compact")
    loss.backward()
    ref = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
    trainer = FlatTrainer(model, lr=0.0, weight_decay=0.0)
    ops.bump_weight_epoch()
    query = torch.from_numpy(inp["query"]).to(DEV)
    document = torch.from_numpy(inp["document"]).to(DEV)
    labels = torch.from_numpy(inp["labels"]).to(DEV)
    for _ in range(3):
        trainer.zero_grad()
        kargs = to_dev(reference_kargs(inp, torch, output_ranking=False))
        ops.cross_entropy(model(query, document, **kargs), labels).backward()
        for k, p in model.named_parameters():
            if k in ref:
                scale = max(float(ref[k].abs().max()), 1e-8)
                assert float((p.grad - ref[k]).abs().max()) <= 2e-5 * scale, k
        trainer.step()          # lr = 0: parameters unchanged, but transposes are refreshed in place and epochs move on


def test_fused_cross_entropy_matches_torch():
    from get_amd import ops
    g = torch.Generator().manual_seed(3)
    for b, c in ((32, 2), (7, 5), (300, 3)):
        phi = (torch.randn(b, c, generator=g) * 3).to(DEV).requires_grad_(True)
        y = torch.randint(0, c, (b,), generator=g).to(DEV)
        l1 = ops.cross_entropy(phi, y)
        (l1 * 1.7).backward()
        g1 = phi.grad.clone()
        phi.grad = None
        l2 = torch.nn.functional.cross_entropy(phi, y)
        (l2 * 1.7).backward()
        assert abs(l1.item() - l2.item()) <= 1e-6 * max(1.0, abs(l2.item()))
        assert float((g1 - phi.grad).abs().max()) <= 1e-6


def test_native_batch_prepare_call_equals_the_separate_launches():
    """NativeBatch.inputs() (one gh_get_prepare call into persistent buffers) against ops.graph_build + ops.RaggedPlan +
    the index_copy_ it replaces: bit-exact, also on the second call (buffers rewritten in place)."""
    from get_amd import ops
    from get_amd.batch import NativeBatch
    from get_amd.keywords import KeyWordSettings as K
    cfg, seed = MODEL_CASES["small"]
    raw = make_raw_batch(cfg, seed)
    for compact in (True, False):
        nb = NativeBatch(raw["claim_tokens"], raw["claim_len"], raw["evd_tokens"], raw["evd_len"], raw["evd_counts"],
                         raw["doc_sources"], raw["query_sources"], raw["labels"], window=cfg.window,
                         n_max=cfg.fixed_num_evidences, device=DEV, compact=compact)
        for _ in range(2):
            q_ids, document, kargs = nb.inputs()
            qa, q2, qn = ops.graph_build(nb.claim_tokens, nb.claim_len, cfg.window)
            da, d2, dn = ops.graph_build(nb.evd_tokens, nb.evd_len, cfg.window)
            assert torch.equal(q_ids, q2) and torch.equal(kargs[K.Query_lens], qn)
            assert torch.equal(kargs[K.DocContentNoPaddingEvidence], d2)
            assert torch.equal(kargs[K.Query_Adj].bits, qa.bits) and torch.equal(kargs[K.Query_Adj].dinv, qa.dinv)
            assert torch.equal(kargs[K.Evd_Docs_Adj].bits, da.bits) and torch.equal(kargs[K.Evd_Docs_Adj].dinv, da.dinv)
            doc_ref = torch.zeros((nb.b * nb.n_max, d2.shape[1]), device=DEV, dtype=torch.int32)
            doc_ref.index_copy_(0, nb._slot, d2)
            assert torch.equal(document.reshape(doc_ref.shape), doc_ref)
            plan = kargs[K.Evd_Docs_Adj].plan
            assert (plan is not None) == compact
            if compact:
                ref = ops.RaggedPlan(dn, d2, nb.m_real)
                for f in ("goff", "rowg", "src", "cids", "maskf"):
                    assert torch.equal(getattr(plan, f), getattr(ref, f)), f


def test_dense_hand_over_is_compacted_and_guarded(monkeypatch):
    """fused._plan_from_dense: a dense adjacency from the reference API gets the node-compact plan from its ids (same
    results as the padded layout, to summation-order noise); an adjacency whose PADDING rows carry edges -- legal input the
    reference never produces -- must keep the padded layout, and then equals the module-by-module path exactly as before."""
    from get_amd import fused
    cfg, seed = MODEL_CASES["small"]
    seen = []
    orig = fused._plan_from_dense
    monkeypatch.setattr(fused, "_plan_from_dense", lambda a, d: (seen.append(orig(a, d)), seen[-1])[1])
    out = {}
    for auto in (False, True):
        monkeypatch.setattr(fused, "AUTO_COMPACT", auto)
        cfg_, model, inp, phi, ww, ew, loss = run_case("small", native_graphs=False)
        loss.backward()
        out[auto] = (phi.detach().clone(), ww.detach().clone(), model.ggnn_with_gsl.last_keep.clone(),
                     {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None})
    assert len(seen) == 1 and seen[0] is not None and 0 < seen[0].m_real < inp["doc_ids"].size
    assert torch.equal(out[False][2], out[True][2])
    assert float((out[False][0] - out[True][0]).abs().max()) <= 2e-6 and float((out[False][1] - out[True][1]).abs().max()) <= 2e-6
    for k, g in out[False][3].items():
        assert float((g - out[True][3][k]).abs().max()) <= 2e-5 * max(float(g.abs().max()), 1e-8), k
    # guard: give one padding node of one evidence graph a self-loop
    model = build_model(cfg, seed)
    raw = make_raw_batch(cfg, seed)
    inp = assemble_inputs(raw, cfg, O.convert_text)
    adj = inp["doc_adj"].copy()
    g0 = 0
    pad = int((inp["doc_ids"][g0] >= 1).sum())
    assert pad < adj.shape[1]
    adj[g0, pad, pad] = 0.5
    inp2 = dict(inp, doc_adj=adj)
    kargs = to_dev(reference_kargs(inp2, torch, output_ranking=False))
    q, d = torch.from_numpy(inp["query"]).to(DEV), torch.from_numpy(inp["document"]).to(DEV)
    seen.clear()
    with torch.no_grad():
        phi_c = model(q, d, **kargs)
        assert seen == [None], "a padding node with an edge must keep the padded layout"
        monkeypatch.setattr(fused, "ENABLED", False)
        phi_m = model(q, d, **kargs)
    assert float((phi_c - phi_m).abs().max()) <= 2e-6


# ---------------------------------------------------------------- opt-in fp32x3 modes (3-way bf16 splits on the bf16 MFMA)
@pytest.fixture(params=["fp32x3", "fp32x3p"])
def x3_mode(request):
    from get_amd import _lib, ops
    _lib.set_gemm_mode(request.param)
    yield request.param
    ops.bump_weight_epoch()          # (frees the pre-split images while the mode is still set)
    _lib.set_gemm_mode("fp32")


@pytest.mark.parametrize("m,k,n", [(9000, 300, 300), (8300, 600, 640), (8200, 76, 300)])
def test_fp32x3_linear_is_fp32_accurate_and_follows_weight_updates(x3_mode, m, k, n):
    """Both fp32x3 modes against a float64 product: the error is the fp32 kernel's (six of nine cross terms of the 3-way
    bf16 splits; bound 1e-5 of the largest output, the exact-fp32 kernel measures 5e-6 on these shapes).  The pre-split
    mode must notice rewritten weights: through ops.bump_weight_epoch (images dropped) and through the library call a
    trainer makes (gh_weights_changed + gh_fp32x3_refresh)."""
    from get_amd import ops
    rng = np.random.default_rng(m + k)
    x = torch.from_numpy(rng.standard_normal((m, k)).astype(np.float32)).to(DEV).requires_grad_(True)
    w = torch.from_numpy((rng.standard_normal((n, k)) / np.sqrt(k)).astype(np.float32)).to(DEV).requires_grad_(True)
    b = torch.from_numpy(rng.standard_normal((n,)).astype(np.float32)).to(DEV)
    g = torch.from_numpy(rng.standard_normal((m, n)).astype(np.float32)).to(DEV)

    def check():
        x.grad = None
        y = ops.linear(x, w, b)
        (y * g).sum().backward()
        ref = x.detach().double() @ w.detach().double().t() + b.double()
        assert float((y.detach().double() - ref).abs().max()) <= 1e-5 * max(1.0, float(ref.abs().max()))
        dx = g.double() @ w.detach().double()
        assert float((x.grad.double() - dx).abs().max()) <= 1e-5 * max(1.0, float(dx.abs().max()))

    check()
    with torch.no_grad():
        w.mul_(-1.7)
    ops.bump_weight_epoch()
    check()
    w.data.add_(0.25)                 # a raw-pointer style update: no _version bump, the trainer's calls announce it
    ops._bump_trainer_epoch()
    ops.refresh_transposes([w])       # (transposes first, then -- in the pre-split mode -- every image in one launch)
    check()


def test_fp32x3p_training_steps_track_fp32(monkeypatch):
    """Three FlatTrainer steps of the bench-shaped model (>= 8192 node rows: the big-tile launches read pre-split weights,
    refreshed by the trainer after every update) against the same steps in exact fp32: losses within 2e-6 relative (a stale
    image would show as ~1e-3), parameters within 1e-4 absolute after the third update (lr 1e-3; Adam's g / sqrt(v)
    amplifies the relative noise of near-zero gradients, so this bound is loose by construction)."""
    from bench import build_workload
    from get_amd import _lib, ops
    from get_amd.dist import FlatTrainer
    out = {}
    for mode in ("fp32", "fp32x3p"):
        _lib.set_gemm_mode(mode)
        try:
            wl = build_workload(batch=12, n_evd=30, seed=77, device=DEV)
            model = wl["model"].train(False)
            ops.bump_weight_epoch()
            trainer = FlatTrainer(model, lr=1e-3, weight_decay=0.0)
            losses = []
            for _ in range(3):
                trainer.zero_grad()
                loss = ops.cross_entropy(model(wl["query"], wl["document"], **wl["kargs"]), wl["labels"])
                loss.backward()
                losses.append(float(loss.detach()))
                trainer.step()
            out[mode] = (losses, {k: p.detach().clone() for k, p in model.named_parameters()})
        finally:
            ops.bump_weight_epoch()
            _lib.set_gemm_mode("fp32")
    la, lb = out["fp32"][0], out["fp32x3p"][0]
    assert la[0] != la[2], "the parameters did not move"
    for a, b in zip(la, lb):
        assert abs(a - b) <= 2e-6 * max(1.0, abs(a)), (la, lb)
    for k, p in out["fp32"][1].items():
        assert float((p - out["fp32x3p"][1][k]).abs().max()) <= 1e-4, k


def test_fp32x3p_follows_a_plain_torch_optimizer():
    """No FlatTrainer: torch.optim.SGD rewrites the parameters in place, which the library cannot see.  The forward's
    guard (ops.fp32x3p_guard: parameter versions) and the transposed-copy cache must mark the pre-split images stale, so
    three steps track the exact-fp32 run (losses within 2e-6 relative; a stale image shows as ~1e-2 at this step size)."""
    from bench import build_workload
    from get_amd import _lib, ops
    out = {}
    for mode in ("fp32", "fp32x3p"):
        _lib.set_gemm_mode(mode)
        try:
            wl = build_workload(batch=10, n_evd=30, seed=78, device=DEV)
            model = wl["model"].train(False)
            ops.bump_weight_epoch()
            opt = torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=0.05)
            losses = []
            for _ in range(3):
                opt.zero_grad(set_to_none=True)
                loss = torch.nn.functional.cross_entropy(model(wl["query"], wl["document"], **wl["kargs"]), wl["labels"])
                loss.backward()
                losses.append(float(loss.detach()))
                opt.step()
            out[mode] = losses
        finally:
            ops.bump_weight_epoch()
            _lib.set_gemm_mode("fp32")
    assert abs(out["fp32"][0] - out["fp32"][2]) > 1e-4 * abs(out["fp32"][0]), "the parameters did not move"
    for a, b in zip(out["fp32"], out["fp32x3p"]):
        assert abs(a - b) <= 2e-6 * max(1.0, abs(a)), out


@pytest.mark.parametrize("name", ["small", "full"])
@pytest.mark.parametrize("native", [True, "compact"])
def test_training_mode_model_vs_oracle_replaying_the_dropout_masks(name, native):
    """The mode the bench line is measured in: train(True), input dropout of the four GGNN cells on (wrapper.py:189-190, p from the
    constructor).  The composite path draws four seeds from torch's CPU generator (fused.py) and applies stateless hash masks inside
    its kernels (projection loader, scorer, dX epilogue, the weight-gradient operand); the oracle replays exactly those masks
    (O.model_forward drop_keep) -- indexed by padded row for the padded layout, by node-compact row for the compact one.  Logits,
    attention weights, scorer scores, keep-sets and every live gradient at the eval-mode tolerances."""
    from get_amd import ops
    cfg, seed = MODEL_CASES[name]
    model = build_model(cfg, seed).train(True)
    raw = make_raw_batch(cfg, seed)
    inp = assemble_inputs(raw, cfg, O.convert_text)
    kargs = to_dev(reference_kargs(inp, torch, output_ranking=True))
    query = torch.from_numpy(inp["query"]).to(DEV)
    document = torch.from_numpy(inp["document"]).to(DEV)
    qa, q_ids, q_n = ops.graph_build(torch.from_numpy(raw["claim_tokens"]).to(DEV), torch.from_numpy(raw["claim_len"]).to(DEV), cfg.window)
    da, d_ids, d_n = ops.graph_build(torch.from_numpy(raw["evd_tokens"]).to(DEV), torch.from_numpy(raw["evd_len"]).to(DEV), cfg.window)
    plan = None
    if native == "compact":
        plan = ops.RaggedPlan(d_n, d_ids, int(d_n.sum().item()))
        da = da.with_plan(plan)
    kargs["query_adj"], kargs["docs_adj"] = qa, da
    torch.manual_seed(20240917)
    seeds = torch.randint(0, 2 ** 31 - 1, (4,)).tolist()          # what fused.prepare will draw: claim, cell1, scorer, cell2
    torch.manual_seed(20240917)
    phi, (ww, ew) = model(query, document, **kargs)
    assert getattr(model, "_gh_binding", None) is not None, "the composite path did not run"
    labels = torch.from_numpy(inp["labels"]).to(DEV)
    torch.nn.functional.cross_entropy(phi, labels).backward()
    torch.cuda.synchronize()
    # --- the same masks on the host
    p_claim = float(model.ggnn4claim_1.dropout.p)
    p_gnn = float(model.ggnn_with_gsl.feat_prop1.dropout.p)
    assert p_claim > 0 and p_gnn > 0
    B, L = inp["query"].shape
    B1, R = inp["doc_ids"].shape
    D, H = cfg.emb_dim, cfg.hidden
    def rows_mask(sd, width, p):          # (B1, R, width) in padded order
        kc = ops.dropout_mask_reference(sd, B1 * R, width, p)
        if plan is None:
            return torch.from_numpy(kc.reshape(B1, R, width))
        k = np.zeros_like(kc)
        k[plan.src.cpu().numpy()] = kc     # compact row -> padded row
        return torch.from_numpy(k.reshape(B1, R, width))
    drop_keep = {"claim": (torch.from_numpy(ops.dropout_mask_reference(seeds[0], B * L, D, p_claim).reshape(B, L, D)), p_claim),
                 "cell1": (rows_mask(seeds[1], D, p_gnn), p_gnn),
                 "scorer": (rows_mask(seeds[2], H, p_gnn), p_gnn),
                 "cell2": (rows_mask(seeds[3], H, p_gnn), p_gnn)}
    T = torch.from_numpy
    po = {k: v.detach().cpu().clone().requires_grad_(v.is_floating_point() and k != "embedding.weight") for k, v in model.state_dict().items()}
    ocfg = dict(cfg.__dict__)
    phi_o, ww_o, ew_o, aux = O.model_forward(po, ocfg, T(inp["query"]), T(inp["document"]), T(inp["query_adj"]), T(inp["doc_ids"]),
                                             T(inp["doc_adj"]), T(inp["query_lens"]), inp["evd_counts"], T(inp["doc_sources"]),
                                             T(inp["query_sources"]), return_aux=True, drop_keep=drop_keep)
    O.cross_entropy(phi_o, T(inp["labels"])).backward()
    real = inp["doc_ids"] >= 1
    assert float((phi.detach().cpu() - phi_o.detach()).abs().max()) <= 1e-4
    ww_h = ww.detach().cpu().numpy()
    if plan is not None and ww_h.shape[0] != B1:
        ww_h = plan.to_padded(ww.detach()).cpu().numpy()
    ww_h = ww_h.reshape(B1, R, -1)
    assert np.abs(ww_h - ww_o.detach().numpy())[real].max() <= 1e-5
    assert float((ew.detach().cpu() - ew_o.detach()).abs().max()) <= 1e-5
    sc_h = model.ggnn_with_gsl.last_score.cpu().numpy().reshape(B1, R)
    assert np.abs(sc_h - aux["score"].detach().numpy())[real].max() <= 1e-5
    n_live = 0
    for k, prm in model.named_parameters():
        go = po[k].grad if k in po else None
        if prm.grad is None or go is None:
            continue
        scale = max(float(go.abs().max()), 1e-8)
        assert float((prm.grad.cpu() - go).abs().max()) <= 1e-3 * scale + 1e-7, k
        n_live += 1
    assert n_live >= 40


def test_unit_seed_backward_equals_loss_backward():
    """ops.backward(loss) hands autograd a cached constant 1 as the root gradient, which the fused cross-entropy recognises by
    address and skips its scale launch: gradients bit-identical to loss.backward(); a scaled loss, or another device scalar as the
    root gradient, still takes the multiply."""
    from get_amd import ops
    g = torch.Generator().manual_seed(11)
    phi0 = (torch.randn(32, 2, generator=g) * 2).to(DEV)
    y = torch.randint(0, 2, (32,), generator=g).to(DEV)
    grads = []
    for mode in ("plain", "unit", "scaled", "other_one"):
        phi = phi0.clone().requires_grad_(True)
        loss = ops.cross_entropy(phi, y)
        if mode == "plain":
            loss.backward()
        elif mode == "unit":
            ops.backward(loss)
        elif mode == "scaled":
            (loss * 3.0).backward()
        else:
            torch.autograd.backward(loss, grad_tensors=[torch.full((), 3.0, device=DEV)])
        grads.append(phi.grad.clone())
    assert torch.equal(grads[0], grads[1])
    assert float((grads[2] - 3.0 * grads[0]).abs().max()) <= 1e-7 and torch.equal(grads[2], grads[3])
    # through the whole model: the composite path's gradients are the same either way
    res = []
    for unit in (False, True):
        cfg, model, inp, phi, ww, ew, _ = run_case("small", native_graphs="compact")
        loss = ops.cross_entropy(phi, torch.from_numpy(inp["labels"]).to(DEV))
        ops.backward(loss) if unit else loss.backward()
        res.append({k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None})
    assert set(res[0]) == set(res[1]) and all(torch.equal(res[0][k], res[1][k]) for k in res[0])


def _g9_kargs():
    """The **kargs the reference fitter handed to net(...) (G9, captured by oracle/make_golden.py from
    char_man_fitter_query_repr1.py:164-258), rebuilt VERBATIM: every key incl. the five forward() ignores, int64 / float64
    dtypes, the two 3-tuples of sort indices, the scalar fixed_num_evidences."""
    z, meta = load("g9_fitter_kargs_small.npz")

    def get(name, d):
        if d["kind"] == "scalar":
            return d["value"]
        if d["kind"] == "tuple":
            return tuple(get(f"{name}::{i}", di) for i, di in enumerate(d["items"]))
        t = torch.from_numpy(np.ascontiguousarray(z[name]))
        assert str(z[name].dtype) == d["dtype"] and list(z[name].shape) == d["shape"]
        return t
    kargs = {k: get(f"k::{k}", meta["desc"][k]) for k in meta["keys"]}
    return z, meta, kargs


def test_g9_captured_fitter_kargs_verbatim_on_the_hip_model():
    """the captured hand-over itself -- not oracle.assemble.reference_kargs' restatement of it -- goes into
    the HIP model: `net(query, document, **kargs)` exactly as char_man_fitter_query_repr1.py:234-253 calls it (dense float64
    adjacencies, int64 ids, 3-tuples, the ignored keys), and must reproduce the reference's own outputs on that batch (G7
    `small`: logits 1e-4, both attention-weight tensors and the scorer scores 1e-5)."""
    z9, meta, kargs = _g9_kargs()
    g7, _ = load("g7_model_small.npz")
    cfg, seed = MODEL_CASES[meta["case"]]
    model = build_model(cfg, seed)
    assert {"query_content_without_padding_evidences", "query_char_source", "doc_char_source", "docs_lens", "fc_labels"} <= set(kargs)
    assert kargs["docs_adj"].dtype == torch.float64 and isinstance(kargs["doc_lens_indices"], tuple) and len(kargs["doc_lens_indices"]) == 3
    q, d = torch.from_numpy(z9["query"]).to(DEV), torch.from_numpy(z9["document"]).to(DEV)
    with torch.no_grad():
        phi = model(q, d, **to_dev(kargs))                                   # the training loop's call: logits only
        assert torch.is_tensor(phi) and phi.shape == (cfg.batch, cfg.num_classes)
        phi2, (ww, ew) = model(q, d, **to_dev(dict(kargs, output_ranking=True)))   # the evaluation loop adds output_ranking (:318)
    assert np.abs(phi.cpu().numpy() - g7["phi"]).max() <= 1e-4
    assert torch.equal(phi, phi2)
    assert np.abs(ww.cpu().numpy() - g7["word_w"]).max() <= 1e-5
    assert np.abs(ew.cpu().numpy() - g7["evd_w"]).max() <= 1e-5
    assert np.abs(model.ggnn_with_gsl.last_score.cpu().numpy() - g7["score"]).max() <= 1e-5


def test_g9_padded_fitter_tensors_through_the_depadding_shim():
    """The same captured batch in the form the fitter holds BEFORE its de-padding loop (char_man_fitter_query_repr1.py:196-223:
    (B, n, R) ids and a (B, n, R, R) float64 adjacency, rebuilt here by scattering G9's de-padded rows back by the evidence
    counts) through get_amd.batch.kargs_from_reference_tensors (gh_ref_depad): same reference outputs (G7)."""
    from get_amd.batch import kargs_from_reference_tensors
    z9, meta, kargs = _g9_kargs()
    g7, _ = load("g7_model_small.npz")
    cfg, seed = MODEL_CASES[meta["case"]]
    model = build_model(cfg, seed)
    counts = kargs["evd_cnt_each_query"].numpy()
    B, n, R = cfg.batch, int(kargs["fixed_num_evidences"]), cfg.len_right
    ids = np.zeros((B, n, R), np.int64)
    adj = np.zeros((B, n, R, R), np.float64)
    last = 0
    for b in range(B):
        c = int(counts[b])
        ids[b, :c] = kargs["doc_content_without_padding_evidences"].numpy()[last:last + c]
        adj[b, :c] = kargs["docs_adj"].numpy()[last:last + c]
        last += c
    assert np.array_equal(ids, z9["document"])             # what the fitter passes as `document` IS the padded id tensor
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    k2 = kargs_from_reference_tensors(kargs["query_lens"].to(DEV), T(ids), T(adj), kargs["query_adj"].to(DEV), kargs["evd_cnt_each_query"].to(DEV),
                                      kargs["doc_sources"].to(DEV), kargs["query_sources"].to(DEV), n_max=n)
    k2["output_ranking"] = True
    with torch.no_grad():
        phi, (ww, ew) = model(torch.from_numpy(z9["query"]).to(DEV), T(ids), **k2)
    assert np.abs(phi.cpu().numpy() - g7["phi"]).max() <= 1e-4
    assert np.abs(ww.cpu().numpy() - g7["word_w"]).max() <= 1e-5
    assert np.abs(ew.cpu().numpy() - g7["evd_w"]).max() <= 1e-5
