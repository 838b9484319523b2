"""Pins the CPU oracle (oracle/get_oracle.py) against golden vectors captured from the
upstream reference by oracle/make_golden.py (G1..G8).  CPU only."""
import json
import os

import numpy as np
import pytest
import torch

from get_amd.synth import make_embeddings, make_raw_batch, make_state_dict, state_dict_shapes
from oracle import cases
from oracle import get_oracle as O
from oracle.assemble import assemble_inputs
from oracle.cases_model import MODEL_CASES
from tests.util import check_grad, dense_from_coo, load

torch.set_num_threads(min(8, os.cpu_count() or 1))


def T(a, grad=False):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return t.requires_grad_(True) if grad else t


# ---------------------------------------------------------------- G1 ----------
def test_g1_convert_text_all_cases():
    z, meta = load("g1_convert_text.npz")
    assert len(meta) >= 60
    for i, m in enumerate(meta):
        ids, adj, n = O.convert_text(z[f"c{i}_tokens"].tolist(), m["fixed_length"], m["length"], m["window"])
        assert n == m["n_nodes"], i
        assert np.array_equal(ids, z[f"c{i}_words"]), i
        exp = dense_from_coo(z, i, m["fixed_length"])
        assert np.array_equal(adj != 0, exp != 0), i
        assert np.abs(adj - exp).max() <= 1e-15, i


def test_g1_known_answers():
    # SURVEY.md 8(c) worked examples (probed on the reference)
    ids, adj, n = O.convert_text([5, 7, 5, 9, 0, 0], 6, 4, 2)
    assert ids.tolist() == [5, 7, 9, 0, 0, 0] and n == 3
    s6 = 1 / np.sqrt(6)
    assert np.allclose(adj[:3, :3], [[1 / 3, s6, s6], [s6, .5, 0], [s6, 0, .5]])
    ids, adj, n = O.convert_text([5, 7, 5, 9, 8, 0], 6, 5, 3)
    assert ids.tolist() == [5, 7, 9, 8, 0, 0] and n == 4
    assert np.allclose(adj[0, :4], [.25, .2887, .25, .2887], atol=1e-4)
    assert np.allclose(adj[1, :4], [.2887, .3333, .2887, 0], atol=1e-4)


# ---------------------------------------------------------------- G2 ----------
@pytest.mark.parametrize("ci", range(len(cases.G2_CASES)))
def test_g2_ggnn_cell(ci):
    z, meta = load("g2_ggnn.npz")
    c = cases.g2_inputs(ci, O.convert_text)
    assert np.array_equal(c["toks"], z[f"c{ci}_tokens"])
    p = {k: T(v, grad=True) for k, v in c["p"].items()}
    x = T(c["x"], grad=True)
    out = O.ggnn_cell(T(c["adj"]).float(), x, p)
    assert np.abs(out.detach().numpy() - z[f"c{ci}_out"]).max() <= 2e-5
    (out * T(c["gw"])).sum().backward()
    check_grad(z, f"c{ci}_dx", x.grad.numpy())
    for k, v in p.items():
        check_grad(z, f"c{ci}_g::{k}", v.grad.numpy())


def test_g2_training_mode_dropout_replay_matches_the_reference():
    """Training mode of the cell (wrapper.py:189-190): the reference's own run with nn.Dropout on the cell input, its drawn keep
    mask captured from the dropout module (oracle/make_golden.py).  The oracle's mask-replay form -- ggnn_cell(keep=(mask, p)), the
    form every training-mode parity test of the HIP path goes through -- must reproduce that output and its gradients."""
    z, meta = load("g2_ggnn.npz")
    ci = 0
    c = cases.g2_inputs(ci, O.convert_text)
    n, r, din = c["x"].shape
    keep = np.unpackbits(z[f"c{ci}_train_keep"])[:n * r * din].reshape(n, r, din).astype(bool)
    p_drop = meta[ci]["drop_p"]
    assert abs(keep.mean() - (1 - p_drop)) < 0.03
    p = {k: T(v, grad=True) for k, v in c["p"].items()}
    x = T(c["x"], grad=True)
    out = O.ggnn_cell(T(c["adj"]).float(), x, p, keep=(torch.from_numpy(keep), p_drop))
    assert np.abs(out.detach().numpy() - z[f"c{ci}_train_out"]).max() <= 2e-5
    (out * T(c["gw"])).sum().backward()
    assert np.abs(x.grad.numpy() - z[f"c{ci}_train_dx"]).max() <= 1e-4 * max(1.0, np.abs(z[f"c{ci}_train_dx"]).max())
    gw = z[f"c{ci}_train_g::proj.linear.weight"]
    assert np.abs(p["proj.linear.weight"].grad.numpy() - gw).max() <= 1e-3 * np.abs(gw).max()
    assert np.abs(x.grad.numpy()[~keep]).max() == 0.0          # dropped inputs receive no gradient


# ---------------------------------------------------------------- G3 ----------
def test_g3_gsl_keep_sets():
    z, meta = load("g3_gsl.npz")
    for ci, m in enumerate(meta):
        score = T(z[f"c{ci}_score"])
        r = m["r"]
        exp_mask = np.unpackbits(z[f"c{ci}_mask"], axis=-1)[..., :r].astype(bool)
        adj = torch.ones(m["b"], r, r)
        out, keep = O.gsl_refine(adj, score, m["rate"])
        got_mask = out.numpy() != 0
        assert keep.sum(-1).tolist() == [m["k"]] * m["b"]
        if not m["ties"]:
            assert np.array_equal(got_mask, exp_mask), ci
        else:
            # tie order is unspecified in the reference (torch.topk); the sets must agree
            # up to members whose score equals the k-th score
            s = z[f"c{ci}_score"][..., 0]
            exp_keep = exp_mask.all(-1)          # kept rows are fully 1 on an all-ones adjacency
            for b in range(m["b"]):
                kth = np.sort(s[b])[::-1][m["k"] - 1]
                strict = s[b] > kth
                assert np.array_equal(keep[b].numpy() & strict, strict)
                assert np.array_equal(exp_keep[b] & strict, strict)
                assert (s[b][keep[b].numpy()] >= kth).all() and exp_keep[b].sum() == m["k"]
    known = z["known_out"][0]
    out, _ = O.gsl_refine(torch.ones(1, 4, 4), torch.tensor([[[.9], [.1], [.8], [.2]]]), 0.5)
    assert np.array_equal(out[0].numpy(), known)
    assert [(i, j) for i in range(4) for j in range(4) if known[i, j] == 0] == [(1, 1), (1, 3), (3, 1), (3, 3)]


# ---------------------------------------------------------------- G4 ----------
@pytest.mark.parametrize("ci", range(len(cases.G4_CASES)))
def test_g4_ggnn_with_gsl(ci):
    z, meta = load("g4_ggnn_gsl.npz")
    m = meta[ci]
    c = cases.g4_inputs(ci, O.convert_text)
    p = {k: T(v, grad=True) for k, v in c["p"].items()}
    x = T(c["x"], grad=True)
    adj = T(c["adj"]).float()
    out, aux = O.ggnn_with_gsl(adj, x, p, "", c["rate"], return_aux=True)
    assert np.abs(aux["score"].detach().numpy() - z[f"c{ci}_score"]).max() <= 1e-5
    r = m["r"]
    exp_nz = np.unpackbits(z[f"c{ci}_adjr_nz"], axis=-1)[..., :r].astype(bool)
    adj_r, _ = O.gsl_refine(adj, None, c["rate"], keep=aux["keep"])
    assert np.array_equal(adj_r.numpy() != 0, exp_nz)
    assert np.abs(out.detach().numpy() - z[f"c{ci}_out"]).max() <= 2e-5
    (out * T(c["gw"])).sum().backward()
    check_grad(z, f"c{ci}_dx", x.grad.numpy())
    for k, v in p.items():
        if k in m["none_grads"]:
            assert v.grad is None or float(v.grad.abs().max()) == 0.0, k   # no gradient through GSL
        else:
            check_grad(z, f"c{ci}_g::{k}", v.grad.numpy())
    assert sorted(m["none_grads"]) == sorted(k for k in c["p"] if k.startswith("word_scorer1."))


# ------------------------------------------------------------- G5 / G6 --------
@pytest.mark.parametrize("ci", range(len(cases.G5_CASES)))
def test_g5_concat_att(ci):
    z, meta = load("g5_concat_att.npz")
    c = cases.g5_inputs(ci)
    left, right = T(c["left"], True), T(c["right"], True)
    w1, w2 = T(c["w1"], True), T(c["w2"], True)
    mask = T(c["mask"]) if c["mkind"] == "bool" else T(c["mask"].astype(np.float32))
    att, w = O.concat_att(left, right, mask, w1, w2)
    assert np.abs(att.detach().numpy() - z[f"c{ci}_att"]).max() <= 1e-5
    assert np.abs(w.detach().numpy() - z[f"c{ci}_w"]).max() <= 1e-6
    assert np.allclose(w.detach().sum(1).numpy(), 1.0, atol=1e-5)
    ((att * T(c["g_att"])).sum() + (w * T(c["g_w"])).sum()).backward()
    check_grad(z, f"c{ci}_dleft", left.grad.numpy())
    check_grad(z, f"c{ci}_dright", right.grad.numpy())
    check_grad(z, f"c{ci}_g::linear1.weight", w1.grad.numpy())
    check_grad(z, f"c{ci}_g::linear2.weight", w2.grad.numpy())


@pytest.mark.parametrize("ci", range(len(cases.G6_CASES)))
def test_g6_self_att_extend(ci):
    z, meta = load("g6_self_att.npz")
    c = cases.g6_inputs(ci)
    tsr, w1, w2 = T(c["tsr"], True), T(c["w1"], True), T(c["w2"], True)
    att, w = O.self_att_extend(tsr, T(c["mask"]), w1, w2)
    assert np.abs(att.detach().numpy() - z[f"c{ci}_att"]).max() <= 1e-5
    assert np.abs(w.detach().numpy() - z[f"c{ci}_w"]).max() <= 1e-6
    # backward of the left-less branch (self_attention.py:75-100)
    ((att * T(c["g_att"])).sum() + (w * T(c["g_w"])).sum()).backward()
    check_grad(z, f"c{ci}_dtsr", tsr.grad.numpy())
    check_grad(z, f"c{ci}_g::linear1.weight", w1.grad.numpy())
    check_grad(z, f"c{ci}_g::linear2.weight", w2.grad.numpy())


# ------------------------------------------------------------- G7 / G8 --------
def oracle_model_run(name, dtype=torch.float32):
    cfg, seed = MODEL_CASES[name]
    emb, art, clm = make_embeddings(cfg, seed)
    sd = make_state_dict(cfg, seed)
    p = {k: T(v, grad=True) for k, v in sd.items()}
    p["embedding.weight"] = T(emb)
    p["article_source_embs.weight"] = T(art, grad=True)
    if cfg.use_claim_source:
        p["claim_source_embs.weight"] = T(clm, grad=True)
    raw = make_raw_batch(cfg, seed)
    inp = assemble_inputs(raw, cfg, O.convert_text)
    phi, ww, ew, aux = O.model_forward(
        p, cfg.__dict__, T(inp["query"]), T(inp["document"]), T(inp["query_adj"]), T(inp["doc_ids"]),
        T(inp["doc_adj"]), T(inp["query_lens"]), inp["evd_counts"], T(inp["doc_sources"]),
        T(inp["query_sources"]), return_aux=True)
    loss = O.cross_entropy(phi, T(inp["labels"]))
    loss.backward()
    return cfg, p, phi, ww, ew, aux, loss


@pytest.mark.parametrize("name", list(MODEL_CASES))
def test_g7_full_model(name):
    z, meta = load(f"g7_model_{name}.npz")
    cfg, p, phi, ww, ew, aux, loss = oracle_model_run(name)
    assert np.abs(phi.detach().numpy() - z["phi"]).max() <= 1e-4            # north_star tolerance on logits
    assert np.abs(ww.detach().numpy() - z["word_w"]).max() <= 1e-5
    assert np.abs(ew.detach().numpy() - z["evd_w"]).max() <= 1e-5
    assert np.abs(aux["score"].detach().numpy() - z["score"]).max() <= 1e-5
    assert abs(loss.item() - float(z["loss"])) <= 1e-5
    none = set(meta["none_grads"])
    # dead parameters of the reference (LSTMs, trans) are not modelled by the oracle at all
    assert {k for k in none if k.startswith("ggnn_with_gsl.word_scorer1.")} == \
        {k for k in state_dict_shapes(cfg) if k.startswith("ggnn_with_gsl.word_scorer1.")}
    n_live = 0
    for k, v in p.items():
        if k == "embedding.weight":
            continue
        if k in none:
            assert v.grad is None or float(v.grad.abs().max()) == 0.0, k
            continue
        check_grad(z, f"g::{k}", v.grad.numpy(), what=name + " ")
        n_live += v.numel()
    assert n_live == meta["n_live"]


def test_g8_adam_step():
    z, meta = load("g7_model_small.npz")
    cfg, p, *_ = oracle_model_run("small")
    params = {k: v.detach() for k, v in p.items() if k != "embedding.weight"}
    grads = {k: v.grad for k, v in p.items() if k != "embedding.weight"}
    new = O.adam_step(params, grads, {}, lr=1e-4, weight_decay=1e-3)
    for k, v in new.items():
        if k in meta["none_grads"]:      # no gradient: torch.optim.Adam skips the parameter, no decay either (no fixture entry)
            assert f"adam::{k}" not in z.files
            assert np.array_equal(v.numpy(), params[k].numpy())
            continue
        exp = z[f"adam::{k}"]
        assert np.abs(v.numpy() - exp).max() <= 2e-6, k


def test_state_dict_contract_fixture_lists_dead_params(golden_dir):
    contract = json.load(open(os.path.join(golden_dir, "state_dict_contract_small.json")))
    assert "bilstm.rnn.weight_ih_l0" in contract and "trans.linear.weight" in contract
    cfg, _ = MODEL_CASES["small"]
    for k, shp in state_dict_shapes(cfg).items():
        assert tuple(contract[k]) == tuple(shp), k


def test_g9_reference_kargs_reproduce_what_the_fitter_hands_over():
    """G9: oracle/assemble.py `reference_kargs` -- the hand-written restatement of the fitter's
    de-padding and kargs protocol (char_man_fitter_query_repr1.py:204-250) that every parity test feeds the models with --
    against the kargs CAPTURED from the reference's own `_get_multiple_evidences_predictions_normal` (stub net, `small`
    case).  Every key forward() consumes must match in dtype, shape and value; the captured keys reference_kargs omits
    must be exactly the ones forward() ignores, and the drop-in's key table must cover all of them."""
    from get_amd.keywords import KeyWordSettings as K
    from oracle.assemble import reference_kargs
    z, meta = load("g9_fitter_kargs_small.npz")
    cfg, seed = MODEL_CASES[meta["case"]]
    inp = assemble_inputs(make_raw_batch(cfg, seed), cfg, O.convert_text)
    mine = reference_kargs(inp, torch)
    assert np.array_equal(z["query"], inp["query"]) and np.array_equal(z["document"], inp["document"])
    ignored = {"query_content_without_padding_evidences", "query_char_source", "doc_char_source"}
    assert set(meta["keys"]) - set(mine) == ignored
    assert set(mine) <= set(meta["keys"])

    def same(name, d, v):
        if d["kind"] == "scalar":
            assert v == d["value"], name
            return
        if d["kind"] == "tuple":
            assert isinstance(v, tuple) and len(v) == len(d["items"]), name
            for i, (di, vi) in enumerate(zip(d["items"], v)):
                if vi is not None:                      # (reference_kargs leaves the LSTM-legacy sort indices of the CLAIM side out)
                    same(f"{name}::{i}", di, vi)
            return
        a = v.numpy() if torch.is_tensor(v) else np.asarray(v)
        exp = z[name]
        assert str(a.dtype) == d["dtype"] and list(a.shape) == d["shape"], (name, a.dtype, a.shape, d)
        assert np.array_equal(a, exp), name

    for k, v in mine.items():
        same(f"k::{k}", meta["desc"][k], v)
    # only the claim side's sort indices are omitted (never read: graph_based_semantic_structure.py uses doc_lens_indices[2] alone)
    assert mine["query_lens_indices"][0] is None and all(x is not None for x in mine["doc_lens_indices"])
    # the drop-in's vocabulary holds every key string the fitter uses
    vocab = {v for k, v in vars(K).items() if isinstance(v, str) and not k.startswith("_")}
    assert set(meta["keys"]) - ignored <= vocab | {"fc_labels", "query_lens_indices"}, set(meta["keys"]) - vocab


# ------------------------------------------------------------- the pin checks itself
@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the reference only exists in the build container")
def test_golden_fixtures_regenerate_identically(tmp_path):
    """Re-run oracle/make_golden.py against /root/reference into a scratch directory: every array of every committed
    fixture must come back bit-identical (seeded inputs, deterministic reference) -- so a stale or hand-edited fixture,
    or a dumper that drifted from the committed vectors, fails here."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GET_GOLDEN_OUT=str(tmp_path))
    subprocess.check_call([sys.executable, os.path.join(root, "oracle", "make_golden.py")], env=env,
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    gold = os.path.join(root, "tests", "golden")
    names = sorted(f for f in os.listdir(gold) if f.endswith(".npz"))
    assert names == sorted(f for f in os.listdir(tmp_path) if f.endswith(".npz"))
    for f in names:
        a, b = np.load(os.path.join(gold, f)), np.load(os.path.join(tmp_path, f))
        assert sorted(a.files) == sorted(b.files), f
        for k in a.files:
            assert a[k].dtype == b[k].dtype and a[k].shape == b[k].shape and np.array_equal(a[k], b[k], equal_nan=True), (f, k)
    with open(os.path.join(gold, "state_dict_contract_small.json")) as fa, open(os.path.join(tmp_path, "state_dict_contract_small.json")) as fb:
        assert fa.read() == fb.read()
    # nothing unseeded in the fixtures: Adam results only for parameters that receive a gradient
    z, meta = load("g7_model_small.npz")
    assert not any(k.startswith("adam::") and k[6:] in meta["none_grads"] for k in z.files)
