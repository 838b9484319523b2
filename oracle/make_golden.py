"""TEST INFRASTRUCTURE ONLY -- golden-vector generator (runs ONLY in the build container).

Imports the upstream reference from ``/root/reference`` through
``oracle/_refshim.py`` and records inputs/outputs of the hot-path functions as
small ``.npz`` fixtures under ``tests/golden/`` (G1..G8 of SURVEY.md section
8(c)).  Inputs and parameters are regenerated from seeds by
``get_amd.synth`` so a fixture stores seeds + expected outputs, never weights
by the megabyte.  The reference never travels; the fixtures do.

    python oracle/make_golden.py          # rewrites tests/golden/*.npz
"""
import json
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import _refshim  # noqa: E402

_refshim.install()
warnings.filterwarnings("ignore")

import torch  # noqa: E402

import interactions as ref_inter  # noqa: E402
import Models.BiDAF.wrapper as ref_wrap  # noqa: E402
import thirdparty.self_attention as ref_sa  # noqa: E402
import thirdparty.two_branches_attention as ref_tba  # noqa: E402
from Models.FCWithEvidences import graph_based_semantic_structure as ref_model  # noqa: E402

from get_amd.synth import SynthConfig, make_embeddings, make_raw_batch, make_state_dict, make_tokens  # noqa: E402
from oracle import cases  # noqa: E402
from oracle.assemble import assemble_inputs, reference_kargs  # noqa: E402
from oracle.cases_model import MODEL_CASES  # noqa: E402

OUT = os.environ.get("GET_GOLDEN_OUT") or os.path.join(ROOT, "tests", "golden")
torch.set_num_threads(8)


def ref_convert_text(raw, fixed_length, length, window):
    return ref_inter.ClassificationInteractions.convert_text(None, list(raw), fixed_length, length, window)


def slices(g: torch.Tensor):
    """Summary of a big gradient: head rows, sum, abs-sum (full tensor if small)."""
    g = g.detach().double()
    flat = g.reshape(g.shape[0], -1) if g.dim() > 1 else g.reshape(1, -1)
    return dict(head=flat[:4, :64].numpy().copy(), sum=np.float64(g.sum()), abssum=np.float64(g.abs().sum()),
                sqsum=np.float64((g * g).sum()))


def put_summary(store, key, g):
    s = slices(g)
    for k, v in s.items():
        store[f"{key}::{k}"] = v


# ---------------------------------------------------------------- G1 ----------
def g1():
    rng = np.random.default_rng(101)
    store, meta = {}, []
    combos = [(30, 3), (100, 3), (100, 5), (200, 3), (30, 1), (100, 8)]
    idx = 0
    for fixed_length, window in combos:
        for rep in range(10):
            kind = rep % 5
            if kind == 0:      # heavy repeats from a tiny vocabulary
                length = int(rng.integers(1, fixed_length + 1))
                toks = rng.integers(1, 6, size=fixed_length)
            elif kind == 1:    # full length, zipf-ish
                toks, lens = make_tokens(rng, 1, fixed_length, 500, fixed_length, fixed_length)
                toks, length = toks[0], int(lens[0])
            elif kind == 2:    # length 1 / 2
                length = 1 + rep % 2
                toks = rng.integers(2, 50, size=fixed_length)
            elif kind == 3:    # all OOV
                length = int(rng.integers(2, fixed_length + 1))
                toks = np.ones(fixed_length, dtype=np.int64)
            else:              # all distinct
                length = int(rng.integers(2, fixed_length + 1))
                toks = rng.permutation(5000)[:fixed_length] + 2
            toks = np.asarray(toks, dtype=np.int64).copy()
            toks[length:] = 0
            words, adj, n = ref_convert_text([int(t) for t in toks], fixed_length, length, window)
            adj = np.asarray(adj)
            r, c = np.nonzero(adj)
            store[f"c{idx}_tokens"] = toks.astype(np.int32)
            store[f"c{idx}_words"] = np.asarray(words, dtype=np.int32)
            store[f"c{idx}_rows"] = r.astype(np.uint8)
            store[f"c{idx}_cols"] = c.astype(np.uint8)
            store[f"c{idx}_vals"] = adj[r, c].astype(np.float64)
            meta.append(dict(fixed_length=fixed_length, window=window, length=length, n_nodes=int(n)))
            idx += 1
    # the two worked examples of SURVEY 8(c)
    for toks, fl, ln, w in (([5, 7, 5, 9, 0, 0], 6, 4, 2), ([5, 7, 5, 9, 8, 0], 6, 5, 3)):
        words, adj, n = ref_convert_text(toks, fl, ln, w)
        adj = np.asarray(adj)
        r, c = np.nonzero(adj)
        store[f"c{idx}_tokens"] = np.asarray(toks, np.int32)
        store[f"c{idx}_words"] = np.asarray(words, np.int32)
        store[f"c{idx}_rows"], store[f"c{idx}_cols"] = r.astype(np.uint8), c.astype(np.uint8)
        store[f"c{idx}_vals"] = adj[r, c]
        meta.append(dict(fixed_length=fl, window=w, length=ln, n_nodes=int(n)))
        idx += 1
    store["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, "g1_convert_text.npz"), **store)
    print("G1", idx, "cases")


# ------------------------------------------------------------ helpers ---------
def load_cell(mod, p, prefix=""):
    sd = {k[len(prefix):]: torch.from_numpy(v) for k, v in p.items() if k.startswith(prefix)}
    mod.load_state_dict(sd, strict=True)


# ---------------------------------------------------------------- G2 ----------
def g2():
    store, meta = {}, []
    for ci, (n, r, din, dout, window) in enumerate(cases.G2_CASES):
        c = cases.g2_inputs(ci, ref_convert_text)
        toks, lens, adj, x, gw, p = c["toks"], c["lens"], c["adj"], c["x"], c["gw"], c["p"]
        mod = ref_wrap.GGNN(din, dout, dropout=0.2)
        load_cell(mod, p)
        mod.train(False)
        xt = torch.from_numpy(x).requires_grad_(True)
        out = mod(torch.from_numpy(adj).float(), xt)
        (out * torch.from_numpy(gw)).sum().backward()
        store[f"c{ci}_tokens"], store[f"c{ci}_lens"] = toks, lens
        store[f"c{ci}_out"] = out.detach().numpy()
        small = din * dout <= 64 * 64
        if small:
            store[f"c{ci}_dx"] = xt.grad.numpy()
        else:
            put_summary(store, f"c{ci}_dx", xt.grad)
        for name, prm in mod.named_parameters():
            if small or prm.numel() <= 512:
                store[f"c{ci}_g::{name}"] = prm.grad.numpy()
            else:
                put_summary(store, f"c{ci}_g::{name}", prm.grad)
        if ci == 0:
            # training mode (wrapper.py:189-190: nn.Dropout on the cell INPUT): the mask the reference drew is captured from its
            # own dropout module, so that the oracle's mask-replay form (ggnn_cell keep=) is pinned without relying on RNG streams
            mod.train(True)
            mod.zero_grad()
            cap = {}
            hk = mod.dropout.register_forward_hook(lambda m, i, o: cap.update(keep=(o != 0) | (i[0] == 0)))
            torch.manual_seed(4242 + ci)
            xt2 = torch.from_numpy(x).requires_grad_(True)
            out_t = mod(torch.from_numpy(adj).float(), xt2)
            hk.remove()
            (out_t * torch.from_numpy(gw)).sum().backward()
            store[f"c{ci}_train_keep"] = np.packbits(cap["keep"].numpy().reshape(-1))
            store[f"c{ci}_train_out"] = out_t.detach().numpy()
            store[f"c{ci}_train_dx"] = xt2.grad.numpy()
            store[f"c{ci}_train_g::proj.linear.weight"] = mod.proj.linear.weight.grad.numpy()
            mod.train(False)
        meta.append(dict(n=n, r=r, din=din, dout=dout, window=window, seed=200 + ci, small=bool(small), drop_p=0.2))
    store["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, "g2_ggnn.npz"), **store)
    print("G2 ok")


# ---------------------------------------------------------------- G3 ----------
def g3():
    store, meta = {}, []
    ci = 0
    for r in (30, 100, 200):
        for rate in (0.6, 0.8, 0.5):
            for ties in (False, True):
                b = 3
                score = cases.g3_scores(ci, r, ties, b)      # ties: many exact ties + a padded-node-like block
                adj = np.ones((b, r, r), np.float32)
                out = ref_wrap.GSL(rate)(torch.from_numpy(adj), torch.from_numpy(score)).numpy()
                store[f"c{ci}_score"] = score
                store[f"c{ci}_mask"] = np.packbits(out != 0, axis=-1)
                meta.append(dict(r=r, rate=rate, ties=ties, b=b, k=int(rate * r)))
                ci += 1
    # known answer of SURVEY 8(c): all-ones 4x4, scores [.9,.1,.8,.2], rate .5
    adj = torch.ones(1, 4, 4)
    sc = torch.tensor([[[.9], [.1], [.8], [.2]]])
    out = ref_wrap.GSL(0.5)(adj, sc).numpy()
    store["known_out"] = out
    store["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, "g3_gsl.npz"), **store)
    print("G3", ci, "cases")


# ---------------------------------------------------------------- G4 ----------
def g4():
    store, meta = {}, []
    for ci, (n, r, d, h, window, rate) in enumerate(cases.G4_CASES):
        c = cases.g4_inputs(ci, ref_convert_text)
        toks, lens, adj, x, gw, p = c["toks"], c["lens"], c["adj"], c["x"], c["gw"], c["p"]
        mod = ref_wrap.GGNN_with_GSL(d, h, h, rate=rate, dropout=0.2)
        mod.load_state_dict({k: torch.from_numpy(v) for k, v in p.items()}, strict=True)
        mod.train(False)
        adj_t = torch.from_numpy(adj).float()
        xt = torch.from_numpy(x).requires_grad_(True)
        # intermediate observables, by re-running the sub-modules exactly as forward() chains them
        with torch.no_grad():
            f1 = mod.feat_prop1(adj_t, xt)
            score = mod.word_scorer1(adj_t, f1)
            adj_r = mod.gsl1(adj_t, score)
        out = mod(adj_t, xt)
        (out * torch.from_numpy(gw)).sum().backward()
        store[f"c{ci}_tokens"], store[f"c{ci}_lens"] = toks, lens
        store[f"c{ci}_score"] = score.squeeze(-1).numpy()
        store[f"c{ci}_adjr_nz"] = np.packbits(adj_r.numpy() != 0, axis=-1)
        store[f"c{ci}_out"] = out.detach().numpy()
        small = d * h <= 64 * 64
        if small:
            store[f"c{ci}_dx"] = xt.grad.numpy()
        else:
            put_summary(store, f"c{ci}_dx", xt.grad)
        none_grads = []
        for name, prm in mod.named_parameters():
            if prm.grad is None:
                none_grads.append(name)
            elif small or prm.numel() <= 512:
                store[f"c{ci}_g::{name}"] = prm.grad.numpy()
            else:
                put_summary(store, f"c{ci}_g::{name}", prm.grad)
        meta.append(dict(n=n, r=r, d=d, h=h, window=window, rate=rate, seed=400 + ci, small=bool(small),
                         none_grads=none_grads))
    store["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, "g4_ggnn_gsl.npz"), **store)
    print("G4 ok")


# ------------------------------------------------------------- G5 / G6 --------
def g5_g6():
    store, meta = {}, []
    for ci, (b, l, xl, dr, ha, heads, mkind) in enumerate(cases.G5_CASES):
        c = cases.g5_inputs(ci)
        left, right, valid, mask, w1, w2, g_att, g_w = (c[k] for k in
                                                       ("left", "right", "valid", "mask", "w1", "w2", "g_att", "g_w"))
        mod = ref_tba.ConcatNotEqualSelfAtt(inp_dim=xl + dr, out_dim=ha, num_heads=heads)
        mod.load_state_dict({"linear1.weight": torch.from_numpy(w1), "linear2.weight": torch.from_numpy(w2)})
        lt = torch.from_numpy(left).requires_grad_(True)
        rt = torch.from_numpy(right).requires_grad_(True)
        mt = torch.from_numpy(mask) if mkind == "bool" else torch.from_numpy(mask.astype(np.float32))
        att, w = mod(lt, rt, mt)
        ((att * torch.from_numpy(g_att)).sum() + (w * torch.from_numpy(g_w)).sum()).backward()
        store[f"c{ci}_valid"] = valid
        store[f"c{ci}_att"], store[f"c{ci}_w"] = att.detach().numpy(), w.detach().numpy()
        store[f"c{ci}_dleft"] = lt.grad.numpy()
        if right.size <= 40000:
            store[f"c{ci}_dright"] = rt.grad.numpy()
        else:
            put_summary(store, f"c{ci}_dright", rt.grad)
        for name, prm in mod.named_parameters():
            if prm.numel() <= 4096:
                store[f"c{ci}_g::{name}"] = prm.grad.numpy()
            else:
                put_summary(store, f"c{ci}_g::{name}", prm.grad)
        meta.append(dict(b=b, l=l, xl=xl, dr=dr, ha=ha, heads=heads, mask=mkind, seed=500 + ci))
    store["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, "g5_concat_att.npz"), **store)

    store, meta = {}, []
    for ci, (b, l, d, ha, heads) in enumerate(cases.G6_CASES):
        c = cases.g6_inputs(ci)
        tsr, valid, mask, w1, w2 = c["tsr"], c["valid"], c["mask"], c["w1"], c["w2"]
        mod = ref_sa.MultiHeadSelfAttentionICLR2017Extend(inp_dim=d, out_dim=ha, num_heads=heads)
        mod.load_state_dict({"linear1.weight": torch.from_numpy(w1), "linear2.weight": torch.from_numpy(w2)})
        tt = torch.from_numpy(tsr).requires_grad_(True)
        att, w = mod(tt, torch.from_numpy(mask), return_att_weights=True)
        # backward of the `left is None` branch (self_attention.py:75-100): d tsr and both weight gradients
        ((att * torch.from_numpy(c["g_att"])).sum() + (w * torch.from_numpy(c["g_w"])).sum()).backward()
        store[f"c{ci}_valid"] = valid
        store[f"c{ci}_att"], store[f"c{ci}_w"] = att.detach().numpy(), w.detach().numpy()
        if tsr.size <= 40000:
            store[f"c{ci}_dtsr"] = tt.grad.numpy()
        else:
            put_summary(store, f"c{ci}_dtsr", tt.grad)
        for pname, prm in mod.named_parameters():
            if prm.numel() <= 4096:
                store[f"c{ci}_g::{pname}"] = prm.grad.numpy()
            else:
                put_summary(store, f"c{ci}_g::{pname}", prm.grad)
        meta.append(dict(b=b, l=l, d=d, ha=ha, heads=heads, seed=600 + ci))
    store["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, "g6_self_att.npz"), **store)
    print("G5/G6 ok")


# ------------------------------------------------------------- G7 / G8 --------
def build_ref_model(cfg, seed):
    emb, art, clm = make_embeddings(cfg, seed)
    params = cfg.model_params(emb, art, clm)
    model = ref_model.Graph_basedSemantiStructure(params)
    sd = make_state_dict(cfg, seed)
    live = {k: torch.from_numpy(v) for k, v in sd.items()}
    full = model.state_dict()
    for k, v in live.items():
        assert full[k].shape == v.shape, (k, full[k].shape, v.shape)
        full[k] = v
    model.load_state_dict(full, strict=True)
    model.train(False)
    return model


def g7_g8():
    keys_written = False
    for name, (cfg, seed) in MODEL_CASES.items():
        store = {}
        model = build_ref_model(cfg, seed)
        if not keys_written:
            # state_dict contract (names, shapes) incl. the dead LSTM/trans parameters -- SURVEY 8(b)
            contract = {k: list(v.shape) for k, v in model.state_dict().items()}
            with open(os.path.join(OUT, "state_dict_contract_small.json"), "w") as f:
                json.dump(contract, f, indent=0, sort_keys=True)
            keys_written = True
        raw = make_raw_batch(cfg, seed)
        inp = assemble_inputs(raw, cfg, ref_convert_text)
        kargs = reference_kargs(inp, torch, output_ranking=True)
        phi, (ww, ew) = model(torch.from_numpy(inp["query"]), torch.from_numpy(inp["document"]), **kargs)
        loss = torch.nn.CrossEntropyLoss()(phi, torch.from_numpy(inp["labels"]).float().long())
        loss.backward()
        store["phi"], store["word_w"], store["evd_w"] = phi.detach().numpy(), ww.detach().numpy(), ew.detach().numpy()
        store["loss"] = np.float64(loss.item())
        # GSL keep sets as observed through the refined adjacency of the evidence graphs
        with torch.no_grad():
            g = model.ggnn_with_gsl
            adj_t = torch.from_numpy(inp["doc_adj"]).float()
            f1 = g.feat_prop1(adj_t, model.embedding(torch.from_numpy(inp["doc_ids"])))
            score = g.word_scorer1(adj_t, f1).squeeze(-1)
        store["score"] = score.numpy()
        none_grads = []
        small = cfg.hidden <= 64
        for pname, prm in model.named_parameters():
            if prm.grad is None:
                none_grads.append(pname)
            elif small or prm.numel() <= 2048:
                store[f"g::{pname}"] = prm.grad.numpy()
            else:
                put_summary(store, f"g::{pname}", prm.grad)
        meta = dict(cfg=cfg.__dict__, seed=seed, none_grads=none_grads, small=small,
                    n_live=int(sum(p.numel() for p in model.parameters() if p.grad is not None)))
        if small:
            # G8: one Adam(lr=1e-4, weight_decay=1e-3) step (declare_fitter.py:58-61)
            opt = torch.optim.Adam(model.parameters(), lr=1e-4, weight_decay=1e-3)
            opt.step()
            for pname, prm in model.named_parameters():
                if prm.grad is not None:      # Adam skips grad-None parameters; the dead ones are unseeded noise
                    store[f"adam::{pname}"] = prm.detach().numpy()
        store["meta"] = np.frombuffer(json.dumps(meta, default=str).encode(), dtype=np.uint8)
        np.savez_compressed(os.path.join(OUT, f"g7_model_{name}.npz"), **store)
        print("G7", name, "loss", loss.item(), "none grads", len(none_grads), "live", meta["n_live"])


def g9():
    """G9: the **kargs the reference FITTER itself hands to net(...) -- captured, not restated.  The reference's own
    `_get_multiple_evidences_predictions_normal` (Fitting/FittingFC/char_man_fitter_query_repr1.py:164-258, called here as
    an unbound function on a stub `self`) runs its de-padding loop (:204-223) on the padded tensors its training loop
    builds (:92-112) for the `small` case; the stub `net` records what arrives.  Stored: every key, and for tensor / array
    values dtype, shape and values (tuples element-wise).  tests/test_oracle_golden.py asserts that oracle/assemble.py's
    hand-written `reference_kargs` reproduces it."""
    import types
    from Fitting.FittingFC import char_man_fitter_query_repr1 as F
    from setting_keywords import KeyWordSettings as KW
    cfg, seed = MODEL_CASES["small"]
    raw = make_raw_batch(cfg, seed)
    inp = assemble_inputs(raw, cfg, ref_convert_text)
    B, n, L, R = cfg.batch, cfg.fixed_num_evidences, cfg.len_left, cfg.len_right
    counts = inp["evd_counts"]
    # padded per-claim tensors, as handlers/mz_sampler.py:115-176 materialises them
    adj = np.zeros((B, n, R, R), np.float64)
    last = 0
    for b in range(B):
        c = int(counts[b])
        adj[b, :c] = inp["doc_adj"][last:last + c]
        last += c
    seen = {}

    def net(query, document, **kargs):
        seen["query"], seen["document"], seen["kargs"] = query, document, kargs
        return torch.zeros(query.size(0), 2)

    stub = types.SimpleNamespace(_net=net, _use_cuda=False, _loss_func=lambda pred, labels: pred.sum() * 0.0)
    t = torch.from_numpy
    extra = {KW.EvidenceCountPerQuery: t(counts), KW.FCClass.QueryCharSource: t(np.zeros((B, 1, L), np.int64)),
             KW.FCClass.DocCharSource: t(np.zeros((B, n, R), np.int64)), KW.Query_Adj: t(inp["query_adj"]),
             KW.Evd_Docs_Adj: t(adj)}
    F.CharManFitterQueryRepr1._get_multiple_evidences_predictions_normal(
        stub, t(np.arange(B, dtype=np.int64)), t(inp["query"]), inp["query_lens"], t(inp["query_sources"]),
        t(np.zeros((B, n), np.int64)), t(inp["document"]), inp["docs_lens"], t(inp["doc_sources"]), t(inp["labels"]), n, **extra)
    store, keys = {}, []

    def put(name, v):
        if torch.is_tensor(v):
            v = v.detach().numpy()
        if isinstance(v, np.ndarray):
            store[name] = v
            return {"kind": "array", "dtype": str(v.dtype), "shape": list(v.shape)}
        if isinstance(v, (tuple, list)):
            return {"kind": "tuple", "items": [put(f"{name}::{i}", x) for i, x in enumerate(v)]}
        return {"kind": "scalar", "value": v}

    desc = {}
    for k, v in seen["kargs"].items():
        keys.append(k)
        desc[k] = put(f"k::{k}", v)
    store["query"], store["document"] = seen["query"].numpy(), seen["document"].numpy()
    meta = dict(case="small", seed=seed, keys=keys, desc=desc,
                source="Fitting/FittingFC/char_man_fitter_query_repr1.py:164-258 run on a stub self; net(**kargs) recorded")
    store["meta"] = np.frombuffer(json.dumps(meta, default=str).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, "g9_fitter_kargs_small.npz"), **store)
    print("G9 fitter kargs:", keys)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    g1(); g2(); g3(); g4(); g5_g6(); g7_g8(); g9()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))
