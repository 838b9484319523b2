"""GPU parity tests of the individual hot-path ops: HIP (through the C-ABI) vs the CPU oracle and
vs the golden vectors captured from the reference.  Bit-exact for ids/bits/keep-sets, fp32
tolerances (stated per test) for floating point.  Run with ``-m gpu`` on an MI355X."""
import numpy as np
import pytest
import torch

from oracle import cases
from oracle import get_oracle as O
from tests.util import check_grad, dense_from_coo, load

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def T(a, grad=False, dev=DEV):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    return t.requires_grad_(True) if grad else t


def bits_to_bool(bits: torch.Tensor, r: int) -> np.ndarray:
    """(..., W) int64 words -> (..., r) bool."""
    b = bits.cpu().numpy().view(np.uint64)
    out = np.zeros(b.shape[:-1] + (r,), dtype=bool)
    for j in range(r):
        out[..., j] = (b[..., j // 64] >> np.uint64(j % 64)) & np.uint64(1)
    return out


def maxerr(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max())


# ---------------------------------------------------------------- a1 graph build
def test_graph_build_matches_golden_bit_exact():
    from get_amd import ops
    z, meta = load("g1_convert_text.npz")
    by_shape = {}
    for i, m in enumerate(meta):
        by_shape.setdefault((m["fixed_length"], m["window"]), []).append(i)
    for (fl, win), idxs in by_shape.items():
        toks = np.stack([z[f"c{i}_tokens"] for i in idxs]).astype(np.int32)
        lens = np.array([meta[i]["length"] for i in idxs], np.int32)
        adj, node_ids, n_nodes = ops.graph_build(T(toks), T(lens), win)
        dense = adj.to_dense().cpu().numpy()
        nbits = bits_to_bool(adj.bits, fl)
        for k, i in enumerate(idxs):
            assert int(n_nodes[k]) == meta[i]["n_nodes"], (fl, win, i)
            assert np.array_equal(node_ids[k].cpu().numpy(), z[f"c{i}_words"]), (fl, win, i)
            exp = dense_from_coo(z, i, fl)
            assert np.array_equal(nbits[k], exp != 0), (fl, win, i)
            assert maxerr(dense[k], exp) <= 2e-7, (fl, win, i)          # fp32 product of two fp32 dinv


def test_graph_build_empty_and_full_edge_cases():
    from get_amd import ops
    toks = np.zeros((3, 100), np.int32)
    toks[1, :] = 7                      # one node, full length
    toks[2, :100] = np.arange(100) + 2  # all distinct
    lens = np.array([0, 100, 100], np.int32)
    adj, ids, n = ops.graph_build(T(toks), T(lens), 3)
    assert n.cpu().tolist() == [0, 1, 100]
    d = adj.to_dense().cpu().numpy()
    assert not d[0].any()
    assert d[1, 0, 0] == 1.0 and d[1].sum() == 1.0
    for k in range(3):
        _, exp, _ = O.convert_text(toks[k].tolist(), 100, int(lens[k]), 3)
        assert maxerr(d[k], exp) <= 2e-7


def test_adj_pack_unpack_roundtrip():
    from get_amd.ops import PackedAdj
    rng = np.random.default_rng(5)
    for r in (30, 100, 200):
        a = rng.standard_normal((3, r, r)) * (rng.random((3, r, r)) < 0.1)
        for dt in (np.float64, np.float32):
            p = PackedAdj.from_dense(T(a.astype(dt)))
            back = p.to_dense().cpu().numpy()
            assert np.array_equal(back, a.astype(dt).astype(np.float32))
            # bit pattern = union of A's and A^T's patterns (the transposed aggregation walks row i's bits for A[j][i])
            assert np.array_equal(bits_to_bool(p.bits, r), (a != 0) | (a.transpose(0, 2, 1) != 0))


# ---------------------------------------------------------------- aggregation
@pytest.mark.parametrize("r,h", [(100, 300), (30, 300), (200, 64), (100, 1), (100, 30), (100, 768)])
def test_spmm_modes(r, h):
    from get_amd import ops
    rng = np.random.default_rng(r * 1000 + h)
    toks, lens, ids, adj = cases.graphs(rng, 4, r, 3, O.convert_text)
    x = rng.standard_normal((4, r, h)).astype(np.float32)
    ref = torch.from_numpy(adj).float() @ torch.from_numpy(x)
    # normalised mode (native graph build)
    padj, _, _ = ops.graph_build(T(toks), T(lens), 3)
    y = ops.spmm(padj, T(x)).cpu()
    assert maxerr(y, ref) <= 2e-6 * max(1.0, float(ref.abs().max()))
    # weighted mode (dense hand-over), incl. a non-symmetric adjacency and its transpose in backward
    a2 = adj * (1.0 + 0.1 * rng.standard_normal(adj.shape))
    pw = ops.PackedAdj.from_dense(T(a2))
    xt = T(x, grad=True)
    y2 = ops.spmm(pw, xt)
    ref2 = torch.from_numpy(a2).float() @ torch.from_numpy(x)
    assert maxerr(y2.detach().cpu(), ref2) <= 2e-6 * max(1.0, float(ref2.abs().max()))
    g = rng.standard_normal((4, r, h)).astype(np.float32)
    (y2 * T(g)).sum().backward()
    refg = torch.from_numpy(a2).float().transpose(1, 2) @ torch.from_numpy(g)
    assert maxerr(xt.grad.cpu(), refg) <= 2e-6 * max(1.0, float(refg.abs().max()))
    # keep-set refinement == dense mask of wrapper.py:221-225
    keep = rng.random((4, r)) < 0.5
    kw = np.zeros((4, (r + 63) // 64), np.uint64)
    for j in range(r):
        kw[:, j // 64] |= keep[:, j].astype(np.uint64) << np.uint64(j % 64)
    pk = padj.with_keep(T(kw.view(np.int64)))
    y3 = ops.spmm(pk, T(x)).cpu()
    mask = (keep[:, :, None] | keep[:, None, :])
    ref3 = (torch.from_numpy(adj * mask).float()) @ torch.from_numpy(x)
    assert maxerr(y3, ref3) <= 2e-6 * max(1.0, float(ref3.abs().max()))
    assert maxerr(pk.to_dense().cpu(), (adj * mask).astype(np.float32)) <= 2e-7


@pytest.mark.parametrize("r", [30, 100, 130])
def test_spmm_pattern_asymmetric_adjacency(r):
    """A dense adjacency whose NON-ZERO PATTERN is not symmetric (A[j][i] != 0 while A[i][j] == 0, e.g. a directed or
    row-masked graph handed over through the reference API): forward, the transposed backward, the GGNN cell's
    gradients and the dense round trip must all see every entry."""
    from get_amd import modules, ops
    rng = np.random.default_rng(900 + r)
    n, h = 3, 32
    a = rng.standard_normal((n, r, r)) * (rng.random((n, r, r)) < 0.06)           # random directed pattern
    a[:, np.arange(r), np.arange(r)] = 1.0
    a[:, :, r - 1] = 0.0                                                             # a column with no entries at all ...
    a[:, r - 1, : r // 2] = 0.5                                                      # ... whose row has many
    assert ((a != 0) != (a.transpose(0, 2, 1) != 0)).any()
    x = rng.standard_normal((n, r, h)).astype(np.float32)
    g = rng.standard_normal((n, r, h)).astype(np.float32)
    pw = ops.PackedAdj.from_dense(T(a))
    assert maxerr(pw.to_dense().cpu(), a.astype(np.float32)) == 0.0
    xt = T(x, grad=True)
    y = ops.spmm(pw, xt)
    ref = torch.from_numpy(a).float() @ torch.from_numpy(x)
    assert maxerr(y.detach().cpu(), ref) <= 2e-6 * max(1.0, float(ref.abs().max()))
    (y * T(g)).sum().backward()
    refg = torch.from_numpy(a).float().transpose(1, 2) @ torch.from_numpy(g)
    assert maxerr(xt.grad.cpu(), refg) <= 2e-6 * max(1.0, float(refg.abs().max()))
    # the whole cell (the backward's `dxp += A^T da`) against the oracle
    p = cases.cell_params(rng, h, h)
    cell = modules.GGNN(h, h, dropout=0.0)
    _load_cell(cell, p)
    cell = cell.to(DEV)
    xc = T(x, grad=True)
    (cell(T(a), xc) * T(g)).sum().backward()
    po = {k: torch.from_numpy(v).requires_grad_(True) for k, v in p.items()}
    xo = torch.from_numpy(x).requires_grad_(True)
    (O.ggnn_cell(torch.from_numpy(a).float(), xo, po) * torch.from_numpy(g)).sum().backward()
    assert maxerr(xc.grad.cpu(), xo.grad) <= 1e-3 * float(xo.grad.abs().max()) + 1e-6
    for k, prm in cell.named_parameters():
        assert maxerr(prm.grad.cpu(), po[k].grad) <= 1e-3 * float(po[k].grad.abs().max()) + 1e-6, k


@pytest.mark.parametrize("r,h", [(100, 300), (64, 32), (200, 64)])
def test_spmm_dense_adjacency_beyond_the_edge_list(r, h):
    """A dense hand-over with far more edges than the aggregation kernel's LDS edge list holds (11 R): the kernel's
    in-place bit-walk fallback must give the same product, forward and transposed."""
    from get_amd import ops
    rng = np.random.default_rng(77 + r)
    n = 3
    a = rng.standard_normal((n, r, r)) * (rng.random((n, r, r)) < 0.4)
    a[1] = np.eye(r)                                   # a sparse graph in the same launch takes the list path
    x = rng.standard_normal((n, r, h)).astype(np.float32)
    g = rng.standard_normal((n, r, h)).astype(np.float32)
    xt = T(x, grad=True)
    y = ops.spmm(ops.PackedAdj.from_dense(T(a)), xt)
    ref = torch.from_numpy(a).float() @ torch.from_numpy(x)
    assert maxerr(y.detach().cpu(), ref) <= 4e-6 * max(1.0, float(ref.abs().max()))
    (y * T(g)).sum().backward()
    refg = torch.from_numpy(a).float().transpose(1, 2) @ torch.from_numpy(g)
    assert maxerr(xt.grad.cpu(), refg) <= 4e-6 * max(1.0, float(refg.abs().max()))


# ---------------------------------------------------------------- a2 GGNN cell (G2)
def _load_cell(mod, p, prefix=""):
    sd = {k[len(prefix):]: torch.from_numpy(v) for k, v in p.items() if k.startswith(prefix)}
    mod.load_state_dict(sd, strict=True)


@pytest.mark.parametrize("ci", range(len(cases.G2_CASES)))
@pytest.mark.parametrize("mode", ["dense", "native"])
def test_ggnn_cell_vs_golden_and_oracle(ci, mode, arith_mode):
    from get_amd import modules, ops
    z, meta = load("g2_ggnn.npz")
    n, r, din, dout, window = cases.G2_CASES[ci]
    c = cases.g2_inputs(ci, O.convert_text)
    mod = modules.GGNN(din, dout, dropout=0.2)
    _load_cell(mod, c["p"])
    mod = mod.to(DEV).train(False)
    x = T(c["x"], grad=True)
    if mode == "dense":
        adj = T(c["adj"])                     # float64 dense, as the reference pipeline hands it over
    else:
        adj, _, _ = ops.graph_build(T(c["toks"]), T(c["lens"]), window)
    out = mod(adj, x)
    assert out.shape == (n, r, dout)
    assert maxerr(out.detach().cpu(), z[f"c{ci}_out"]) <= 2e-5
    (out * T(c["gw"])).sum().backward()
    check_grad(z, f"c{ci}_dx", x.grad.cpu().numpy(), what=f"{mode} ")
    for name, prm in mod.named_parameters():
        check_grad(z, f"c{ci}_g::{name}", prm.grad.cpu().numpy(), what=f"{mode} ")
    # full-tensor comparison against the oracle's gradients as well
    p = {k: torch.from_numpy(v).requires_grad_(True) for k, v in c["p"].items()}
    xo = torch.from_numpy(c["x"]).requires_grad_(True)
    oo = O.ggnn_cell(torch.from_numpy(c["adj"]).float(), xo, p)
    (oo * torch.from_numpy(c["gw"])).sum().backward()
    assert maxerr(x.grad.cpu(), xo.grad) <= 1e-3 * float(xo.grad.abs().max()) + 1e-6
    for name, prm in mod.named_parameters():
        go = p[name].grad
        assert maxerr(prm.grad.cpu(), go) <= 1e-3 * float(go.abs().max()) + 1e-5, name


def test_ggnn_cell_fused_embedding_gather():
    from get_amd import modules, ops
    rng = np.random.default_rng(77)
    n, r, d, h = 5, 30, 48, 64
    toks, lens, ids, adj = cases.graphs(rng, n, r, 3, O.convert_text, vocab=50)
    emb = rng.standard_normal((50, d)).astype(np.float32)
    p = cases.cell_params(rng, d, h)
    mod = modules.GGNN(d, h)
    _load_cell(mod, p)
    mod = mod.to(DEV).train(False)
    e = torch.nn.Embedding.from_pretrained(torch.from_numpy(emb), freeze=False).to(DEV)
    padj, node_ids, _ = ops.graph_build(T(toks), T(lens), 3)
    assert np.array_equal(node_ids.cpu().numpy(), ids)
    out = mod.forward_ids(padj, e, node_ids)
    po = {k: torch.from_numpy(v) for k, v in p.items()}
    embo = torch.from_numpy(emb).requires_grad_(True)
    oo = O.ggnn_cell(torch.from_numpy(adj).float(), embo[torch.from_numpy(ids)], po)
    assert maxerr(out.detach().cpu(), oo.detach()) <= 2e-5
    gw = torch.from_numpy(rng.standard_normal((n, r, h)).astype(np.float32))
    (out * gw.to(DEV)).sum().backward()
    (oo * gw).sum().backward()
    assert maxerr(e.weight.grad.cpu(), embo.grad) <= 1e-3 * float(embo.grad.abs().max()) + 1e-6


# ---------------------------------------------------------------- a3 GSL (G3)
def test_gsl_keep_sets_golden():
    from get_amd import modules
    z, meta = load("g3_gsl.npz")
    for ci, m in enumerate(meta):
        r = m["r"]
        score = T(z[f"c{ci}_score"])
        exp_mask = np.unpackbits(z[f"c{ci}_mask"], axis=-1)[..., :r].astype(bool)
        out = modules.GSL(m["rate"])(torch.ones(m["b"], r, r, device=DEV), score).cpu().numpy()
        got_mask = out != 0
        kept = got_mask.all(-1)
        assert kept.sum(-1).tolist() == [m["k"]] * m["b"], ci
        # identical to the oracle's convention (ties -> lower index), bit for bit
        _, keep_o = O.gsl_refine(torch.ones(m["b"], r, r), torch.from_numpy(z[f"c{ci}_score"]), m["rate"])
        assert np.array_equal(kept, keep_o.numpy()), ci
        if not m["ties"]:
            assert np.array_equal(got_mask, exp_mask), ci            # and to the reference where it is defined
    out = modules.GSL(0.5)(torch.ones(1, 4, 4, device=DEV), torch.tensor([[[.9], [.1], [.8], [.2]]], device=DEV))
    assert np.array_equal(out.cpu().numpy(), z["known_out"])


# ---------------------------------------------------------------- a4 GGNN_with_GSL (G4)
@pytest.mark.parametrize("ci", range(len(cases.G4_CASES)))
def test_ggnn_with_gsl_vs_golden(ci, arith_mode):
    from get_amd import modules
    z, meta = load("g4_ggnn_gsl.npz")
    m = meta[ci]
    n, r, d, h, window, rate = cases.G4_CASES[ci]
    c = cases.g4_inputs(ci, O.convert_text)
    mod = modules.GGNN_with_GSL(d, h, h, rate=rate, dropout=0.2)
    mod.load_state_dict({k: torch.from_numpy(v) for k, v in c["p"].items()}, strict=True)
    mod = mod.to(DEV).train(False)
    x = T(c["x"], grad=True)
    out = mod(T(c["adj"]), x)
    assert maxerr(mod.last_score.cpu(), z[f"c{ci}_score"]) <= 1e-5
    # keep-set: must reproduce the reference's refined adjacency pattern on real (non-padding) nodes
    exp_nz = np.unpackbits(z[f"c{ci}_adjr_nz"], axis=-1)[..., :r].astype(bool)
    keep = bits_to_bool(mod.last_keep, r)
    nz = (c["adj"] != 0) & (keep[:, :, None] | keep[:, None, :])
    assert np.array_equal(nz, exp_nz)
    assert maxerr(out.detach().cpu(), z[f"c{ci}_out"]) <= 2e-5
    (out * T(c["gw"])).sum().backward()
    check_grad(z, f"c{ci}_dx", x.grad.cpu().numpy())
    for name, prm in mod.named_parameters():
        if name in m["none_grads"]:
            assert prm.grad is None, name                          # no gradient through GSL (wrapper.py:219)
        else:
            check_grad(z, f"c{ci}_g::{name}", prm.grad.cpu().numpy())


# ---------------------------------------------------------------- a5/a6 attention (G5, G6)
@pytest.mark.parametrize("ci", range(len(cases.G5_CASES)))
def test_concat_att_vs_golden(ci, arith_mode):
    from get_amd import modules
    z, meta = load("g5_concat_att.npz")
    b, l, xl, dr, ha, heads, mkind = cases.G5_CASES[ci]
    c = cases.g5_inputs(ci)
    mod = modules.ConcatNotEqualSelfAtt(inp_dim=xl + dr, out_dim=ha, num_heads=heads)
    mod.load_state_dict({"linear1.weight": torch.from_numpy(c["w1"]), "linear2.weight": torch.from_numpy(c["w2"])})
    mod = mod.to(DEV)
    left, right = T(c["left"], True), T(c["right"], True)
    mask = T(c["mask"]) if mkind == "bool" else T(c["mask"].astype(np.float32))
    att, w = mod(left, right, mask)
    assert att.shape == (b, dr, heads) and w.shape == (b, l, heads)
    assert maxerr(att.detach().cpu(), z[f"c{ci}_att"]) <= 1e-5
    assert maxerr(w.detach().cpu(), z[f"c{ci}_w"]) <= 1e-6
    assert np.allclose(w.detach().sum(1).cpu().numpy(), 1.0, atol=1e-5)      # fitter's own self-check (:433,448)
    ((att * T(c["g_att"])).sum() + (w * T(c["g_w"])).sum()).backward()
    check_grad(z, f"c{ci}_dleft", left.grad.cpu().numpy())
    check_grad(z, f"c{ci}_dright", right.grad.cpu().numpy())
    check_grad(z, f"c{ci}_g::linear1.weight", mod.linear1.weight.grad.cpu().numpy())
    check_grad(z, f"c{ci}_g::linear2.weight", mod.linear2.weight.grad.cpu().numpy())


@pytest.mark.parametrize("ci", range(len(cases.G6_CASES)))
def test_self_att_extend_vs_golden(ci):
    from get_amd import modules
    z, meta = load("g6_self_att.npz")
    b, l, d, ha, heads = cases.G6_CASES[ci]
    c = cases.g6_inputs(ci)
    mod = modules.MultiHeadSelfAttentionICLR2017Extend(inp_dim=d, out_dim=ha, num_heads=heads)
    mod.load_state_dict({"linear1.weight": torch.from_numpy(c["w1"]), "linear2.weight": torch.from_numpy(c["w2"])})
    mod = mod.to(DEV)
    tsr = T(c["tsr"], True)
    att, w = mod(tsr, T(c["mask"]), return_att_weights=True)
    assert att.shape == (b, heads, d)
    assert maxerr(att.detach().cpu(), z[f"c{ci}_att"]) <= 1e-5
    assert maxerr(w.detach().cpu(), z[f"c{ci}_w"]) <= 1e-6
    # backward of the `left = NULL` branch of gh_concat_att_bwd against the reference's gradients (self_attention.py:75-100)
    ((att * T(c["g_att"])).sum() + (w * T(c["g_w"])).sum()).backward()
    check_grad(z, f"c{ci}_dtsr", tsr.grad.cpu().numpy())
    check_grad(z, f"c{ci}_g::linear1.weight", mod.linear1.weight.grad.cpu().numpy())
    check_grad(z, f"c{ci}_g::linear2.weight", mod.linear2.weight.grad.cpu().numpy())


# ---------------------------------------------------------------- linear / ragged helpers / Adam
# (32, 3556, 300): the head -- 55 K chunks, workgroup-per-row finish; (32, 4096, 768): the same finish over three 64-float4 column
# passes; (40, 300, 3556): 12 column-block problems in one grouped launch (the head's backward); (3, 2000, 300): one row tile,
# linear work decode, rows that are not a multiple of the finish kernel's four
@pytest.mark.parametrize("m,k,n", [(32, 3556, 300), (32, 300, 2), (960, 300, 300), (7, 20, 5), (2048, 64, 48),
                                   (32, 4096, 768), (40, 300, 3556), (3, 2000, 300)])
def test_linear_fwd_bwd(m, k, n):
    from get_amd import ops
    rng = np.random.default_rng(m + k + n)
    x = rng.standard_normal((m, k)).astype(np.float32)
    w = (rng.standard_normal((n, k)) / np.sqrt(k)).astype(np.float32)
    b = rng.standard_normal((n,)).astype(np.float32)
    g = rng.standard_normal((m, n)).astype(np.float32)
    xt, wt, bt = T(x, True), T(w, True), T(b, True)
    y = ops.linear(xt, wt, bt)
    (y * T(g)).sum().backward()
    xo, wo, bo = (torch.from_numpy(a).double().requires_grad_(True) for a in (x, w, b))
    yo = xo @ wo.t() + bo
    (yo * torch.from_numpy(g).double()).sum().backward()
    assert maxerr(y.detach().cpu(), yo.detach()) <= 2e-5 * max(1.0, float(yo.abs().max()))
    for got, exp in ((xt.grad, xo.grad), (wt.grad, wo.grad), (bt.grad, bo.grad)):
        assert maxerr(got.cpu(), exp) <= 1e-4 * float(exp.abs().max()) + 1e-6


def test_segment_helpers_and_masked_mean():
    from get_amd import ops
    rng = np.random.default_rng(3)
    counts = [1, 30, 7, 12, 3]
    b1, x = sum(counts), 44
    seg = ops.Segments(torch.tensor(counts, device=DEV), b1, 30)
    assert seg.offsets.cpu().tolist() == np.concatenate([[0], np.cumsum(counts)]).tolist()
    src = rng.standard_normal((len(counts), x)).astype(np.float32)
    st = T(src, True)
    out = ops.seg_broadcast(st, seg)
    ref = O.pad_left(torch.from_numpy(src), counts)
    assert np.array_equal(out.detach().cpu().numpy(), ref.numpy())
    g = rng.standard_normal((b1, x)).astype(np.float32)
    (out * T(g)).sum().backward()
    so = torch.from_numpy(src).requires_grad_(True)
    (O.pad_left(so, counts) * torch.from_numpy(g)).sum().backward()
    assert maxerr(st.grad.cpu(), so.grad) <= 1e-5
    pairs = rng.standard_normal((b1, x)).astype(np.float32)
    pt = T(pairs, True)
    padded = ops.seg_pad(pt, seg)
    refp = O.pad_right(torch.from_numpy(pairs), counts, 30)
    assert np.array_equal(padded.detach().cpu().numpy(), refp.numpy())
    g2 = rng.standard_normal((len(counts), 30, x)).astype(np.float32)
    (padded * T(g2)).sum().backward()
    po = torch.from_numpy(pairs).requires_grad_(True)
    (O.pad_right(po, counts, 30) * torch.from_numpy(g2)).sum().backward()
    assert np.array_equal(pt.grad.cpu().numpy(), po.grad.numpy())
    # masked mean
    hid = rng.standard_normal((4, 30, 52)).astype(np.float32)
    ids = (rng.random((4, 30)) < 0.6).astype(np.int64) * 5
    lens = np.maximum(ids.astype(bool).sum(1), 1)
    ht = T(hid, True)
    mm = ops.masked_mean(ht, T(ids), T(lens))
    ho = torch.from_numpy(hid).requires_grad_(True)
    mo = (ho * torch.from_numpy(ids > 0).float()[:, :, None]).sum(1) / torch.from_numpy(lens).float()[:, None]
    assert maxerr(mm.detach().cpu(), mo.detach()) <= 1e-5
    gm = rng.standard_normal((4, 52)).astype(np.float32)
    (mm * T(gm)).sum().backward()
    (mo * torch.from_numpy(gm)).sum().backward()
    assert maxerr(ht.grad.cpu(), ho.grad) <= 1e-6


def test_adam_flat_matches_torch():
    from get_amd import ops
    rng = np.random.default_rng(9)
    n = 10007
    p0 = rng.standard_normal(n).astype(np.float32)
    ref = torch.nn.Parameter(torch.from_numpy(p0.copy()))
    opt = torch.optim.Adam([ref], lr=1e-4, weight_decay=1e-3)
    p, m, v = T(p0.copy()), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    for step in range(1, 4):
        g = rng.standard_normal(n).astype(np.float32)
        ref.grad = torch.from_numpy(g.copy())
        opt.step()
        ops.adam_step_flat(p, T(g), m, v, step, lr=1e-4, weight_decay=1e-3)
        assert maxerr(p.cpu(), ref.detach()) <= 2e-7, step


# ---------------------------------------------------------------- fused (in-kernel) dropout
@pytest.mark.parametrize("n", [6, 90])
def test_fused_dropout_matches_masked_oracle(n):
    """Training-mode input dropout of the GGNN cell is applied inside the GEMM loaders with a stateless
    hash mask: replaying the same mask on the host, output and every gradient must equal the oracle cell
    run on x * mask / (1 - p) (so forward, dW_proj and dX all use one consistent mask).
    n = 90: 9000 rows -- the activation-sized launches, whose weight-gradient kernel gathers the embedding rows and masks
    its B fragments itself (no materialised masked operand)."""
    from get_amd import modules, ops
    rng = np.random.default_rng(31)
    r, d, h, p_drop, seed = 100, 48, 64, 0.25, 123457
    toks, lens, ids, adj = cases.graphs(rng, n, r, 3, O.convert_text, vocab=80)
    x = rng.standard_normal((n, r, d)).astype(np.float32)
    prm = cases.cell_params(rng, d, h)
    gw = rng.standard_normal((n, r, h)).astype(np.float32)
    mod = modules.GGNN(d, h, dropout=p_drop)
    _load_cell(mod, prm)
    mod = mod.to(DEV)
    padj, node_ids, _ = ops.graph_build(T(toks), T(lens), 3)
    keep = ops.dropout_mask_reference(seed, n * r, d, p_drop).reshape(n, r, d)
    assert abs(keep.mean() - (1 - p_drop)) < 0.01                       # the mask has the right density
    xt = T(x, grad=True)
    out = ops.ggnn_cell(padj, xt, None, mod._params(), p_drop, seed)
    (out * T(gw)).sum().backward()
    po = {k: torch.from_numpy(v).requires_grad_(True) for k, v in prm.items()}
    xo = torch.from_numpy(x).requires_grad_(True)
    oo = O.ggnn_cell(torch.from_numpy(adj).float(), xo * torch.from_numpy(keep).float() / (1 - p_drop), po)
    (oo * torch.from_numpy(gw)).sum().backward()
    assert maxerr(out.detach().cpu(), oo.detach()) <= 2e-5
    assert maxerr(xt.grad.cpu(), xo.grad) <= 1e-3 * float(xo.grad.abs().max()) + 1e-6
    for name, q in mod.named_parameters():
        go = po[name].grad
        assert maxerr(q.grad.cpu(), go) <= 1e-3 * float(go.abs().max()) + 1e-5, name
    # embedding-gather path uses the same mask indexing (logical row, not table row)
    emb = rng.standard_normal((80, d)).astype(np.float32)
    e = torch.nn.Embedding.from_pretrained(torch.from_numpy(emb), freeze=True).to(DEV)
    out2 = ops.ggnn_cell(padj, e.weight, node_ids.reshape(-1), mod._params(), p_drop, seed)
    xe = torch.from_numpy(emb)[torch.from_numpy(ids)]
    oo2 = O.ggnn_cell(torch.from_numpy(adj).float(), xe * torch.from_numpy(keep).float() / (1 - p_drop),
                      {k: v.detach() for k, v in po.items()})
    assert maxerr(out2.detach().cpu(), oo2) <= 2e-5
    # ... and so does the projection's weight gradient on that path (ids + mask inside the weight-gradient kernel's loader)
    mod.zero_grad()
    (out2 * T(gw)).sum().backward()
    po2 = {k: v.detach().clone().requires_grad_(True) for k, v in po.items()}
    oo2g = O.ggnn_cell(torch.from_numpy(adj).float(), xe * torch.from_numpy(keep).float() / (1 - p_drop), po2)
    (oo2g * torch.from_numpy(gw)).sum().backward()
    for name, q in mod.named_parameters():
        go = po2[name].grad
        assert maxerr(q.grad.cpu(), go) <= 1e-3 * float(go.abs().max()) + 1e-5, name
    # module-level: train mode draws a fresh seed per call, eval mode is deterministic
    mod.train(True)
    a1, a2 = mod(padj, T(x)), mod(padj, T(x))
    assert maxerr(a1.detach().cpu(), a2.detach().cpu()) > 1e-3
    mod.train(False)
    assert maxerr(mod(padj, T(x)).detach().cpu(), mod(padj, T(x)).detach().cpu()) == 0.0


# ---------------------------------------------------------------- node-compact row layout (ops.RaggedPlan)
def _compact_fixture(rng, n=7, r=100, vocab=60, empty=True):
    from get_amd import ops
    toks, lens, ids, adj = cases.graphs(rng, n, r, 3, O.convert_text, vocab=vocab)
    if empty:
        lens[1] = 0                               # an empty text: no real node at all
    toks[2] = np.arange(1000, 1000 + r); lens[2] = r  # all tokens distinct: no padding node at all
    for i in (1, 2):
        w, a, _ = O.convert_text([int(t) for t in toks[i]], r, int(lens[i]), 3)
        ids[i], adj[i] = np.asarray(w), np.asarray(a)
    padj, node_ids, n_nodes = ops.graph_build(T(toks), T(lens), 3)
    nn = n_nodes.cpu().numpy()
    plan = ops.RaggedPlan(n_nodes, node_ids, int(nn.sum()))
    return toks, lens, ids, adj, padj, node_ids, nn, plan


def test_compact_plan_maps():
    rng = np.random.default_rng(51)
    toks, lens, ids, adj, padj, node_ids, nn, plan = _compact_fixture(rng)
    n, r = ids.shape
    assert nn[1] == 0 and nn[2] == r
    goff = plan.goff.cpu().numpy()
    assert np.array_equal(goff, np.concatenate([[0], np.cumsum(nn)]))
    src = plan.src.cpu().numpy()
    assert np.array_equal(np.sort(src), np.arange(n * r))                 # a permutation of the padded rows
    assert np.array_equal(plan.rowg.cpu().numpy(), src // r)
    assert np.array_equal(plan.cids.cpu().numpy(), ids.reshape(-1)[src])
    g, j = src // r, src % r
    assert np.all(j[:plan.m_real] < nn[g[:plan.m_real]]) and np.all(j[plan.m_real:] >= nn[g[plan.m_real:]])
    assert np.all(np.diff(src[:plan.m_real]) > 0) and np.all(np.diff(src[plan.m_real:]) > 0)
    x = T(rng.standard_normal((n, r, 8)).astype(np.float32))
    assert torch.equal(plan.to_padded(plan.from_padded(x)), x)


@pytest.mark.parametrize("h", [300, 30])
def test_compact_spmm_equals_padded(h):
    from get_amd import ops
    rng = np.random.default_rng(52)
    toks, lens, ids, adj, padj, node_ids, nn, plan = _compact_fixture(rng)
    n, r = ids.shape
    x = T(rng.standard_normal((n, r, h)).astype(np.float32))
    keep = ops.gsl_topk(T(rng.standard_normal((n, r)).astype(np.float32)), 60)
    for a in (padj, padj.with_keep(keep)):
        yp = ops.spmm(a, x)
        yc = ops.spmm(a, plan.from_padded(x, plan.m_real), plan)
        assert torch.equal(yc, plan.from_padded(yp, plan.m_real))           # same neighbours, same order: bit-exact


@pytest.mark.parametrize("h,n_graphs,window", [(768, 300, 5), (256, 40, 3), (520, 9, 5)])
def test_spmm_bf16_is_the_fp32_aggregation_rounded_once(h, n_graphs, window):
    """gh_spmm_bf16 (the aggregation of the bf16 storage pipeline; R <= 128: a dense product per graph on the matrix pipe with the
    fp32 edge weights split three ways into bf16, csrc/graph_ops.hip spmm_mfma_bf16_kernel) against gh_spmm on the same bf16 values held
    in fp32: products exact to fp32 precision, fp32 accumulation, ONE rounding to bf16 -- so the result is the fp32 kernel's rounded
    once, except where the different fp32 summation order moves a sum across a rounding boundary: never more than one bf16 ulp, and
    for at most 0.2 % of the elements.  Padded and node-compact layout, with a keep set, transposed, accumulating.  Hub rows included."""
    from get_amd import _lib, ops
    rng = np.random.default_rng(4000 + h)
    n, r = n_graphs, 100
    toks = rng.integers(100, 5000, size=(n, r)).astype(np.int32)
    for g in range(n):
        toks[g, ::3] = 7 + (g % 3)                       # a hub token
    lens = rng.integers(30, r + 1, size=(n,)).astype(np.int32)
    padj, node_ids, n_nodes = ops.graph_build(T(toks), T(lens), window)
    plan = ops.RaggedPlan(n_nodes, node_ids, int(n_nodes.sum().item()))
    keep = ops.gsl_topk(T(rng.standard_normal((n, r)).astype(np.float32)), 60)
    x16 = T(rng.standard_normal((n, r, h)).astype(np.float32)).to(torch.bfloat16)
    y0 = T(rng.standard_normal((n, r, h)).astype(np.float32)).to(torch.bfloat16)
    for a in (padj, padj.with_keep(keep)):
        for pl in (None, plan):
            xin = x16 if pl is None else plan.from_padded(x16.float(), plan.m_real).to(torch.bfloat16).contiguous()
            yin = y0 if pl is None else plan.from_padded(y0.float(), plan.m_real).to(torch.bfloat16).contiguous()
            for tr, acc in ((0, 0), (1, 1)):
                x32 = xin.float().contiguous()
                y32 = yin.float().contiguous() if acc else torch.empty_like(x32)
                _lib.call("gh_spmm", *a._args(), *ops._plan_args(pl), _lib.ptr(x32), _lib.ptr(y32), n, r, h, tr, acc, _lib.stream())
                y16 = yin.clone() if acc else torch.empty_like(xin)
                _lib.call("gh_spmm_bf16", *a._args(), *ops._plan_args(pl), _lib.ptr(xin), _lib.ptr(y16), n, r, h, tr, acc, _lib.stream())
                torch.cuda.synchronize()
                # scale of the sum: the same aggregation of |x| (the weights are positive), plus |y| when accumulating
                s32 = yin.float().abs().contiguous() if acc else torch.empty_like(x32)
                _lib.call("gh_spmm", *a._args(), *ops._plan_args(pl), _lib.ptr(x32.abs().contiguous()), _lib.ptr(s32), n, r, h, tr, acc, _lib.stream())
                torch.cuda.synchronize()
                want = y32.to(torch.bfloat16)
                # (1) against the fp32 sum: half a bf16 ulp of the result + fp32-level error of the sum (cancelling sums included)
                err = (y16.float() - y32).abs()
                assert bool((err <= y32.abs() * 2.0 ** -8 + s32 * 4e-6 + 1e-30).all()), (tr, acc, float((err - y32.abs() * 2.0 ** -8 - s32 * 4e-6).max()))
                # (2) against the fp32 kernel's result rounded once: identical bits except at rounding ties (sums that do not cancel)
                solid = y32.abs() > 1e-2 * s32
                d = (y16.view(torch.int16).int() - want.view(torch.int16).int()).abs()
                assert int(d[solid].max()) <= 1, (tr, acc, int(d[solid].max()))
                assert float((d[solid] != 0).float().mean()) <= 2e-3, (tr, acc, float((d[solid] != 0).float().mean()))


@pytest.mark.parametrize("r", [30, 64, 128, 130])
def test_spmm_bf16_weighted_asymmetric_adjacency(r):
    """gh_spmm_bf16 on a dense WEIGHTED adjacency with an asymmetric pattern (reference-API hand-over: values, not d^-1/2 products),
    forward and transposed, against a float64 product of the same bf16 activations: half a bf16 ulp of the result + 4e-6 of the
    sum's scale.  r = 30 / 64 / 128: one and two bit words per row, k-steps that end exactly at and before the tile edge (the matrix-pipe
    kernel); r = 130: the edge-list kernel.  13 graphs: not a multiple of the 8 XCD queues."""
    from get_amd import _lib, ops
    rng = np.random.default_rng(700 + r)
    n, h = 13, 264
    a = rng.standard_normal((n, r, r)) * (rng.random((n, r, r)) < 0.08)
    a[:, np.arange(r), np.arange(r)] = 1.0
    a[:, :, r - 1] = 0.0
    a[:, r - 1, : r // 2] = 0.5
    a[3] = 0.0                                                                       # a graph without any edge
    pw = ops.PackedAdj.from_dense(T(a))
    a32 = pw.to_dense().double()
    x16 = T(rng.standard_normal((n, r, h)).astype(np.float32)).to(torch.bfloat16)
    y0 = T(rng.standard_normal((n, r, h)).astype(np.float32)).to(torch.bfloat16)
    for tr, acc in ((0, 0), (1, 0), (1, 1)):
        y16 = y0.clone() if acc else torch.full_like(x16, float("nan"))
        _lib.call("gh_spmm_bf16", *pw._args(), None, 0, _lib.ptr(x16), _lib.ptr(y16), n, r, h, tr, acc, _lib.stream())
        torch.cuda.synchronize()
        am = a32.transpose(1, 2) if tr else a32
        ref = am @ x16.double() + (y0.double() if acc else 0.0)
        scale = am.abs() @ x16.double().abs() + (y0.double().abs() if acc else 0.0)
        err = (y16.double() - ref).abs()
        assert bool(torch.isfinite(y16.float()).all()), (tr, acc)
        assert bool((err <= ref.abs() * 2.0 ** -8 + scale * 4e-6 + 1e-30).all()), (tr, acc, float((err - ref.abs() * 2.0 ** -8 - scale * 4e-6).max()))


@pytest.mark.parametrize("n_graphs", [9, 300])
def test_spmm_hub_rows_split_over_work_items_match_dense_fp64(n_graphs):
    """Word graphs with a hub node (a token at every third position: degree ~2/3 of the nodes).  In the node-compact layout
    the aggregation kernel cuts such rows into 8-edge groups handled by separate work items whose partial sums meet in the
    slab image's free rows (csrc/graph_ops.hip); the result must match a dense float64 product, the padded layout (which
    never splits) bit for bit, and the backward (transposed, accumulating launch) through autograd.  300 graphs: the
    one-workgroup-per-graph launch mode; 9: one slab per workgroup."""
    from get_amd import ops
    rng = np.random.default_rng(77)
    n, r, h = n_graphs, 100, 300
    toks = rng.integers(100, 5000, size=(n, r)).astype(np.int32)
    for g in range(n):
        toks[g, ::3] = 7 + (g % 3)                       # the hub token
        if g % 4 == 1:
            toks[g, 1::6] = 50                           # a second, mid-degree hub
    lens = np.full((n,), r, np.int32)
    lens[0] = 40                                         # a short text: few nodes, many free rows
    padj, node_ids, n_nodes = ops.graph_build(T(toks), T(lens), 3)
    nn = n_nodes.cpu().numpy()
    dense = padj.to_dense()
    deg = (dense != 0).sum(-1).max(-1).values.cpu().numpy()
    assert deg.max() >= 40 and nn.max() < r, "fixture: hub rows and free rows are both present"
    plan = ops.RaggedPlan(n_nodes, node_ids, int(nn.sum()))
    x = T(rng.standard_normal((n, r, h)).astype(np.float32))
    keep = ops.gsl_topk(T(rng.standard_normal((n, r)).astype(np.float32)), 60)
    for a in (padj, padj.with_keep(keep)):
        xc = plan.from_padded(x, plan.m_real).requires_grad_(True)
        yc = ops.spmm(a, xc, plan)
        yp = ops.spmm(a, x)
        assert torch.equal(yc.detach(), plan.from_padded(yp, plan.m_real))          # same groups, same order: bit-exact
        ref = torch.matmul(a.to_dense().double(), x.double())
        assert float((plan.from_padded(ref, plan.m_real) - yc.detach().double()).abs().max()) <= 2e-5
        gy = T(rng.standard_normal(tuple(yc.shape)).astype(np.float32))
        yc.backward(gy)
        gref = torch.matmul(a.to_dense().double().transpose(1, 2), plan.to_padded(gy).double())
        assert float((plan.from_padded(gref, plan.m_real) - xc.grad.double()).abs().max()) <= 2e-5


def test_compact_cell_with_dropout_vs_oracle():
    """Cell on the node-compact layout, training-mode dropout on: forward over ALL rows (padding rows included),
    backward over the real rows; the stateless mask is indexed by the compact row."""
    from get_amd import modules, ops
    rng = np.random.default_rng(53)
    toks, lens, ids, adj, padj, node_ids, nn, plan = _compact_fixture(rng)
    n, r = ids.shape
    d, h, p_drop, seed = 48, 64, 0.25, 99173
    x = rng.standard_normal((n, r, d)).astype(np.float32)
    prm = cases.cell_params(rng, d, h)
    gw = rng.standard_normal((n, r, h)).astype(np.float32)
    src = plan.src.cpu().numpy()
    real = np.zeros(n * r, bool); real[src[:plan.m_real]] = True
    gw.reshape(n * r, h)[~real] = 0.0                               # no consumer reads the padding rows' output
    mod = modules.GGNN(d, h, dropout=p_drop)
    _load_cell(mod, prm)
    mod = mod.to(DEV)
    keep_c = ops.dropout_mask_reference(seed, n * r, d, p_drop)      # indexed by compact row
    keep = np.zeros_like(keep_c); keep[src] = keep_c
    keep = keep.reshape(n, r, d)
    xc = plan.from_padded(T(x)).requires_grad_(True)                # (m_tot, d)
    out = ops.ggnn_cell(padj, xc, None, mod._params(), p_drop, seed, plan=plan, rows=plan.m_tot)
    assert out.shape == (n * r, h)
    (out * plan.from_padded(T(gw))).sum().backward()
    po = {k: torch.from_numpy(v).requires_grad_(True) for k, v in prm.items()}
    xo = torch.from_numpy(x).requires_grad_(True)
    oo = O.ggnn_cell(torch.from_numpy(adj).float(), xo * torch.from_numpy(keep).float() / (1 - p_drop), po)
    (oo * torch.from_numpy(gw)).sum().backward()
    assert maxerr(plan.to_padded(out.detach()).cpu(), oo.detach()) <= 2e-5
    dx = plan.to_padded(xc.grad).cpu()
    assert maxerr(dx, xo.grad) <= 1e-3 * float(xo.grad.abs().max()) + 1e-6
    assert float(xc.grad[plan.m_real:].abs().max()) == 0.0
    for name, q in mod.named_parameters():
        go = po[name].grad
        assert maxerr(q.grad.cpu(), go) <= 1e-3 * float(go.abs().max()) + 1e-5, name
    # real rows only (the second cell of GGNN_with_GSL): same numbers on the prefix
    out_r = ops.ggnn_cell(padj, xc.detach(), None, mod._params(), p_drop, seed, plan=plan, rows=plan.m_real)
    # (not bit-equal: few-row launches split K by the number of row tiles, so the summation order depends on the row count)
    assert maxerr(out_r.detach().cpu(), out.detach()[:plan.m_real].cpu()) <= 2e-6


def test_compact_scorer_gsl_and_attention_equal_padded():
    from get_amd import modules, ops
    rng = np.random.default_rng(54)
    toks, lens, ids, adj, padj, node_ids, nn, plan = _compact_fixture(rng)
    n, r = ids.shape
    h, heads = 64, 5
    feat = T(rng.standard_normal((n, r, h)).astype(np.float32))
    w_p = T(rng.standard_normal((1, h)).astype(np.float32) * 0.2)
    gate = T(rng.standard_normal((12,)).astype(np.float32))
    for drop in ((0.0, 0), ):
        s_p, k_p = ops.scorer_gsl(padj, feat, w_p, gate, 60, *drop)
        s_c, k_c = ops.scorer_gsl(padj, plan.from_padded(feat), w_p, gate, 60, *drop, plan=plan)
        assert torch.equal(s_p, s_c) and torch.equal(k_p, k_c)
    # training mode: the mask follows the compact row index, so compare against the padded kernel fed the
    # already-masked features
    p_drop, seed = 0.3, 4711
    keep_c = torch.from_numpy(ops.dropout_mask_reference(seed, n * r, h, p_drop)).to(DEV)
    fc = plan.from_padded(feat)
    s_c, k_c = ops.scorer_gsl(padj, fc, w_p, gate, 60, p_drop, seed, plan=plan)
    s_p, k_p = ops.scorer_gsl(padj, plan.to_padded(fc * keep_c / (1 - p_drop)), w_p, gate, 60)
    assert maxerr(s_p.cpu(), s_c.cpu()) <= 1e-5
    # attention: softmax over the real rows only == masked softmax over the padded rows.  An empty graph gives NaN
    # in both layouts (softmax over nothing, two_branches_attention.py:144-146) ...
    att = modules.ConcatNotEqualSelfAtt(2 * h, h, heads).to(DEV)
    left = T(rng.standard_normal((n, h)).astype(np.float32))
    a_p, _ = att(left, feat, node_ids >= 1)
    a_c, _ = att(left, plan.from_padded(feat, plan.m_real), plan.cids[:plan.m_real] >= 1, plan=plan)
    assert torch.isnan(a_p[1]).all() and torch.isnan(a_c[1]).all()
    # ... so the gradient comparison uses a batch without one (NaN would spread into the weight gradients)
    toks, lens, ids, adj, padj, node_ids, nn, plan = _compact_fixture(rng, empty=False)
    mask = (node_ids >= 1)
    ga = T(rng.standard_normal((n, h, heads)).astype(np.float32))
    res = {}
    for mode in ("padded", "compact"):
        att.zero_grad()
        l_ = left.clone().requires_grad_(True)
        if mode == "padded":
            r_ = feat.clone().requires_grad_(True)
            a_, w_ = att(l_, r_, mask)
        else:
            r_ = plan.from_padded(feat, plan.m_real).requires_grad_(True)
            a_, w_ = att(l_, r_, plan.cids[:plan.m_real] >= 1, plan=plan)
        (a_ * ga).sum().backward()
        dr = r_.grad if mode == "padded" else plan.to_padded(r_.grad)
        wp = w_ if mode == "padded" else plan.to_padded(w_)
        res[mode] = (a_.detach(), wp.detach(), l_.grad, dr, att.linear1.weight.grad.clone(), att.linear2.weight.grad.clone())
    for i, (a, b) in enumerate(zip(res["padded"], res["compact"])):
        scale = float(a.abs().max())
        assert maxerr(a.cpu(), b.cpu()) <= 2e-6 * max(scale, 1.0) + 1e-6 * scale, i


# ---------------------------------------------------------------- opt-in bf16 operand mode of the big NT/NN GEMMs
@pytest.fixture
def bf16_mode():
    from get_amd import _lib
    _lib.set_gemm_mode("bf16")
    yield
    _lib.set_gemm_mode("fp32")


@pytest.mark.parametrize("m,k,n", [(9000, 300, 300), (8200, 768, 768), (12345, 64, 48)])
def test_bf16_mode_linear_matches_bf16_rounded_reference(bf16_mode, m, k, n):
    """gh_set_gemm_mode(1): operands rounded to bf16 (RNE) in LDS, fp32 accumulate.  Against the same rounding done on
    the host the result must agree to fp32 summation noise (forward, input gradient and weight gradient)."""
    from get_amd import ops
    rng = np.random.default_rng(m)
    x = T(rng.standard_normal((m, k)).astype(np.float32), grad=True)
    w = T((rng.standard_normal((n, k)) / np.sqrt(k)).astype(np.float32), grad=True)
    b = T(rng.standard_normal((n,)).astype(np.float32), grad=True)
    g = T(rng.standard_normal((m, n)).astype(np.float32))
    y = ops.linear(x, w, b)
    (y * g).sum().backward()
    r = lambda t: t.detach().bfloat16().float()
    ref = r(x) @ r(w).t() + b.detach()
    assert maxerr(y.detach().cpu(), ref.cpu()) <= 2e-5 * max(1.0, float(ref.abs().max()))
    dx_ref = r(g) @ r(w)
    assert maxerr(x.grad.cpu(), dx_ref.cpu()) <= 2e-5 * max(1.0, float(dx_ref.abs().max()))
    dw_ref = r(g).t() @ r(x)                                     # the weight-gradient GEMM rounds its operands too
    assert maxerr(w.grad.cpu(), dw_ref.cpu()) <= 1e-4 * float(dw_ref.abs().max())
    # and the mode really is different from fp32
    full = x.detach() @ w.detach().t() + b.detach()
    assert maxerr(y.detach().cpu(), full.cpu()) > 1e-4


@pytest.mark.parametrize("compact", [False, True])
@pytest.mark.parametrize("drop_p", [0.0, 0.2])
def test_bf16_storage_cell_tracks_fp32(bf16_mode, compact, drop_p):
    """BASELINE configs[4] path: gh_ggnn_cell_fwd/bwd_bf16 (bf16 activations and weights in HBM, v_mfma_f32_16x16x32_bf16, the
    transpose-read weight-gradient kernel) against the fp32 cell on the same inputs and the same dropout mask.
    Stated tolerance: outputs 3e-2 absolute (|out| <= ~2), gradients 5e-2 of their largest entry."""
    from get_amd import _lib, modules, ops
    rng = np.random.default_rng(77)
    n, r, d, h = (300 if compact else 100), 100, 128, 256
    from get_amd.synth import make_tokens
    toks, lens = make_tokens(rng, n, r, 5000, r // 2, r)
    padj, node_ids, nn = ops.graph_build(T(toks), T(lens), 3)
    plan = ops.RaggedPlan(nn, node_ids, int(nn.sum().item())) if compact else None
    assert (plan.m_real if compact else n * r) >= 8192
    x = rng.standard_normal((n, r, d)).astype(np.float32) * 0.5
    gw = rng.standard_normal((n, r, h)).astype(np.float32)
    prm = cases.cell_params(rng, d, h)
    mod = modules.GGNN(d, h, dropout=0.0)
    _load_cell(mod, prm)
    mod = mod.to(DEV)
    res = {}
    for mode in ("fp32", "bf16"):
        _lib.set_gemm_mode(mode)
        for p_ in mod.parameters():
            p_.grad = None
        if compact:
            xc = plan.from_padded(T(x))[:plan.m_real].clone().requires_grad_(True)
            out = ops.ggnn_cell(padj, xc, None, mod._params(), drop_p, 4242, plan=plan, rows=plan.m_real)
            (out * plan.from_padded(T(gw))[:plan.m_real]).sum().backward()
        else:
            xc = T(x, grad=True)
            out = ops.ggnn_cell(padj, xc, None, mod._params(), drop_p, 4242)
            (out * T(gw)).sum().backward()
        res[mode] = (out.detach().clone(), xc.grad.clone(), {k: q.grad.clone() for k, q in mod.named_parameters()})
    assert getattr(res["bf16"][0], "dtype") == torch.float32
    o32, dx32, g32 = res["fp32"]
    o16, dx16, g16 = res["bf16"]
    assert 1e-6 < float((o16 - o32).abs().max()) <= 3e-2
    assert float((dx16 - dx32).abs().max()) <= 5e-2 * float(dx32.abs().max())
    for k in g32:
        assert float((g16[k] - g32[k]).abs().max()) <= 5e-2 * float(g32[k].abs().max()) + 1e-6, k


@pytest.mark.parametrize("h,din,rows_kind", [(256, 256, "padded"), (512, 256, "compact"), (256, 128, "compact")])
def test_bf16_cell_on_the_256_tile_tracks_fp32(bf16_mode, h, din, rows_kind):
    """The 256 x 256 x 64 ping-pong tile (gemm_nt_pp.hip.h; default for bf16-storage launches of >= 32 768 rows) at widths other than
    the bench's 768: one and two column blocks, a projection that is (256) and is not (128) tile-eligible, padded and node-compact
    rows, M not a multiple of 256, dropout on -- gh_ggnn_cell_fwd/bwd_bf16 against the fp32 cell on the same inputs and mask.
    Same stated tolerance as test_bf16_storage_cell_tracks_fp32: outputs 3e-2 absolute, gradients 5e-2 of their largest entry."""
    from get_amd import _lib, modules, ops
    from get_amd.synth import make_tokens
    rng = np.random.default_rng(1000 + h + din)
    compact = rows_kind == "compact"
    n, r = (760 if compact else 333), 100
    toks, lens = make_tokens(rng, n, r, 5000, r // 2, r)
    padj, node_ids, nn = ops.graph_build(T(toks), T(lens), 3)
    plan = ops.RaggedPlan(nn, node_ids, int(nn.sum().item())) if compact else None
    rows = plan.m_real if compact else n * r
    assert rows >= 32768 and rows % 256 != 0
    x = rng.standard_normal((n, r, din)).astype(np.float32) * 0.5
    gw = rng.standard_normal((n, r, h)).astype(np.float32)
    prm = cases.cell_params(rng, din, h)
    mod = modules.GGNN(din, h, dropout=0.0)
    _load_cell(mod, prm)
    mod = mod.to(DEV)
    res = {}
    for mode in ("fp32", "bf16"):
        _lib.set_gemm_mode(mode)
        for p_ in mod.parameters():
            p_.grad = None
        if compact:
            xc = plan.from_padded(T(x))[:plan.m_real].clone().requires_grad_(True)
            out = ops.ggnn_cell(padj, xc, None, mod._params(), 0.2, 777, plan=plan, rows=plan.m_real)
            (out * plan.from_padded(T(gw))[:plan.m_real]).sum().backward()
        else:
            xc = T(x, grad=True)
            out = ops.ggnn_cell(padj, xc, None, mod._params(), 0.2, 777)
            (out * T(gw)).sum().backward()
        res[mode] = (out.detach().clone(), xc.grad.clone(), {k: q.grad.clone() for k, q in mod.named_parameters()})
    o32, dx32, g32 = res["fp32"]
    o16, dx16, g16 = res["bf16"]
    assert 1e-6 < float((o16 - o32).abs().max()) <= 3e-2
    assert float((dx16 - dx32).abs().max()) <= 5e-2 * float(dx32.abs().max())
    for k in g32:
        assert float((g16[k] - g32[k]).abs().max()) <= 5e-2 * float(g32[k].abs().max()) + 1e-6, k


@pytest.mark.parametrize("m,n,k,pad", [
    (62208, 768, 768, 0),        # the configs[4] cell shape: 9 tiles x 28 chunks
    (20011, 256, 512, 0),        # ragged rows: the last K tile of every chunk is partial, 2 tiles
    (16384 + 37, 512, 256, 64),  # operand rows with a leading dimension (column views of wider buffers)
    (40000, 1024, 768, 0),       # 12 tiles: two rounds' worth of chunks
    (5000, 768, 768, 0),         # below the row threshold: the 128 x 320 kernel (same contract)
    (33000, 264, 520, 0),        # widths that are not multiples of 256: the 128 x 320 kernel
])
def test_bf16_weight_gradient_gemm_exact(m, n, k, pad):
    """gh_linear_wgrad_bf16 (the weight-gradient GEMM of the bf16 storage pipeline; outputs in multiples of 256 over >= 16 384 rows run
    on the 256 x 256 x 64 ping-pong tile, gemm_tn_pp.hip.h) against float64 products of the SAME bf16 operand values: the kernel's
    products are exact and its accumulation is fp32, so the bound is accumulation rounding only -- 2e-5 of the largest entry, which a
    lost or doubled K tile, a wrong chunk boundary or a swizzle slip would exceed by orders of magnitude.  dw and db ACCUMULATE into
    the caller's buffers (checked with non-zero initial contents)."""
    from get_amd import ops
    gen = torch.Generator(device="cpu").manual_seed(m + n + k)
    gfull = (torch.randn(m, n + pad, generator=gen) * 0.5).to(torch.bfloat16).to(DEV)
    xfull = (torch.randn(m, k + pad, generator=gen)).to(torch.bfloat16).to(DEV)
    g16, x16 = gfull[:, pad // 2: pad // 2 + n], xfull[:, pad // 2: pad // 2 + k]
    if pad:
        assert g16.data_ptr() % 16 == 0 and x16.data_ptr() % 16 == 0
    dw0 = torch.randn(n, k, generator=gen).to(DEV)
    db0 = torch.randn(n, generator=gen).to(DEV)
    dw, db = dw0.clone(), db0.clone()
    ops.linear_wgrad_bf16(g16, x16, dw, db)
    torch.cuda.synchronize()
    ref_w = torch.zeros(n, k, dtype=torch.float64, device=DEV)
    for r0 in range(0, m, 8192):      # (float64 GEMM in row slabs: bounded memory)
        ref_w += g16[r0:r0 + 8192].double().t() @ x16[r0:r0 + 8192].double()
    ref_b = g16.double().sum(0)
    ew = float(((dw - dw0).double() - ref_w).abs().max()) / float(ref_w.abs().max())
    eb = float(((db - db0).double() - ref_b).abs().max()) / float(ref_b.abs().max())
    assert ew <= 2e-5, ew
    assert eb <= 2e-5, eb


@pytest.mark.parametrize("ws_mib", [16, 3])
def test_bf16_weight_gradient_gemm_with_a_small_workspace(ws_mib):
    """The ping-pong weight-gradient kernel only writes partial tiles, so its K split is bounded by the split-K workspace: with 16 MiB a
    768 x 768 output (2.25 MiB of partials per chunk + its bias-gradient partials) gets 6 chunks instead of 28, with 3 MiB a single one
    (the reduction then only adds that one tile into the caller's buffer).  Same exact bound as above in both cases; the default
    workspace is restored afterwards."""
    from get_amd import _lib, ops
    m, n, k = 20000, 768, 768
    gen = torch.Generator(device="cpu").manual_seed(ws_mib)
    g16 = (torch.randn(m, n, generator=gen) * 0.5).to(torch.bfloat16).to(DEV)
    x16 = torch.randn(m, k, generator=gen).to(torch.bfloat16).to(DEV)
    dw, db = torch.zeros(n, k, device=DEV), torch.zeros(n, device=DEV)
    _lib.ensure_workspace(g16.device)
    small = torch.empty(ws_mib << 18, device=DEV, dtype=torch.float32)
    st = _lib.stream()
    try:
        _lib.call("gh_set_stream_workspace", st, small.data_ptr(), small.numel() * 4)
        _lib.call("gh_linear_wgrad_bf16", g16.data_ptr(), n, x16.data_ptr(), k, m, n, k, dw.data_ptr(), k, db.data_ptr(), st)
        torch.cuda.synchronize()
    finally:
        _lib._workspaces.clear()              # the next ensure_workspace registers a default-sized buffer again
        _lib.ensure_workspace(g16.device)
    ref_w = g16.double().t() @ x16.double()
    ref_b = g16.double().sum(0)
    assert float((dw.double() - ref_w).abs().max()) <= 2e-5 * float(ref_w.abs().max())
    assert float((db.double() - ref_b).abs().max()) <= 2e-5 * float(ref_b.abs().max())


def test_evd_assemble_bwd_writes_every_row_of_d_avg():
    """gh_evd_assemble_bwd needs no cleared d_avg: the rows of real (claim, slot) pairs receive their gradient, the rows of a claim's
    evidences beyond its n_max-th (no slot in the padded tensor) receive zeros -- checked on a NaN-filled buffer with one claim over
    n_max, one empty claim and one exactly at n_max."""
    from get_amd import ops
    from get_amd._lib import call, ptr, stream
    rng = np.random.default_rng(5)
    n_max, xa, ds = 4, 24, 8
    counts = np.array([2, 7, 0, 4, 1], dtype=np.int64)            # 7 > n_max: pairs 4..6 of that claim have no slot
    b, b1 = len(counts), int(counts.sum())
    seg = ops.Segments(T(counts), b1, n_max)
    g = rng.standard_normal((b, n_max, xa + ds)).astype(np.float32)
    src = rng.integers(0, 6, size=(b, n_max)).astype(np.int32)
    d_avg = torch.full((b1, xa), float("nan"), device=DEV)
    d_table = torch.zeros((6, ds), device=DEV)
    gt, st = T(g), T(src)
    call("gh_evd_assemble_bwd", ptr(gt), ptr(seg.offsets), ptr(st), 0, 6, b, n_max, xa, ds, ptr(d_avg), ptr(d_table), stream())
    torch.cuda.synchronize()
    got = d_avg.cpu().numpy()
    assert np.isfinite(got).all()
    lo = 0
    for c, n in enumerate(counts):
        for j in range(int(n)):
            want = g[c, j, :xa] if j < n_max else np.zeros(xa, np.float32)
            assert np.array_equal(got[lo + j], want), (c, j)
        lo += int(n)
