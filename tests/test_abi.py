"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol
include/get_hip.h declares with the arity the Python binding assumes; the drop-in modules carry the
reference's state_dict names/shapes; the install() shim resolves the reference's import paths."""
import ctypes
import json
import os
import re
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "get_hip.h")


@pytest.fixture(scope="module")
def lib_path():
    p = os.path.join(ROOT, "get_amd", "lib", "libget_hip.so")
    if not os.path.exists(p):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "get_amd", "csrc"), "-j4"])
    return p


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(?:int|const char\*)\s+(gh_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if args == "void" else len([a for a in args.split(",") if a.strip()])
    return out


def test_library_exports_every_declared_symbol(lib_path):
    from get_amd import _lib
    lib = ctypes.CDLL(lib_path)
    decl = declared_functions()
    assert len(decl) >= 20
    for name, nargs in decl.items():
        assert hasattr(lib, name), f"{name} declared in get_hip.h but not exported"
        if name in _lib.SIGNATURES:
            assert len(_lib.SIGNATURES[name]) == nargs, f"{name}: binding has {len(_lib.SIGNATURES[name])} args, header {nargs}"
    for name in _lib.SIGNATURES:
        assert name in decl, f"{name} bound in Python but not declared in get_hip.h"
    lib.gh_abi_version.restype = ctypes.c_int
    assert lib.gh_abi_version() == 10
    _lib.load()


def test_no_cpu_fallback_ops_refuse_cpu_tensors(lib_path):
    from get_amd import ops
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.PackedAdj.from_dense(torch.zeros(1, 4, 4))
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.graph_build(torch.zeros(1, 4, dtype=torch.int32), torch.ones(1, dtype=torch.int32), 3)


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "get_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, f


def test_state_dict_contract_matches_reference(golden_dir):
    from get_amd import modules
    from get_amd.synth import make_embeddings
    from oracle.cases_model import MODEL_CASES
    cfg, seed = MODEL_CASES["small"]
    emb, art, clm = make_embeddings(cfg, seed)
    params = cfg.model_params(emb, art, clm)
    model = modules.Graph_basedSemantiStructure(params)
    assert params["embedding_input_dim"] == cfg.vocab and params["embedding_output_dim"] == cfg.emb_dim
    contract = json.load(open(os.path.join(golden_dir, "state_dict_contract_small.json")))
    sd = {k: list(v.shape) for k, v in model.state_dict().items()}
    assert sd == contract
    assert not model.embedding.weight.requires_grad and model.article_source_embs.weight.requires_grad


@pytest.mark.skipif(not os.path.isdir("/root/reference/MasterFC"), reason="reference tree only exists in the build container")
def test_install_shim_resolves_reference_import_paths():
    code = r"""
import sys
sys.path.insert(0, %r)
from oracle import _refshim
_refshim.install()
import get_amd
M = get_amd.install()
from Models.FCWithEvidences import graph_based_semantic_structure      # as MasterFC/master_get.py:5 does
from Models.BiDAF.wrapper import GGNN, GGNN_with_GSL, Linear
from thirdparty.two_branches_attention import *
assert graph_based_semantic_structure.Graph_basedSemantiStructure is M.Graph_basedSemantiStructure
assert GGNN is M.GGNN and ConcatNotEqualSelfAtt is M.ConcatNotEqualSelfAtt
import setting_keywords                                                  # the reference's own vocabulary
from get_amd.keywords import KeyWordSettings as K
for name in [n for n in dir(K) if not n.startswith('_') and isinstance(getattr(K, n), str)]:
    assert getattr(setting_keywords.KeyWordSettings, name) == getattr(K, name), name
print('ok')
""" % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-2000:]


@pytest.mark.skipif(not os.path.isdir("/root/reference/MasterFC"), reason="reference tree only exists in the build container")
def test_master_get_itself_imports_against_the_shim_and_builds_the_hip_backed_model():
    """the reference's driver, MasterFC/master_get.py, imported UNMODIFIED (its own `from Models...
    import graph_based_semantic_structure` at :5, the fitter, matchzoo, handlers) after get_amd.install(): the model class
    it would construct at :145 is the HIP-backed drop-in, and constructing it with the params dict of :118-144 works."""
    code = r"""
import sys, os
sys.path.insert(0, %r)
from oracle import _refshim
_refshim.install()
import get_amd
M = get_amd.install()
sys.path.insert(0, '/root/reference/MasterFC')
os.chdir('/root/reference/MasterFC')
import master_get                                                        # the driver module itself, top to bottom
assert master_get.graph_based_semantic_structure.Graph_basedSemantiStructure is M.Graph_basedSemantiStructure
import numpy as np
# the params dict master_get.fit_models builds at :118-144 (sizes shrunk)
params = dict(embedding=np.random.rand(50, 16).astype('float32'), embedding_freeze=True, num_classes=2,
              fixed_length_left=30, fixed_length_right=100, use_claim_source=0, claim_source_embeddings=np.zeros((4, 8), 'float32'),
              use_article_source=1, article_source_embeddings=np.zeros((6, 8), 'float32'), cuda=0,
              num_att_heads_for_words=5, num_att_heads_for_evds=2, dropout_gnn=0.2, dropout_left=0.2, dropout_right=0.2,
              hidden_size=16, output_size=2, gsl_rate=0.6)
net = master_get.graph_based_semantic_structure.Graph_basedSemantiStructure(params)
assert type(net.ggnn_with_gsl).__module__ == 'get_amd.modules'
assert params['embedding_input_dim'] == 50 and params['embedding_output_dim'] == 16
# the fitter class the driver instantiates at :148 accepts the drop-in model
fit = master_get.char_man_fitter_query_repr1.CharManFitterQueryRepr1
assert callable(fit)
print('ok')
""" % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-3000:]


def test_composite_descriptor_mirrors_and_buffer_plan(lib_path):
    """The ctypes mirrors in get_amd/fused.py must have the C structs' sizes (gh_get_struct_sizes), and gh_get_plan_buffers --
    pure host arithmetic, no device call -- must lay the observables out 256-byte aligned inside an arena that grows with the
    batch and rejects shapes the composite path does not take."""
    import ctypes
    from get_amd import _lib, fused
    _lib.load()
    sz = (ctypes.c_int64 * 5)()
    _lib.call("gh_get_struct_sizes", ctypes.cast(sz, ctypes.c_void_p))
    assert tuple(sz) == (ctypes.sizeof(fused.GetModel), ctypes.sizeof(fused.GetBatch), ctypes.sizeof(fused.GetPlan),
                         ctypes.sizeof(fused.CellParams), ctypes.sizeof(fused.CellBf16))

    def plan(b, b1, m_real, h=300, d=300, storage=0, twins=False):
        M, B, P = fused.GetModel(), fused.GetBatch(), fused.GetPlan()
        M.d, M.h, M.word_heads, M.evd_heads, M.n_classes, M.article_src_dim = d, h, 5, 2, 2, 128
        M.storage = storage
        if twins:          # (only checked for being non-NULL)
            M.embedding16 = 4096
            for c16 in (M.cell1_16, M.cell2_16):
                for name, _ in fused.CellBf16._fields_:
                    setattr(c16, name, 4096)
        B.b, B.b1, B.l, B.r, B.n_max, B.m_real, B.k_keep = b, b1, 30, 100, 30, m_real, 60
        if m_real >= 0:
            B.goff = B.rowg = B.cids = B.maskf = 4096        # (only checked for being non-NULL)
        _lib.call("gh_get_plan_buffers", ctypes.addressof(M), ctypes.addressof(B), ctypes.addressof(P))
        return P

    small, big, padded = plan(4, 40, 2600), plan(32, 960, 62128), plan(32, 960, -1)
    for p in (small, big, padded):
        for off in (p.phi, p.word_w, p.evd_w, p.score, p.keep):
            assert off % 64 == 0 and 0 <= off < p.obs_floats
        assert 0 < 4 * p.obs_floats < 16e6          # observables: a few MB, never the activation arena
    assert small.fwd_floats < big.fwd_floats < padded.fwd_floats and small.bwd_floats < big.bwd_floats
    assert 1.2e9 < 4 * big.fwd_floats < 2.5e9          # ~1.5 GB of saved activations at the bench shape
    # ABI 9: wide hidden layers (BASELINE configs[4], h = 768) take the composite path; beyond 1024 they do not
    wide = plan(32, 960, 62128, h=768, d=768)
    assert wide.fwd_floats > 2.3 * big.fwd_floats
    with pytest.raises(RuntimeError, match="1024"):
        plan(4, 40, 2600, h=1028, d=1028)
    # bf16 storage inside the evidence cells: needs the bf16 twins, halves the cells' saved activations, and only applies to
    # batches with >= 8192 real node rows (a small batch keeps the fp32 plan)
    with pytest.raises(RuntimeError, match="twins"):
        plan(32, 960, 62128, h=768, d=768, storage=1)
    wide16 = plan(32, 960, 62128, h=768, d=768, storage=1, twins=True)
    assert 0.5 * wide.fwd_floats < wide16.fwd_floats < 0.75 * wide.fwd_floats and wide16.bwd_floats < wide.bwd_floats
    assert plan(4, 40, 2600, h=768, d=768, storage=1, twins=True).fwd_floats == plan(4, 40, 2600, h=768, d=768).fwd_floats
    with pytest.raises(RuntimeError, match="storage"):
        plan(4, 40, 2600, storage=2)
    with pytest.raises(RuntimeError, match="m_real"):
        plan(4, 40, 40 * 100 + 1)
    # arena size classes: at most 1/8 above the request for large sizes
    f = fused._arena_floats
    assert f(1000) == 1000 and all(n <= f(n) <= n * 1.125 + 1 for n in (10 ** 8, 4 * 10 ** 8, 3 * 10 ** 9))


def test_driver_build_entry_point_runs(lib_path):
    """__graft_entry__.build() is what the driver calls every round: it must agree with the library's ABI version (it once
    hard-coded the previous one and failed after a bump)."""
    import importlib
    ge = importlib.import_module("__graft_entry__")
    ge.build()
