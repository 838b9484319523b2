"""ctypes binding of libget_hip.so (the C-ABI declared in include/get_hip.h).

There is NO fallback: if the library is missing or a call fails, a RuntimeError is
raised.  The product never routes through the oracle or a torch re-implementation.
"""
from __future__ import annotations

import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# GET_AMD_LIB: load another build of the same ABI instead (A/B runs of kernel variants on one box)
LIB_PATH = os.environ.get("GET_AMD_LIB") or os.path.join(_HERE, "lib", "libget_hip.so")

_P = ctypes.c_void_p
_I = ctypes.c_int
_F = ctypes.c_float
_L = ctypes.c_int64
_U = ctypes.c_uint32
ABI_VERSION = 10         # GH_ABI_VERSION of include/get_hip.h

# name -> argtypes (mirrors include/get_hip.h; tests/test_abi.py checks the two agree)
SIGNATURES = {
    "gh_graph_build": [_P, _P, _I, _I, _I, _P, _P, _P, _P, _P],
    "gh_adj_pack_f64": [_P, _I, _I, _P, _P, _P],
    "gh_ref_depad": [_P, _I, _I, _I, _P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _P],
    "gh_adj_pack_f32": [_P, _I, _I, _P, _P, _P],
    "gh_ragged_plan": [_P, _P, _I, _I, _P, _P, _P, _P, _P, _P],
    "gh_spmm": [_P, _P, _P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _P],
    "gh_spmm_bf16": [_P, _P, _P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _P],
    "gh_transpose": [_P, _P, _I, _I, _P],
    "gh_transpose_batch": [_I, _P, _P, _P, _P, _P],
    "gh_ggnn_cell_fwd": [_P] * 5 + [_I, _I] + [_P] * 2 + [_I] * 4 + [_P] * 13 + [_P] * 7 + [_F, _U, _P, _P, _F, _U, _P],
    "gh_ggnn_cell_bwd": [_P] * 5 + [_I] + [_P] * 2 + [_I] * 4 + [_P] * 7 + [_P] * 7 + [_P] * 5 + [_P] * 14 + [_F, _U, _P],
    "gh_ggnn_cell_fwd_bf16": [_P] * 5 + [_I, _I] + [_P] * 2 + [_I] * 4 + [_P] * 13 + [_P] * 8 + [_F, _U, _P, _P, _F, _U, _P],
    "gh_ggnn_cell_bwd_bf16": [_P] * 5 + [_I] + [_P] * 2 + [_I] * 4 + [_P] * 7 + [_P] * 7 + [_P] * 5 + [_P] * 14 + [_F, _U, _P],
    "gh_scorer_gsl": [_P, _P, _P, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _F, _U, _P],
    "gh_gsl_topk": [_P, _I, _I, _I, _P, _P],
    "gh_adj_unpack": [_P, _P, _P, _P, _I, _I, _P, _P],
    "gh_concat_att_fwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P],
    "gh_concat_att_bwd": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "gh_linear_fwd": [_P, _P, _P, _P, _I, _I, _I, _P],
    "gh_linear_bwd": [_P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P],
    "gh_linear_wgrad_bf16": [_P, _I, _P, _I, _I, _I, _I, _P, _I, _P, _P],
    "gh_evd_assemble_fwd": [_P, _P, _P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P],
    "gh_evd_assemble_bwd": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P],
    "gh_clamp_events": [_P, _I],
    "gh_weights_refresh": [_I, _P, _P, _P, _P, _P, _P, _P],
    "gh_seg_offsets": [_P, _I, _P, _P, _I, _P, _P],
    "gh_seg_broadcast": [_P, _P, _P, _I, _I, _P],
    "gh_seg_sum": [_P, _P, _P, _I, _I, _P],
    "gh_seg_pad": [_P, _P, _P, _I, _I, _I, _I, _P],
    "gh_seg_unpad": [_P, _P, _P, _I, _I, _I, _I, _P],
    "gh_masked_mean_fwd": [_P, _P, _P, _P, _I, _I, _I, _P],
    "gh_masked_mean_bwd": [_P, _P, _P, _P, _I, _I, _I, _P],
    "gh_adam_step": [_P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _I, _F, _P],
    "gh_set_workspace": [_P, _L],
    "gh_set_stream_workspace": [_P, _P, _L],
    "gh_profile_enable": [_I],
    "gh_profile_select": [_U],
    "gh_profile_collect": [_P, _I],
    "gh_gemm_path_counters": [_P, _I],
    "gh_set_gemm_mode": [_I],
    "gh_weights_changed": [],
    "gh_fp32x3_refresh": [_P],
    "gh_fp32x3_clear": [],
    # composite entry points (get_amd/fused.py holds the ctypes mirrors of the descriptor structs)
    "gh_get_plan_buffers": [_P, _P, _P],
    "gh_get_forward": [_P, _P, _P, _P, _P, _P],
    "gh_get_backward": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P],
    "gh_cross_entropy": [_P, _P, _I, _I, _P, _P, _P],
    "gh_get_prepare": [_P, _P, _I, _I, _P, _P, _I, _I, _I] + [_P] * 8 + [_I] + [_P] * 5 + [_P, _P, _P],
    "gh_get_struct_sizes": [_P],
    # library-owned RCCL communicator (csrc/comm_ops.hip; librccl is dlopen'ed on first use)
    "gh_comm_unique_id": [_P],
    "gh_comm_init": [_P, _I, _I, _P],
    "gh_comm_destroy": [_P],
    "gh_comm_info": [_P, _P, _P],
    "gh_flat_allreduce": [_P, _P, _L, _P],
    "gh_flat_broadcast": [_P, _P, _L, _I, _P],
}

PROFILE_ROWS = ["gemm_big", "gemm_big_tn", "gemm_small", "gemm_small_tn", "spmm", "scorer_gsl",
                "graph_build", "att_softmax_fwd", "att_softmax_bwd", "att_dpre", "gate_bwd_pre", "colsum", "adam",
                "few_row_streams"]


def profile_enable(on: bool, only=None):
    """Turn the per-launch HIP-event timing on/off; `only` = iterable of PROFILE_ROWS names to instrument (default all)."""
    mask = 0xFFFFFFFF if only is None else sum(1 << PROFILE_ROWS.index(n) for n in only)
    call("gh_profile_select", mask)
    call("gh_profile_enable", 1 if on else 0)


def profile_collect() -> dict:
    """{kernel: {"ms": total, "work": total algorithmic flops|bytes, "launches": n}} since the last collect."""
    buf = (ctypes.c_double * (len(PROFILE_ROWS) * 3))()
    call("gh_profile_collect", ctypes.cast(buf, ctypes.c_void_p), len(PROFILE_ROWS))
    return {name: {"ms": buf[3 * i], "work": buf[3 * i + 1], "launches": int(buf[3 * i + 2])}
            for i, name in enumerate(PROFILE_ROWS)}

_GEMM_MODE = "fp32"


def set_gemm_mode(mode: str):
    """"fp32" (default, exact fp32 MFMA everywhere); "bf16" (BASELINE configs[4]): the evidence cells run the bf16
    STORAGE pipeline (bf16 activations/weights in HBM, v_mfma_f32_16x16x32_bf16, fp32 accumulate) where their shape
    allows it, every other activation-sized GEMM rounds its operands to bf16 in registers; "fp32x3" (opt-in): fp32
    storage and fp32 results, every product formed on the bf16 MFMA from 3-way bf16 splits of both operands (six of the
    nine cross terms, error below one fp32 rounding of the product; csrc/gemm_nt.hip.h MODE 3)."""
    global _GEMM_MODE
    call("gh_set_gemm_mode", {"fp32": 0, "bf16": 1, "fp32x3": 2, "fp32x3p": 3}[mode])
    _GEMM_MODE = mode


def gemm_mode() -> str:
    return _GEMM_MODE


def gemm_path_counters(reset: bool = False) -> dict:
    """{"fast", "generic", "generic_large"} GEMM launch counts since the last reset (see include/get_hip.h)."""
    buf = (ctypes.c_int64 * 3)()
    call("gh_gemm_path_counters", ctypes.cast(buf, ctypes.c_void_p), 1 if reset else 0)
    return {"fast": int(buf[0]), "generic": int(buf[1]), "generic_large": int(buf[2])}


_lib = None


def load():
    """Load the shared library once; raise loudly if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"get_amd: HIP library not found at {LIB_PATH}. Build it with "
            "`python -c \"import __graft_entry__ as g; g.build()\"` or `make -C get_amd/csrc`. "
            "There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    lib.gh_abi_version.restype = _I
    lib.gh_last_error.restype = ctypes.c_char_p
    lib.gh_comm_library.restype = ctypes.c_char_p
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.argtypes = args
        fn.restype = _I
    if lib.gh_abi_version() != ABI_VERSION:
        raise RuntimeError(f"get_amd: ABI version mismatch ({lib.gh_abi_version()} != {ABI_VERSION})")
    _lib = lib
    return lib


_workspaces = {}


def has_workspace(device, raw_stream) -> bool:
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else _get_device()
    return (idx, raw_stream) in _workspaces


def ensure_workspace(device, nbytes: int = 256 << 20):
    """Register (once per device and stream) the split-K scratch buffer of the weight-gradient GEMMs for the CURRENT
    stream of `device`: two streams or two devices in one process never share a scratch area."""
    dev = torch.device(device)
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    st = _get_raw_stream(dev.index) if _get_raw_stream is not None else torch.cuda.current_stream(dev).cuda_stream
    key = (dev.index, st)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() * 4 < nbytes:
        ws = torch.empty(nbytes // 4, device=dev, dtype=torch.float32)
        _workspaces[key] = ws
        # the C registry is keyed by (hipGetDevice(), stream): register under `dev`, whatever the current device is
        if dev.index != _get_device():
            with torch.cuda.device(dev):
                call("gh_set_stream_workspace", st, ws.data_ptr(), ws.numel() * 4)
        else:
            call("gh_set_stream_workspace", st, ws.data_ptr(), ws.numel() * 4)
    return ws


# The host side issues ~120 launches per training step; at realistic evidence counts (~220 pairs per step) the step is
# bound by that issue rate, so the per-call helpers avoid torch.cuda's Python wrappers (current_device() 0.75 us,
# current_stream().cuda_stream 10 us per call) in favour of the raw bindings underneath them.
_get_device = torch._C._cuda_getDevice
_get_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  The tensor must be contiguous and live on the CURRENT device: the
    kernels are launched on the current device's current stream (wrap multi-device use in torch.cuda.device(...))."""
    if t is None:
        return None
    assert t.is_contiguous(), "get_amd: non-contiguous tensor handed to the C-ABI"
    if t.is_cuda and t.device.index != _get_device():
        raise RuntimeError(f"get_amd: tensor on cuda:{t.device.index} but the current device is cuda:{torch.cuda.current_device()}; "
                           "launches go to the current device's stream -- use `with torch.cuda.device(tensor.device):`")
    return t.data_ptr()


def stream():
    """Raw handle of the current device's current HIP stream."""
    if _get_raw_stream is not None:
        return _get_raw_stream(_get_device())
    return torch.cuda.current_stream().cuda_stream


_fns: dict = {}


def call(name: str, *args):
    fn = _fns.get(name)
    if fn is None:
        fn = _fns[name] = getattr(load(), name)
    rc = fn(*args)
    if rc != 0:
        raise RuntimeError(f"get_amd: {name} failed (rc={rc}): {load().gh_last_error().decode(errors='replace')}")


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("get_amd: tensors must live on a ROCm device (cuda:N); there is no CPU path")
