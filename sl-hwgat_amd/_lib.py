"""ctypes binding of libhwgat_hip.so (the C-ABI declared in include/hwgat_hip.h).

There is deliberately NO fallback: if the HIP library is missing or a launcher
returns an error, this raises.  torch is used only to obtain device pointers
and the current HIP stream.
"""
import ctypes
import os
import re

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libhwgat_hip.so")
HEADER = os.path.join(os.path.dirname(_HERE), "include", "hwgat_hip.h")

F32, BF16 = 0, 1
_ERR = {-1: "HWGAT_EINVAL (null pointer / bad size)", -2: "HWGAT_ESHAPE (unsupported shape)",
        -3: "HWGAT_EDTYPE (unknown dtype)"}

_P, _I, _L = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
_U, _F = ctypes.c_uint32, ctypes.c_float
_SIGS = {
    "hwgat_abi_version": [],
    "hwgat_is_lab_build": [],
    "hwgat_seed_advance": [_P, _P],
    "hwgat_ln_bwd_det_bytes": [_I],
    "hwgat_ln_bwd_det": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _P, _U, _F, _P, _P, _P, _L, _P],
    "hwgat_linear_tn_det_bytes": [_L, _I, _I],
    "hwgat_linear_tn_f32_det": [_P, _P, _P, _P, _L, _I, _I, _U, _F, _P, _P, _P, _P, _P, _P, _L, _P],
    "hwgat_linear_tn_bf16_det": [_P, _P, _P, _P, _L, _I, _I, _U, _F, _P, _P, _P, _P, _P, _P, _L, _P],
    "hwgat_seed_set": [_P, _U, _U, _U, _P],
    "hwgat_debug_mfma32x32x2": [_P, _P, _P, _P],
    "hwgat_debug_mfma_peak": [_P, _I, _I, _I, _P],
    "hwgat_embed_fwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _U, _F, _P, _P],
    "hwgat_ln_fwd": [_P, _P, _P, _P, _P, _P, _L, _I, _I, _P],
    "hwgat_ln_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _P],
    "hwgat_ln_bwd_masked": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _P, _U, _F, _P, _P],
    "hwgat_ln_bwd_xn": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _P, _U, _F, _P, _P, _P],
    "hwgat_win_attn_fwd": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "hwgat_win_attn_bwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "hwgat_weight_prep": [_P, _I, _I, _I, _P],
    "hwgat_win_attn_fwd_drop": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _U, _F, _P, _P],
    "hwgat_win_attn_bwd_drop": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _U, _F, _P, _P],
    "hwgat_blk_attn_fwd": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "hwgat_blk_attn_bwd": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "hwgat_band_attn_fwd": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "hwgat_blk_attn_fwd_drop": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _U, _F, _P, _P],
    "hwgat_blk_attn_bwd_drop": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _U, _F, _P, _P],
    "hwgat_band_attn_fwd_drop": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _U, _F, _P, _P],
    "hwgat_band_attn_bwd_drop": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _U, _F, _P, _P],
    "hwgat_band_attn_bwd": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "hwgat_debug_mfma16x16x4": [_P, _P, _P, _P],
    "hwgat_lnpool_fwd": [_P, _P, _P, _P, _I, _I, _I, _I, _P],
    "hwgat_lnpool_partial_rows": [_I, _I],
    "hwgat_lnpool_fwd_det": [_P, _P, _P, _P, _I, _I, _I, _I, _P, _P],
    "hwgat_lnpool_bwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "hwgat_lnpool_bwd_masked": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _U, _F, _P, _P],
    "hwgat_unmerge_masked": [_P, _P, _P, _I, _I, _I, _I, _I, _U, _F, _P, _P],
    "hwgat_merge": [_P, _P, _I, _I, _I, _I, _I, _I, _P],
    "hwgat_linear_nt_f32": [_P, _P, _P, _P, _L, _I, _I, _I, _P, _P, _P, _P, _U, _F, _I, _P, _P, _P, _U, _F, _P, _P],
    "hwgat_linear_nt_f32_ex": [_P, _P, _P, _P, _L, _I, _I, _I, _P, _P, _P, _P, _U, _F, _I, _P, _P, _P, _U, _F, _P, _P, _I, _I, _P, _P],
    "hwgat_linear_nt_bf16_ex": [_P, _P, _P, _P, _L, _I, _I, _I, _P, _P, _P, _P, _U, _F, _I, _P, _P, _P, _U, _F, _P, _P, _I, _I, _P, _P],
    "hwgat_ln_finalize": [_P, _P, _L, _I, _P],
    "hwgat_ln_fold": [_P, _P, _P, _P, _I, _I, _P, _P, _P, _I, _P],
    "hwgat_linear_tn_f32": [_P, _P, _P, _P, _L, _I, _I, _U, _F, _P, _P, _P, _P, _P, _P],
    "hwgat_linear_nt_bf16": [_P, _P, _P, _P, _L, _I, _I, _I, _P, _P, _P, _P, _U, _F, _I, _P, _P, _P, _U, _F, _P, _P],
    "hwgat_linear_tn_bf16": [_P, _P, _P, _P, _L, _I, _I, _U, _F, _P, _P, _P, _P, _P, _P],
    "hwgat_linear_tn_f32_ws_bytes": [_L, _I, _I],
    "hwgat_linear_tn_f32_ws": [_P, _P, _P, _P, _L, _I, _I, _U, _F, _P, _P, _P, _P, _P, _L, _P, _P],
    "hwgat_linear_tn_bf16_ws_bytes": [_L, _I, _I],
    "hwgat_linear_tn_bf16_ws": [_P, _P, _P, _P, _L, _I, _I, _P, _L, _P],
    "hwgat_transpose_f32": [_P, _P, _I, _I, _P],
    "hwgat_dropout_mask_f32": [_P, _L, _U, _F, _P, _P],
}
_lib = None


def declared_symbols():
    """every function name declared in include/hwgat_hip.h"""
    with open(HEADER) as fh:
        src = re.sub(r"/\*.*?\*/", "", fh.read(), flags=re.S)
    return sorted(set(re.findall(r"\b(?:int|int64_t)\s+(hwgat_\w+)\s*\(", src)))


def header_abi_version():
    """HWGAT_ABI_VERSION of include/hwgat_hip.h, the header these bindings were written against"""
    with open(HEADER) as fh:
        return int(re.search(r"#define\s+HWGAT_ABI_VERSION\s+(\d+)", fh.read()).group(1))


def _check_stamp():
    """refuse a library that was built from other sources than the ones in the tree (a stale .so would otherwise be
    tested and benchmarked silently): build.py writes the digest of csrc/ + the header + the flags next to the .so"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("_hwgat_build", os.path.join(_HERE, "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    stamp = LIB_PATH + ".stamp"
    have = open(stamp).read().strip() if os.path.exists(stamp) else None
    if have != mod._digest():
        raise RuntimeError(
            f"{LIB_PATH} does not match the sources under sl-hwgat_amd/csrc (digest stamp "
            f"{'missing' if have is None else 'differs'}): rebuild with `python sl-hwgat_amd/build.py`")


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HWGAT HIP backend is not built. Run "
                "`python sl-hwgat_amd/build.py` (hipcc --offload-arch=gfx950). "
                "There is no CPU fallback.")
        _check_stamp()
        handle = ctypes.CDLL(LIB_PATH)
        for name, args in _SIGS.items():
            fn = getattr(handle, name)
            fn.argtypes = args
            fn.restype = _L if name.endswith("_bytes") else _I
        have, want = handle.hwgat_abi_version(), header_abi_version()
        if have != want:
            raise RuntimeError(f"{LIB_PATH} reports ABI {have}, include/hwgat_hip.h declares {want}: rebuild the library")
        _lib = handle
    return _lib


def dtype_code(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise TypeError(f"unsupported activation dtype {t.dtype}")


def ptr(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("HWGAT HIP kernels need tensors on an MI355X device (got a CPU tensor); "
                           "there is no CPU fallback")
    if not t.is_contiguous():
        raise ValueError("HWGAT HIP kernels need contiguous tensors")
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def call(name, *args):
    rc = getattr(lib(), name)(*args)
    if rc != 0:
        raise RuntimeError(f"{name} failed: {_ERR.get(rc, 'hipError ' + str(rc))}")
