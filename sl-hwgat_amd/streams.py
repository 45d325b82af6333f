"""HIP streams restricted to a subset of the compute units (hipExtStreamCreateWithCUMask), wrapped for torch.

Used only by the opt-in overlap of the weight-gradient GEMMs with the critical path (block.OVERLAP_DW): the dW / db
launches of a block are off the backward critical path, and ~10 % of the fp32 step is HBM-bound kernels (attention
backward, LayerNorm backward) that leave the matrix pipes idle.  A side stream that owns k CUs of every XCD can run dW
there while the main stream keeps the rest.  Measured in round 3 (profiles/r03_cu_split_sweep.json); off by default.

MI355X has 256 CUs in 8 XCDs; bit i of the mask is CU (i // 8) of XCD (i % 8), so a mask of the low 8k bits is k CUs on
each XCD."""
import ctypes

import torch

_hip = None


def _lib():
    global _hip
    if _hip is None:
        _hip = ctypes.CDLL("libamdhip64.so")
        _hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32,
                                                      ctypes.POINTER(ctypes.c_uint32)]
        _hip.hipExtStreamCreateWithCUMask.restype = ctypes.c_int
    return _hip


def cu_mask(lo: int, hi: int, total: int = 256):
    """words of a mask with CU bits [lo, hi) set"""
    words = [0] * ((total + 31) // 32)
    for i in range(lo, hi):
        words[i // 32] |= 1 << (i % 32)
    return words


def masked_stream(device, words):
    """a torch.cuda.ExternalStream on `device` whose kernels may only use the CUs of `words` (list of uint32)"""
    with torch.cuda.device(device):
        torch.cuda.current_stream()                 # the context exists
        handle = ctypes.c_void_p()
        arr = (ctypes.c_uint32 * len(words))(*words)
        rc = _lib().hipExtStreamCreateWithCUMask(ctypes.byref(handle), len(words), arr)
        if rc != 0:
            raise RuntimeError(f"hipExtStreamCreateWithCUMask failed: hipError {rc}")
        return torch.cuda.ExternalStream(handle.value, device=device)
