"""EXPERIMENT, outside the product: fp32 NT GEMM from nine exact bf16 partial products.

Builds tools/x9/libx9.so from tools/x9/gemm_f32x9.hip (it only borrows csrc/common.h for the vector typedefs) and
binds its two entry points.  Nothing under sl-hwgat_amd/ knows about this file; libhwgat_hip.so does not contain it.
`python tools/x9/x9lib.py` builds and, on a GPU box, runs the exact-split / accuracy check."""
import ctypes
import os
import subprocess
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
LIB = os.path.join(HERE, "libx9.so")
SRC = os.path.join(HERE, "gemm_f32x9.hip")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
            subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                            "-Wno-pass-failed", "-I", os.path.join(ROOT, "sl-hwgat_amd", "csrc"), "-o", LIB, SRC],
                           check=True)
        _lib = ctypes.CDLL(LIB)
        P, L, I = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
        _lib.hwgat_split3_bf16.argtypes = [P, P, L, P]
        _lib.hwgat_linear_nt_f32x9.argtypes = [P, P, P, L, I, I, P]
    return _lib


def _check(rc, what):
    if rc:
        raise RuntimeError(f"{what} failed: {rc}")


def split3(W):
    """fp32 tensor -> (3, *W.shape) bf16 planes with W == p0 + p1 + p2 exactly (8 significand bits each)"""
    out = torch.empty(3, *W.shape, device=W.device, dtype=torch.bfloat16)
    _check(lib().hwgat_split3_bf16(W.data_ptr(), out.data_ptr(), W.numel(), torch.cuda.current_stream().cuda_stream), "split3")
    return out


def linear_nt_x9(A, W3, out=None):
    K = A.shape[-1]
    M = A.numel() // K
    N = W3.shape[1]
    C = out if out is not None else torch.empty(*A.shape[:-1], N, device=A.device, dtype=torch.float32)
    _check(lib().hwgat_linear_nt_f32x9(A.data_ptr(), W3.data_ptr(), C.data_ptr(), M, N, K,
                                       torch.cuda.current_stream().cuda_stream), "linear_nt_f32x9")
    return C


def check():
    """the former tests/test_gpu_gemm.py::test_x9_...: exact 3-way split, fp32-grade accuracy"""
    import importlib
    sys.path.insert(0, ROOT)
    HF = importlib.import_module("sl-hwgat_amd").functional
    for M, N, K in [(128, 128, 32), (640, 384, 128), (1152, 512, 1536), (1280, 512, 1024), (256, 256, 32)]:
        g = torch.Generator().manual_seed(11 + M)
        A = torch.randn(M, K, generator=g) * torch.exp(3 * torch.randn(M, 1, generator=g))
        W = torch.randn(N, K, generator=g) * 0.1
        Wd = W.cuda()
        W3 = split3(Wd)
        assert torch.equal((W3[0].float() + (W3[1].float() + W3[2].float())).cpu(), W)
        ref = A.double() @ W.double().t()
        e9 = ((linear_nt_x9(A.cuda(), W3).cpu().double() - ref).norm() / ref.norm()).item()
        e32 = ((HF.linear_nt(A.cuda(), Wd, None, epi=HF.EPI_NONE).cpu().double() - ref).norm() / ref.norm()).item()
        print(f"M={M} N={N} K={K}: x9 rel err {e9:.2e}, fp32-MFMA kernel {e32:.2e}")
        assert e9 < 2e-5 and e9 < 2 * e32 + 1e-8


if __name__ == "__main__":
    lib()
    print("built", LIB)
    if torch.cuda.is_available():
        check()
