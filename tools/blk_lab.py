#!/usr/bin/env python3
"""HGATE fp32 block attention through the LAB library: start skew of the workgroups that share a CU (HWGAT_BLK_SKEW, in units
of 64 cycles per resident index), old 32x32-tile kernels (HWGAT_BLK_F32=0) for reference."""
import ctypes, importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
hw = importlib.import_module("sl-hwgat_amd")
HF, L = hw.functional, hw._lib
from oracle import hgat_oracle as OH
lab = ctypes.CDLL(os.path.join(ROOT, "sl-hwgat_amd", "libhwgat_hip_lab.so"))
assert lab.hwgat_is_lab_build() == 1, "not the LAB library: build it with `python sl-hwgat_amd/build.py --lab`"
for name, args in L._SIGS.items():
    fn = getattr(lab, name)
    fn.argtypes, fn.restype = args, (ctypes.c_int64 if name.endswith("_bytes") else ctypes.c_int)
L.lib()
L._lib = lab
dev = "cuda:0"
B, F, K, nH, d = 64, 128, 29, 2, 128
qkv = torch.randn(B, F, K, 3 * d, device=dev)
do = torch.randn(B, F, K, d, device=dev)
o, dq = torch.empty_like(do), torch.empty_like(qkv)
bits = HF.blk_mask_bits(OH.block_adjacency(), K).to(dev)
E = B * F * K * d * 4


def timed(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


fwd = lambda: HF.attn_fwd("blk", qkv, o, bits, None, nH, False)
bwd = lambda: HF.attn_bwd("blk", qkv, do, dq, bits, None, nH, False)
for occ in (2, 3):
    os.environ["HWGAT_BLK_OCC"] = str(occ)
    for skew in [int(a) for a in (sys.argv[1:] or "0 128 256".split())]:
        os.environ["HWGAT_BLK_SKEW"] = str(skew)
        tf, tb = timed(fwd), timed(bwd)
        print(f"fwd workgroups per CU {occ} skew/flags {skew:#8x}: fwd {tf:7.1f} us {4 * E / tf / 1e6:5.2f} TB/s | bwd {tb:7.1f} us {7 * E / tb / 1e6:5.2f} TB/s", flush=True)
