#!/usr/bin/env python3
"""WGATE fp32 band attention through the LAB library: frames per staged group (HWGAT_BAND_PF), frame segments per clip
(HWGAT_BAND_FSEG / HWGAT_BAND_BSEG), and the one-wave-per-head kernels of band_attn.hip (HWGAT_BAND_F32=0) for reference."""
import ctypes, importlib, os, subprocess, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "old":                   # (the switch is read once per process)
    os.environ["HWGAT_BAND_F32"] = "0"
hw = importlib.import_module("sl-hwgat_amd")
HF, L = hw.functional, hw._lib
from oracle import wgat_oracle as OW
lab = ctypes.CDLL(os.path.join(ROOT, "sl-hwgat_amd", "libhwgat_hip_lab.so"))
assert lab.hwgat_is_lab_build() == 1, "not the LAB library: build it with `python sl-hwgat_amd/build.py --lab`"
for name, args in L._SIGS.items():
    fn = getattr(lab, name)
    fn.argtypes, fn.restype = args, (ctypes.c_int64 if name.endswith("_bytes") else ctypes.c_int)
L.lib()
L._lib = lab
dev = "cuda:0"
B, F, nW, nH, hd = 64, 128, 4, 8, 16
d, K = nH * hd, nW * 16
qkv = torch.randn(B, F, K, 3 * d, device=dev)
do = torch.randn(B, F, K, d, device=dev)
o, dq = torch.empty_like(do), torch.empty_like(qkv)
rows = HF.band_mask_rows(OW.band_adjacency(F, nW), F).to(dev)
E = B * F * K * d * 4


def timed(fn, n=100):
    for _ in range(20):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


fwd = lambda: HF.attn_fwd("band", qkv, o, rows, None, nH, False)
bwd = lambda: HF.attn_bwd("band", qkv, do, dq, rows, None, nH, False)
if os.environ.get("HWGAT_BAND_F32") == "0":
    tf, tb = timed(fwd), timed(bwd)
    print(f"one wave per head (band_attn.hip): fwd {tf:7.1f} us {4 * E / tf / 1e6:5.2f} TB/s | bwd {tb:7.1f} us {7 * E / tb / 1e6:5.2f} TB/s")
    sys.exit(0)
for nhw, pf, seg in ((4, 2, 1), (4, 1, 1), (8, 1, 1), (8, 2, 1), (8, 1, 2), (4, 2, 1), (4, 1, 1), (8, 1, 1), (8, 2, 1), (8, 1, 2)):
    os.environ["HWGAT_BAND_NHW"] = str(nhw)                      # (forward only: the backward pass keeps 4 heads per workgroup)
    os.environ["HWGAT_BAND_PF"] = str(pf)
    os.environ["HWGAT_BAND_FSEG"] = os.environ["HWGAT_BAND_BSEG"] = str(seg)
    tf = timed(fwd)
    os.environ["HWGAT_BAND_PF"] = str(min(pf, 2))
    tb = timed(bwd)
    print(f"heads per workgroup {nhw} frames per group {pf} segments {seg}: fwd {tf:7.1f} us {4 * E / tf / 1e6:5.2f} TB/s | bwd (4 heads) {tb:7.1f} us {7 * E / tb / 1e6:5.2f} TB/s", flush=True)
subprocess.run([sys.executable, os.path.abspath(__file__), "old"], check=False)
