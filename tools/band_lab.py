#!/usr/bin/env python3
"""sweep the frame-segment counts of the bf16 band-attention kernels (LAB library: HWGAT_BAND_FSEG / HWGAT_BAND_BSEG)"""
import ctypes, importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hw = importlib.import_module("sl-hwgat_amd")
HF, L = hw.functional, hw._lib
from oracle import wgat_oracle as OW
lab = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sl-hwgat_amd", "libhwgat_hip_lab.so"))
assert lab.hwgat_is_lab_build() == 1, "not the LAB library: build it with `python sl-hwgat_amd/build.py --lab`"
for name, args in L._SIGS.items():
    fn = getattr(lab, name)
    fn.argtypes, fn.restype = args, ctypes.c_int
L.lib()
L._lib = lab
B, F, nW, nH, hd = 64, 128, 4, 8, 16
if len(sys.argv) > 5:
    B, F, nW, nH, hd = [int(a) for a in sys.argv[1:6]]
d, K = nH * hd, nW * 16
dev = "cuda:0"
qkv = torch.randn(B, F, K, 3 * d, device=dev).to(torch.bfloat16)
do = torch.randn(B, F, K, d, device=dev).to(torch.bfloat16)
o, dq = torch.empty_like(do), torch.empty_like(qkv)
rows = HF.band_mask_rows(OW.band_adjacency(F, nW), F).to(dev)
E = B * F * K * d * 2


def t(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for var, fn, mult in (("HWGAT_BAND_FSEG", lambda: HF.attn_fwd("band", qkv, o, rows, None, nH, False), 4),
                      ("HWGAT_BAND_BSEG", lambda: HF.attn_bwd("band", qkv, do, dq, rows, None, nH, False), 7)):
    for n in (1, 2, 4, 8):
        os.environ[var] = str(n)
        us = t(fn)
        print(f"{var}={n}: {us:7.1f} us  {mult * E / us / 1e6:5.2f} TB/s  {mult * E / us / 8e6:5.3f} of 8 TB/s", flush=True)
    del os.environ[var]

for dbg, what in ((0, "shipped: workgroup-staged tiles"), (5, "staged, memory only"), (6, "staged, groups of 2 frames"), (7, "one wave fetches its own head"),
                  (1, "own head, memory only"), (3, "own head, no loads in the loop"), (4, "memory probe: whole 128-byte lines")):
    os.environ["HWGAT_BAND_DBG"] = str(dbg)
    us = t(lambda: HF.attn_fwd("band", qkv, o, rows, None, nH, False))
    print(f"fwd DBG={dbg} ({what}): {us:7.1f} us", flush=True)
del os.environ["HWGAT_BAND_DBG"]

for dbg, what in ((0, "shipped: staged, groups of 4 frames"), (6, "staged, groups of 2 frames"), (7, "one wave fetches its own head")):
    os.environ["HWGAT_BAND_DBG"] = str(dbg)
    us = t(lambda: HF.attn_bwd("band", qkv, do, dq, rows, None, nH, False))
    print(f"bwd DBG={dbg} ({what}): {us:7.1f} us", flush=True)
del os.environ["HWGAT_BAND_DBG"]
