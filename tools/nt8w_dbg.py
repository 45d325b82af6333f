#!/usr/bin/env python3
"""What each part of the eight-wave bf16 NT schedule costs: the plain launch at three K depths with parts of the kernel
compiled out (LAB library, HWGAT_NT8W_DBG read per call): 0 = the real kernel, 1 = no epilogue stores, 2 = no DMA waits,
3 = no DMA, 4 = no DMA and no fragment reads (MFMA clusters + barriers only).  Results of 1-4 are wrong by design.
time = tiles_per_block * (phases * t_phase + t_tile): the K sweep separates the two."""
import ctypes
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
hw = importlib.import_module("sl-hwgat_amd")
HF, L = hw.functional, hw._lib
lab = ctypes.CDLL(os.path.join(ROOT, "sl-hwgat_amd", "libhwgat_hip_lab.so"))
assert lab.hwgat_is_lab_build() == 1, "not the LAB library: build it with `python sl-hwgat_amd/build.py --lab`"
for name, args in L._SIGS.items():
    fn = getattr(lab, name)
    fn.argtypes, fn.restype = args, ctypes.c_int
L.lib()
L._lib = lab
dev = "cuda:0"
M = int(os.environ.get("NT8W_M", 163840))
N = int(os.environ.get("NT8W_N", 512))
g = torch.Generator(device=dev).manual_seed(0)
print(f"M={M} N={N}: tiles/block = {M // 256 * (N // 256) / 256:.2f}")
rows = {}
for K in (128, 512, 1024, 1536):
    A = torch.randn(M, K, device=dev, generator=g).bfloat16()
    W = (torch.randn(N, K, device=dev, generator=g) * 0.05).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for dbg in (0, 1, 2, 3, 4):
        os.environ["HWGAT_NT8W_DBG"] = str(dbg)
        ts = []
        for r in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            HF.linear_nt(A, W, None, epi=HF.EPI_NONE, out=out)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        rows[(K, dbg)] = ts[len(ts) // 2]
        print(f"K={K:5d} dbg={dbg}: median {ts[len(ts) // 2]:8.1f} us  min {ts[0]:8.1f} us   TF(real flops) {2.0 * M * N * K / ts[len(ts) // 2] / 1e6:7.0f}", flush=True)
tiles = M // 256 * (N // 256) / 256
for dbg in (0, 1, 2, 3, 4):
    t_phase = (rows[(1536, dbg)] - rows[(512, dbg)]) / tiles / 64          # 16 K-tiles = 64 phases apart
    t_tile = rows[(512, dbg)] / tiles - 32 * t_phase
    print(f"dbg={dbg}: t_phase = {t_phase * 1e3:6.1f} ns (ideal 2 x 256 cycles = {512 / 2.1:.0f} ns at 2.1 GHz), per-tile overhead = {t_tile:6.2f} us")
