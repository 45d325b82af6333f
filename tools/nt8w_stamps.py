#!/usr/bin/env python3
"""Where a tile's time goes in the eight-wave bf16 NT kernel: s_memtime stamps (LAB library, HWGAT_NT8W_DBG=5) of waves 0
and 4 of every block: 0 tile start (accumulators zeroed), 1 main loop entered, 2 main loop done, 3 epilogue operands loaded,
4 epilogue done.  s_memtime ticks at 100 MHz-independent shader-clock rate (cycles)."""
import ctypes, importlib, os, sys
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
M, N = 163840, 512
g = torch.Generator(device=dev).manual_seed(0)
for K in (128, 512, 1536):
    A = torch.randn(M, K, device=dev, generator=g).bfloat16()
    W = (torch.randn(N, K, device=dev, generator=g) * 0.05).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    st = torch.zeros(256 * 8 * 2 * 6, device=dev, dtype=torch.int64)
    os.environ["HWGAT_NT8W_DBG"] = "5"
    for _ in range(3):
        L.call("hwgat_linear_nt_bf16", L.ptr(A), L.ptr(W), None, L.ptr(out), M, N, K, 0, None, None, None, None, 0, 0.0, 4, None,
               L.ptr(st), None, 0, 0.0, None, L.stream())
    torch.cuda.synchronize()
    s = st.view(256, 8, 2, 6).cpu().double()
    tiles = 5
    print(f"K={K}: cycles (median over 256 blocks), per tile slot; columns: zero->loop | main loop | realign+operands | epilogue body | tile total | gap to next tile start")
    for gmi in (0, 1):
        for ts in range(tiles):
            r = s[:, ts, gmi]
            d01 = (r[:, 1] - r[:, 0]).median().item()
            d12 = (r[:, 2] - r[:, 1]).median().item()
            d23 = (r[:, 3] - r[:, 2]).median().item()
            d34 = (r[:, 4] - r[:, 3]).median().item()
            tot = (r[:, 4] - r[:, 0]).median().item()
            gap = (s[:, ts + 1, gmi, 0] - r[:, 4]).median().item() if ts + 1 < tiles else float("nan")
            print(f"  group {gmi} tile {ts}: {d01:8.0f} | {d12:8.0f} | {d23:8.0f} | {d34:8.0f} | {tot:8.0f} | {gap:8.0f}")
    t0 = s[:, 0, 0, 0]
    print(f"  block start skew: min {0:.0f} max {(t0.max() - t0.min()).item():.0f} cycles; last tile end spread {(s[:, tiles - 1, 0, 4].max() - s[:, tiles - 1, 0, 4].min()).item():.0f}")
    print(f"  whole block (first stamp -> last): median {(s[:, tiles - 1, 0, 4] - s[:, 0, 0, 0]).median().item():.0f} cycles")
