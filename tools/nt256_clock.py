#!/usr/bin/env python3
"""The shader clock the chip holds under the fp32 256-tile NT linear kernel (LAB library): s_memtime at the start and the end of
every persistent workgroup against the launch's wall time (HIP events), for plain launches of the three stages and for a
register-only MFMA loop (tools/mfma_peak.py measures 155 TFLOP/s with that).  What `peak` of the fp32-MFMA roofline is worth
under a real GEMM's LDS / HBM traffic."""
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
    fn.argtypes, fn.restype = args, (ctypes.c_int64 if name.endswith("_bytes") else ctypes.c_int)
L.lib()
L._lib = lab
lab.hwgat_lab_nt256_stamps.argtypes, lab.hwgat_lab_nt256_stamps.restype = [ctypes.c_void_p], ctypes.c_int
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
st = torch.zeros(256 * 2, device=dev, dtype=torch.int64)
print("fp32 256-tile NT kernel, plain epilogue: shape, wall time, TFLOP/s, s_memtime ticks per workgroup (median), ticks / wall = clock")
for M, N, K in ((163840, 512, 1536), (163840, 512, 512), (327680, 256, 768), (327680, 512, 256), (655360, 256, 128)):
    A = torch.randn(M, K, device=dev, generator=g)
    W = torch.randn(N, K, device=dev, generator=g) * 0.05
    out = torch.empty(M, N, device=dev)
    run = lambda: HF.linear_nt(A, W, None, epi=HF.EPI_NONE, out=out)
    for _ in range(20):
        run()
    torch.cuda.synchronize()
    st.zero_()
    assert lab.hwgat_lab_nt256_stamps(st.data_ptr()) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    run()
    e1.record()
    torch.cuda.synchronize()
    lab.hwgat_lab_nt256_stamps(None)
    us = e0.elapsed_time(e1) * 1e3
    s = st.view(256, 2).cpu().double()
    live = s[:, 1] > 0
    ticks = (s[live, 1] - s[live, 0]).median().item()            # (the counters of different XCDs are not aligned: per-workgroup spans only)
    ghz = ticks / us / 1e3                                       # a workgroup lives for (almost) the whole launch: a lower bound
    print(f"M={M} N={N} K={K}: {us:8.1f} us {2.0 * M * N * K / us / 1e6:6.1f} TF | {int(live.sum())} workgroups, median {ticks:.0f} ticks each "
          f"-> >= {ghz:.2f} GHz; 157.3 TF at 2.4 GHz = {157.3 * ghz / 2.4:.1f} TF at this clock", flush=True)
