#!/usr/bin/env python3
"""EXPERIMENT: time the nine-bf16-product fp32 NT GEMM against the fp32-MFMA kernel on the dX shapes"""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import x9lib
hw = importlib.import_module("sl-hwgat_amd")
HF = hw.functional
dev = "cuda:0"
for M, N, K in ((163840, 512, 1536), (163840, 512, 1024), (163840, 512, 512), (327680, 256, 768), (655360, 128, 384)):
    A = torch.randn(M, K, device=dev)
    W = torch.randn(N, K, device=dev) * 0.05
    W3 = x9lib.split3(W)
    C = torch.empty(M, N, device=dev)
    ref = (A[:512].double() @ W.double().t())
    res = {}
    for name, fn in (("f32 mfma", lambda: HF.linear_nt(A, W, None, epi=HF.EPI_NONE, out=C)), ("bf16 x9", lambda: x9lib.linear_nt_x9(A, W3, out=C))):
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 10 * 1e3
        err = ((C[:512].double() - ref).norm() / ref.norm()).item()
        res[name] = us
        print(f"M={M} N={N} K={K} {name}: {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s-equivalent  rel err {err:.2e}")
    print(f"   speed-up {res['f32 mfma'] / res['bf16 x9']:.2f}x")
