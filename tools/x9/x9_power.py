#!/usr/bin/env python3
"""is the x9 kernel power-limited?  same launch with random vs constant vs zero operands"""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import x9lib
hw = importlib.import_module("sl-hwgat_amd")
HF = hw.functional
dev = "cuda:0"
M, N, K = 163840, 512, 1024
C = torch.empty(M, N, device=dev)
for label, A, W in (("random", torch.randn(M, K, device=dev), torch.randn(N, K, device=dev) * 0.05),
                    ("ones", torch.ones(M, K, device=dev), torch.ones(N, K, device=dev)),
                    ("zeros", torch.zeros(M, K, device=dev), torch.zeros(N, K, device=dev))):
    W3 = x9lib.split3(W)
    for name, fn in (("f32 mfma", lambda: HF.linear_nt(A, W, None, epi=HF.EPI_NONE, out=C)), ("bf16 x9", lambda: x9lib.linear_nt_x9(A, W3, out=C))):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print(f"{label:7s} {name}: {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s-equivalent")
