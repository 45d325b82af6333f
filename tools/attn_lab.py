#!/usr/bin/env python3
"""Window-attention kernel rates at a given shape (default: one micro-batch of BASELINE config 5, head_dim 128):
algorithmic bytes (4 E s forward, 7 E s backward) / time against the 8 TB/s HBM roof.
    python tools/attn_lab.py [B F nW nH hd]          HWGAT_ATTN_SPLIT=0 selects the one-wave hd=128 backward"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hw = importlib.import_module("sl-hwgat_amd")
HF = hw.functional
from oracle import hwgat_oracle as O
dev = "cuda:0"
B, F, nW, nH, hd = [int(a) for a in sys.argv[1:6]] if len(sys.argv) >= 6 else (16, 256, 7, 2, 128)
d, K = nH * hd, nW * 16
bits = HF.mask_bits(O.window_adjacency(nW)).to(dev)
thr = torch.tensor([0.3], device=dev)


def bench(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


for dt in (torch.float32, torch.bfloat16):
    qkv = torch.randn(B, F, K, 3 * d, device=dev).to(dt)
    o = torch.empty(B, F, K, d, device=dev, dtype=dt)
    do = torch.randn(B, F, K, d, device=dev).to(dt)
    dqkv = torch.empty_like(qkv)
    E = B * F * K * d * qkv.element_size()
    for shifted in (0, 1):
        tf = bench(lambda: HF.attn_fwd("win", qkv, o, bits, thr, nH, shifted))
        tb = bench(lambda: HF.attn_bwd("win", qkv, do, dqkv, bits, thr, nH, shifted))
        print(f"{str(dt)[6:]:9s} B={B} F={F} nW={nW} nH={nH} hd={hd} shifted={shifted}: fwd {tf * 1e6:7.1f} us {4 * E / tf / 8e12:5.3f} of HBM | "
              f"bwd {tb * 1e6:7.1f} us {7 * E / tb / 8e12:5.3f} of HBM", flush=True)
