#!/usr/bin/env python3
"""time the HGATE block-attention kernels alone at the bench shape"""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hw = importlib.import_module("sl-hwgat_amd")
HF = hw.functional
from oracle import hgat_oracle as OH
B, F, K, nH, hd = 64, 128, 29, 2, 64
dt = torch.bfloat16 if len(sys.argv) > 1 and sys.argv[1] == "bf16" else torch.float32
d = nH * hd
dev = "cuda:0"
qkv = torch.randn(B, F, K, 3 * d, device=dev).to(dt)
do = torch.randn(B, F, K, d, device=dev).to(dt)
o, dq = torch.empty_like(do), torch.empty_like(qkv)
bits = HF.blk_mask_bits(OH.block_adjacency(), K).to(dev)
E = B * F * K * d * qkv.element_size()
for name, fn, mult in (("fwd", lambda: HF.attn_fwd("blk", qkv, o, bits, None, nH, True), 4),
                       ("bwd", lambda: HF.attn_bwd("blk", qkv, do, dq, bits, None, nH, True), 7)):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"{dt} {name}: {us:8.1f} us  {mult * E / us / 1e6:7.2f} TB/s")
