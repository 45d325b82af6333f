#!/usr/bin/env python3
"""HGATE block-attention kernels alone, bf16, stage-0 bench shape (for rocprofv3 --pmc passes; tools/pmc_sum.py sums them)"""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hw = importlib.import_module("sl-hwgat_amd")
from oracle import hgat_oracle as OH
HF = hw.functional
dev = "cuda:0"
B, F, K, nH, d = 64, 128, 29, 2, 128
dt = torch.float32 if len(sys.argv) > 1 and sys.argv[1] == "f32" else torch.bfloat16
qkv = torch.randn(B, F, K, 3 * d, device=dev).to(dt)
do = torch.randn(B, F, K, d, device=dev).to(dt)
o, dq = torch.empty_like(do), torch.empty_like(qkv)
bits = HF.blk_mask_bits(OH.block_adjacency(), K).to(dev)
for it in range(5):
    HF.attn_fwd("blk", qkv, o, bits, None, nH, False)
    HF.attn_bwd("blk", qkv, do, dq, bits, None, nH, False)
torch.cuda.synchronize()
