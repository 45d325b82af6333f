#!/usr/bin/env python3
"""bf16 band-attention kernels against the fp32 kernels on the same (bf16-representable) data: where do they differ"""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hw = importlib.import_module("sl-hwgat_amd")
HF = hw.functional
from oracle import wgat_oracle as OW
B, F, nW, nH, hd = [int(a) for a in (sys.argv[1:6] if len(sys.argv) > 5 else (2, 8, 2, 8, 16))]
d, K = nH * hd, nW * 16
dev = "cuda:0"
g = torch.Generator().manual_seed(1)
qkv = (torch.randn(B, F, K, 3 * d, generator=g) * 0.8).to(torch.bfloat16).to(dev)
do = torch.randn(B, F, K, d, generator=g).to(torch.bfloat16).to(dev)
rows = HF.band_mask_rows(OW.band_adjacency(F, nW), F).to(dev)


def run(dt):
    x, gy = qkv.to(dt), do.to(dt)
    o, dq = torch.empty_like(gy), torch.empty_like(x)
    HF.attn_fwd("band", x, o, rows, None, nH, False)
    HF.attn_bwd("band", x, gy, dq, rows, None, nH, False)
    torch.cuda.synchronize()
    return o.float(), dq.float()


o32, g32 = run(torch.float32)
o16, g16 = run(torch.bfloat16)
for name, a, b in (("o", o16, o32), ("dq", g16[..., :d], g32[..., :d]), ("dk", g16[..., d:2 * d], g32[..., d:2 * d]), ("dv", g16[..., 2 * d:], g32[..., 2 * d:])):
    bad = ~torch.isfinite(a)
    err = (a - b).abs()
    err[bad] = 0
    print(f"{name}: non-finite {int(bad.sum())} of {a.numel()}, rel err {float(err.norm() / b.norm()):.3e}, max abs {float(err.max()):.3e}")
    if bad.any():
        idx = bad.nonzero()
        print("   first non-finite (b, f, k, c):", idx[:8].tolist(), " frames:", sorted(set(idx[:, 1].tolist())), " heads:", sorted(set((idx[:, 3] // hd).tolist())))
    per_f = (a - b).nan_to_num(0).pow(2).sum(dim=(0, 2, 3)).sqrt() / b.pow(2).sum(dim=(0, 2, 3)).sqrt()
    print("   per-frame rel err:", " ".join(f"{float(v):.1e}" for v in per_f[:40]))
