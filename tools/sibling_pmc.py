"""Run the HGATE block-attention and WGATE band-attention kernels at their bench sizes (for rocprofv3
--pmc FETCH_SIZE / WRITE_SIZE passes).  `hwgat_merge` runs beside them as a calibration kernel with a
known byte count (reads E*4, writes E*4)."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hw = importlib.import_module("sl-hwgat_amd")
from oracle import hgat_oracle as OH, wgat_oracle as OW
HF = hw.functional
dev = "cuda:0"
dt = torch.bfloat16 if len(sys.argv) > 1 and sys.argv[1] == "bf16" else torch.float32      # `bf16`: the bf16-storage kernels
g = torch.Generator(device=dev).manual_seed(0)
# HGATE stage-0 shape: B64 F128 K29, 2 heads x 64
B, F, K, nH, d = 64, 128, 29, 2, 128
qkv = torch.randn(B, F, K, 3 * d, device=dev, generator=g).to(dt)
do = torch.randn(B, F, K, d, device=dev, generator=g).to(dt)
o, dq = torch.empty_like(do), torch.empty_like(qkv)
bits = HF.blk_mask_bits(OH.block_adjacency(), K).to(dev)
for it in range(5):
    for sh in (False, True):
        HF.attn_fwd("blk", qkv, o, bits, None, nH, sh)
        HF.attn_bwd("blk", qkv, do, dq, bits, None, nH, sh)
print("HGATE E bytes", B * F * K * d * qkv.element_size())
# WGATE shape: B64 F128 K64, 8 heads x 16
K, nH = 64, 8
qkv = torch.randn(B, F, K, 3 * d, device=dev, generator=g).to(dt)
do = torch.randn(B, F, K, d, device=dev, generator=g).to(dt)
o, dq = torch.empty_like(do), torch.empty_like(qkv)
rows = HF.band_mask_rows(OW.band_adjacency(F, K // 16), F).to(dev)
for it in range(10):
    HF.attn_fwd("band", qkv, o, rows, None, nH, False)
    HF.attn_bwd("band", qkv, do, dq, rows, None, nH, False)
    HF.temporal_merge(do)
torch.cuda.synchronize()
print("WGATE E bytes", B * F * K * d * qkv.element_size())
