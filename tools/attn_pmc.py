"""Run the fused window-attention kernels at the BASELINE config-2 size (for rocprofv3 --pmc
FETCH_SIZE / WRITE_SIZE passes).  `hwgat_merge` runs beside them as a calibration kernel with a
known byte count in the same 16 B/lane access width (reads E*4, writes E*4)."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hw = importlib.import_module("sl-hwgat_amd")
from oracle import hwgat_oracle as O
HF = hw.functional
dev = "cuda:0"
B, F, nW, nH, hd = [int(a) for a in sys.argv[1:6]] if len(sys.argv) >= 6 else (64, 128, 5, 2, 64)   # default: config 2, stage 0
d, K = nH * hd, nW * 16
dt = torch.bfloat16 if os.environ.get("ATTN_PMC_DTYPE", "f32") == "bf16" else torch.float32    # bf16: config 3
g = torch.Generator(device=dev).manual_seed(0)
qkv = torch.randn(B, F, K, 3 * d, device=dev, generator=g).to(dt)
do = torch.randn(B, F, K, d, device=dev, generator=g).to(dt)
bits = HF.mask_bits(O.window_adjacency(nW)).to(dev)
thr = torch.tensor([0.2], device=dev)
o = torch.empty(B, F, K, d, device=dev, dtype=dt)
dqkv = torch.empty_like(qkv)
for it in range(5):
    for shifted in (0, 1):
        HF.call("hwgat_win_attn_fwd", HF.ptr(qkv), HF.ptr(o), HF.ptr(bits), HF.ptr(thr), B, F, nW, nH, hd, shifted, HF.dtype_code(qkv), HF.stream())
        HF.call("hwgat_win_attn_bwd", HF.ptr(qkv), HF.ptr(do), HF.ptr(dqkv), HF.ptr(bits), HF.ptr(thr), B, F, nW, nH, hd, shifted, HF.dtype_code(qkv), HF.stream())
    HF.temporal_merge(do)
torch.cuda.synchronize()
print("E bytes", B * F * K * d * qkv.element_size())
