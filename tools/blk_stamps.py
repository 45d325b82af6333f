#!/usr/bin/env python3
"""Where a unit's time goes in the four-wave fp32 block-attention backward (blk_attn_f32.hip): s_memtime stamps (LAB library)
of wave 0 of every persistent workgroup over its first 8 units.  Stamps: 0 loop top, 1 next unit's loads issued + images
written, 2 barrier passed, 3 S / dP products done, 4 softmax / dS done, 5 dQ done, 6 exchange done (two barriers), 7 phase B
done, 8 end-of-unit barrier passed."""
import ctypes, importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
hw = importlib.import_module("sl-hwgat_amd")
HF, L = hw.functional, hw._lib
from oracle import hgat_oracle as OH
lab = ctypes.CDLL(os.path.join(ROOT, "sl-hwgat_amd", "libhwgat_hip_lab.so"))
assert lab.hwgat_is_lab_build() == 1, "not the LAB library: build it with `python sl-hwgat_amd/build.py --lab`"
for name, args in L._SIGS.items():
    fn = getattr(lab, name)
    fn.argtypes, fn.restype = args, (ctypes.c_int64 if name.endswith("_bytes") else ctypes.c_int)
L.lib()
L._lib = lab
lab.hwgat_lab_blk_stamps.argtypes, lab.hwgat_lab_blk_stamps.restype = [ctypes.c_void_p], ctypes.c_int
dev = "cuda:0"
B, F, K, nH, d = 64, 128, 29, 2, 128
qkv = torch.randn(B, F, K, 3 * d, device=dev)
do = torch.randn(B, F, K, d, device=dev)
dq = torch.empty_like(qkv)
bits = HF.blk_mask_bits(OH.block_adjacency(), K).to(dev)
NWG = 512
st = torch.zeros(NWG * 8 * 10, device=dev, dtype=torch.int64)
def one(flags):
    os.environ["HWGAT_BLK_SKEW"] = str(flags)
    for _ in range(3):
        HF.attn_bwd("blk", qkv, do, dq, bits, None, nH, False)
    torch.cuda.synchronize()
    st.zero_()
    assert lab.hwgat_lab_blk_stamps(st.data_ptr()) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    HF.attn_bwd("blk", qkv, do, dq, bits, None, nH, False)
    e1.record()
    torch.cuda.synchronize()
    lab.hwgat_lab_blk_stamps(None)
    us = e0.elapsed_time(e1) * 1e3
    s = st.view(NWG, 8, 10).cpu().double()
    # stamp order inside a unit slot: 2 top, 3 S/dP done, 4 softmax/dS done, 5 dQ done, 6 exchange done, 7 phase B done,
    # 8 images free, 0 images written + last stores issued, 1 next unit decoded
    order = [2, 3, 4, 5, 6, 7, 8, 0, 1]
    names = ["S, dP (+ next loads)", "softmax, dS", "dQ + stores", "exchange (2 barriers)", "phase B", "barrier: images free",
             "write images + dK dV stores", "decode next"]
    print(f"flags {flags:#x}: kernel {us:.1f} us")
    for ts in (2, 4):
        r = s[:, ts]
        parts = [(r[:, order[i + 1]] - r[:, order[i]]).median().item() for i in range(8)]
        nxt = (s[:, ts + 1, 2] - r[:, 1]).median().item()
        print(f"  unit {ts}: " + " | ".join(f"{n} {p:.0f}" for n, p in zip(names, parts)) + f" | barrier: images visible {nxt:.0f}")
    period = ((s[:, 6, 2] - s[:, 1, 2]) / 5).median().item()
    print(f"  unit period {period:.0f} ticks; 16 units = {16 * period:.0f} ticks in {us:.1f} us -> s_memtime at {16 * period / us / 1e3:.2f} GHz")


for flags in (0, 0x10000, 0x20000, 0x30000):
    one(flags)
