"""plain bf16 NT linear on a stage-0 and a stage-2 shape, for PMC passes"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hw = importlib.import_module("sl-hwgat_amd")
HF = hw.functional
dev = "cuda:0"
for M, N, K in ((655360, 384, 128), (163840, 1536, 512)):
    x = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for _ in range(4):
        HF.linear_nt(x, w, None, out=out, epi=HF.EPI_NONE)
torch.cuda.synchronize()
