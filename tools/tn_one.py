"""TN (dW) kernel alone on the stage-2 and stage-0 shapes, for PMC passes"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hw = importlib.import_module("sl-hwgat_amd")
HF = hw.functional
dev = "cuda:0"
for M, N, K in ((163840, 1536, 512), (655360, 384, 128)):
    x = torch.randn(M, K, device=dev)
    dy = torch.randn(M, N, device=dev)
    w = torch.randn(N, K, device=dev) * 0.02
    out = torch.empty(M, N, device=dev)
    dW = torch.zeros(N, K, device=dev)
    db = torch.zeros(N, device=dev)
    for _ in range(4):
        HF.linear_tn(dy, x, dW, db)
        HF.linear_nt(x, w, None, out=out, epi=HF.EPI_NONE)
torch.cuda.synchronize()
