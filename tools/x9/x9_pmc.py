"""run the x9 experiment kernel and the fp32-MFMA NT kernel on one dX shape (for rocprofv3 --pmc passes)"""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import x9lib
hw = importlib.import_module("sl-hwgat_amd")
HF = hw.functional
dev = "cuda:0"
M, N, K = 163840, 512, 1024
A = torch.randn(M, K, device=dev)
W = torch.randn(N, K, device=dev) * 0.05
W3 = x9lib.split3(W)
C = torch.empty(M, N, device=dev)
for _ in range(6):
    x9lib.linear_nt_x9(A, W3, out=C)
    HF.linear_nt(A, W, None, epi=HF.EPI_NONE, out=C)
torch.cuda.synchronize()
