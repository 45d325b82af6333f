"""the fused bf16 NT launches of one block at a stage-0 and a stage-2 shape, a few times each (for rocprofv3 --pmc)"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hw = importlib.import_module("sl-hwgat_amd")
HF = hw.functional
dev = "cuda:0"
dt = torch.bfloat16 if os.environ.get("NT_LAB_DTYPE", "bf16") == "bf16" else torch.float32
for M, d in ((655360, 128), (163840, 512)):
    r = lambda *s: torch.randn(*s, device=dev).to(dt)
    x, aux, res = r(M, d), r(M, 2 * d), r(M, d)
    w2t, w1, wq = (torch.randn(2 * d, d, device=dev) * .05).to(dt), (torch.randn(2 * d, d, device=dev) * .05).to(dt), (torch.randn(d, d, device=dev) * .05).to(dt)
    gamma, beta = torch.randn(d, device=dev), torch.randn(d, device=dev)
    mean, rstd = HF.ln_stats(x, gamma, beta)
    o2, o1 = torch.empty(M, 2 * d, device=dev, dtype=dt), torch.empty(M, d, device=dev, dtype=dt)
    b2 = torch.randn(2 * d, device=dev)
    for _ in range(3):
        HF.linear_nt(x, w2t, None, pro=HF.PRO_DROP, pro_seed=3, pro_p=.1, epi=HF.EPI_GELU_BWD, aux=aux, epi_seed=2, epi_p=.1, out=o2)   # d_h1
        HF.linear_nt(x, w1, b2, pro=HF.PRO_LN, ln=(mean, rstd, gamma, beta), epi=HF.EPI_BIAS_GELU_DROP, epi_seed=2, epi_p=.1, out=o2)     # fc1
        HF.linear_nt(x, wq, None, epi=HF.EPI_NONE, out=o1)                                                                                   # plain
torch.cuda.synchronize()
