#!/usr/bin/env python3
"""The stage-2 bf16 linear launches of BASELINE config 3 (M = 163 840, d = 512) a few times each, for rocprofv3 passes:
    rocprofv3 --kernel-trace --pmc FETCH_SIZE  -d out/fetch -- python3 tools/nt8w_pmc.py
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d out/write -- python3 tools/nt8w_pmc.py
(separate passes; FETCH_SIZE x 2 on gfx950, MI355X_MICROARCH.md) and `python tools/pmc_sum.py out/fetch gemm_` to add up.
Also runs merge_k (a plain permutation copy with known bytes) as the calibration kernel."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hw = importlib.import_module("sl-hwgat_amd")
HF = hw.functional
dev, dt = "cuda:0", torch.bfloat16
M, d, REPS = 163840, 512, 3
g = torch.Generator(device=dev).manual_seed(0)
act = lambda *s: torch.randn(*s, device=dev, generator=g).to(dt)
rnd = lambda *s: torch.randn(*s, device=dev, generator=g)
x, y3, u2, res, aux = act(M, d), act(M, 3 * d), act(M, 2 * d), act(M, d), act(M, 2 * d)
gamma, beta = 1 + 0.2 * rnd(d), 0.2 * rnd(d)
mean, rstd = HF.ln_stats(x, gamma, beta)
ln = (mean, rstd, gamma, beta)
wq, wp, w1, w2 = rnd(3 * d, d) * .05, rnd(d, d) * .05, rnd(2 * d, d) * .05, rnd(d, 2 * d) * .05
b3, b1, b2 = rnd(3 * d), rnd(d), rnd(2 * d)
wqt, wpt, w1t, w2t = (w.to(dt).t().contiguous() for w in (wq, wp, w1, w2))
dwq, dwp, dw1, dw2 = (torch.zeros_like(w) for w in (wq, wp, w1, w2))
dbq, dbp, db1, db2 = (torch.zeros(w.shape[0], device=dev) for w in (wq, wp, w1, w2))
for _ in range(REPS):
    HF.linear_nt_ln(x, wq, b3, ln)                                                                     # qkv
    HF.linear_nt(x, wp.to(dt), b1, epi=HF.EPI_BIAS_DROP_RES, res=res, epi_seed=1, epi_p=.1, stats=True)   # proj
    HF.linear_nt_ln(x, w1, b2, ln, epi=HF.EPI_BIAS_GELU_DROP_G, epi_seed=2, epi_p=.1)                  # fc1
    HF.linear_nt(u2, w2.to(dt), b1, epi=HF.EPI_BIAS_DROP_RES, res=res, epi_seed=3, epi_p=.1, stats=True)  # fc2
    HF.linear_nt(x, w2t, None, epi=HF.EPI_MUL_AUX, aux=aux)                                            # d_h1
    HF.linear_nt(u2, w1t, None, epi=HF.EPI_NONE)                                                       # d_z
    HF.linear_nt(x, wpt, None, epi=HF.EPI_NONE)                                                        # d_o
    HF.linear_nt(y3, wqt, None, epi=HF.EPI_NONE)                                                       # d_xn
    HF.linear_tn(x, res, dwp, dbp)                                                                     # dWproj
    HF.linear_tn(x, u2, dw2, db2)                                                                      # dW2
    HF.linear_tn(u2, x, dw1, db1)                                                                      # dW1 (plain xn operand)
    HF.linear_tn(y3, x, dwq, dbq)                                                                      # dWqkv
    xm = x.view(8, 256, 80, d)
    HF.temporal_merge(xm)                                                                              # calibration: 2 x 168 MB
torch.cuda.synchronize()
print("done")
