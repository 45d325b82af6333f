#!/usr/bin/env python3
"""Is the eight-wave bf16 dW kernel waiting for HBM?  The same launch (N = 512, K = 1024, plain bf16 operands) over M slices
from 8 K rows (24 MB of operands: Infinity-Cache resident on the second pass) to 160 K rows (503 MB: streamed from HBM),
beside the plain NT launch of the same flops and bytes.  If the dW rate per row does not move with residency, the waves are
not waiting for HBM."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hw = importlib.import_module("sl-hwgat_amd")
HF = hw.functional
dev, dt = "cuda:0", torch.bfloat16
N, K = 512, 1024
g = torch.Generator(device=dev).manual_seed(0)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


print(f"{'M':>8} {'MB':>6} | {'dW us':>8} {'TF':>7} | {'NT us':>8} {'TF':>7}")
for M in (8192, 16384, 32768, 65536, 163840, 327680):
    A = torch.randn(M, N, device=dev, generator=g).to(dt)
    Bm = torch.randn(M, K, device=dev, generator=g).to(dt)
    W = (torch.randn(N, K, device=dev, generator=g) * 0.05).to(dt)
    dW, db = torch.zeros(N, K, device=dev), torch.zeros(N, device=dev)
    out = torch.empty(M, N, device=dev, dtype=dt)
    t_tn = timed(lambda: HF.linear_tn(A, Bm, dW, db))
    t_nt = timed(lambda: HF.linear_nt(Bm, W, None, epi=HF.EPI_NONE, out=out))
    fl = 2.0 * M * N * K
    print(f"{M:8d} {M * (N + K) * 2 / 1e6:6.0f} | {t_tn:8.1f} {fl / t_tn / 1e6:7.0f} | {t_nt:8.1f} {fl / t_nt / 1e6:7.0f}", flush=True)
