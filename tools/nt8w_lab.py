#!/usr/bin/env python3
"""A/B of the bf16 NT linear kernels on the HWGAT shapes of BASELINE config 3 (B=64): the eight-wave LDS-DMA kernel
(gemm_bf16_nt8w.hip, what the product library dispatches) against the one-wave-per-SIMD kernel it replaces
(gemm_bf16_nt256.hip, reached through the LAB library `python sl-hwgat_amd/build.py --lab` with HWGAT_NT8W=0).
Both libraries live in ONE process and the launches alternate (new, old, new, old, ...) so clock / thermal drift hits
both alike.  Per case: outputs of both against an fp64 reference on a row sample, new vs old on the whole tensor, median
and minimum launch time, TFLOP/s, algorithmic TB/s.

    python tools/nt8w_lab.py            # all three stages
    NT8W_STAGES=2 NT8W_REPS=12 python tools/nt8w_lab.py
"""
import ctypes
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["HWGAT_NT8W"] = "0"                          # read (once) only by the LAB library
hw = importlib.import_module("sl-hwgat_amd")
HF, L = hw.functional, hw._lib
dev = "cuda:0"
new = L.lib()
lab_path = os.path.join(ROOT, "sl-hwgat_amd", "libhwgat_hip_lab.so")
old = ctypes.CDLL(lab_path)
assert old.hwgat_is_lab_build() == 1, "not the LAB library: build it with `python sl-hwgat_amd/build.py --lab`"
for name, args in L._SIGS.items():
    fn = getattr(old, name)
    fn.argtypes, fn.restype = args, ctypes.c_int


class use:
    def __init__(self, handle):
        self.h = handle

    def __enter__(self):
        self.prev, L._lib = L._lib, self.h

    def __exit__(self, *a):
        L._lib = self.prev


B, T, K = 64, 128, 80
stages = [int(a) for a in os.environ.get("NT8W_STAGES", "0,1,2").split(",")]
reps = int(os.environ.get("NT8W_REPS", "10"))
dt = torch.float32 if os.environ.get("NT8W_DTYPE", "bf16") == "f32" else torch.bfloat16      # NT8W_DTYPE=f32: the fp32 twin (gemm_f32_nt8w.hip vs gemm_nt256_k)


def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    return e0, e1


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


tot_new = tot_old = 0.0
print(f"{'case':44s} {'M':>7s} {'N':>5s} {'K':>5s} | new us (min)   old us (min)  speedup | new TF  old TF | new TB/s | err new/old vs fp64, new vs old")
for i in stages:
    d = 128 << i
    M = B * (T >> i) * K
    depth = (2, 2, 4)[i]
    g = torch.Generator(device=dev).manual_seed(i)

    def rnd(*s):
        return torch.randn(*s, device=dev, generator=g)

    def act(*s):
        return torch.randn(*s, device=dev, generator=g).to(dt)
    x, y3, u2 = act(M, d), act(M, 3 * d), act(M, 2 * d)
    res, aux = act(M, d), act(M, 2 * d)
    gamma, beta = 1 + 0.2 * rnd(d), 0.2 * rnd(d)
    mean, rstd = HF.ln_stats(x, gamma, beta)
    w_qkv32, w_p32, w1_32, w2_32 = (rnd(*s) * .05 for s in ((3 * d, d), (d, d), (2 * d, d), (d, 2 * d)))
    w_qkv, w_p, w1, w2 = (w.to(dt) for w in (w_qkv32, w_p32, w1_32, w2_32))
    w_qkv_t, w_p_t, w1_t, w2_t = (w.t().contiguous() for w in (w_qkv, w_p, w1, w2))
    b3, b1, b2 = rnd(3 * d), rnd(d), rnd(2 * d)
    ln = (mean, rstd, gamma, beta)
    sample = torch.cat([torch.arange(0, 384), torch.arange(M // 2 + 100, M // 2 + 228), torch.arange(M - 256, M)]).to(dev)

    def lnx():
        return torch.nn.functional.layer_norm(x[sample].double(), (d,), gamma.double(), beta.double())

    def mask(shape_n, seed):
        full = HF.dropout_mask((M, shape_n), seed, 0.1, dev)
        return full[sample].double()
    # (name, N, K, bytes moved beyond A and C in units of M*N elements, launch, fp64 reference on `sample` rows)
    cases = [
        ("qkv   LN(folded) + bias", 3 * d, d, 0, lambda: HF.linear_nt_ln(x, w_qkv32, b3, ln),
         lambda: lnx() @ w_qkv32.double().t() + b3.double()),
        ("proj  bias+drop+res +row stats", d, d, 1, lambda: HF.linear_nt(x, w_p, b1, epi=HF.EPI_BIAS_DROP_RES, res=res, epi_seed=1, epi_p=.1, stats=True)[0],
         lambda: res[sample].double() + (x[sample].double() @ w_p.double().t() + b1.double()) * mask(d, 1)),
        ("fc1   LN(folded) + bias+gelu+drop (gelu' saved)", 2 * d, d, 1, lambda: HF.linear_nt_ln(x, w1_32, b2, ln, epi=HF.EPI_BIAS_GELU_DROP_G, epi_seed=2, epi_p=.1)[0],
         lambda: torch.nn.functional.gelu(lnx() @ w1_32.double().t() + b2.double()) * mask(2 * d, 2)),
        ("fc2   bias+drop+res +row stats", d, 2 * d, 1, lambda: HF.linear_nt(u2, w2, b1, epi=HF.EPI_BIAS_DROP_RES, res=res, epi_seed=3, epi_p=.1, stats=True)[0],
         lambda: res[sample].double() + (u2[sample].double() @ w2.double().t() + b1.double()) * mask(d, 3)),
        ("d_h1  x gelu' (masked gradient in)", 2 * d, d, 1, lambda: HF.linear_nt(x, w2_t, None, epi=HF.EPI_MUL_AUX, aux=aux),
         lambda: (x[sample].double() @ w2_t.double().t()) * aux[sample].double()),
        ("d_z   plain", d, 2 * d, 0, lambda: HF.linear_nt(u2, w1_t, None, epi=HF.EPI_NONE),
         lambda: u2[sample].double() @ w1_t.double().t()),
        ("d_o   plain", d, d, 0, lambda: HF.linear_nt(x, w_p_t, None, epi=HF.EPI_NONE),
         lambda: x[sample].double() @ w_p_t.double().t()),
        ("d_xn  plain", d, 3 * d, 0, lambda: HF.linear_nt(y3, w_qkv_t, None, epi=HF.EPI_NONE),
         lambda: y3[sample].double() @ w_qkv_t.double().t()),
    ]
    for name, N, Kd, extra, fn, ref in cases:
        if N % 256:
            continue                                        # neither kernel takes it (128-wide tiles)
        with use(new):
            a = fn()
        with use(old):
            b = fn()
        torch.cuda.synchronize()
        r = ref()
        e_new, e_old, e_ab = rel(a[sample], r), rel(b[sample], r), rel(a, b.float())
        ev_n, ev_o = [], []
        for _ in range(reps):
            with use(new):
                ev_n.append(timed(fn))
            with use(old):
                ev_o.append(timed(fn))
        torch.cuda.synchronize()
        tn = sorted(x0.elapsed_time(x1) for x0, x1 in ev_n)
        to = sorted(x0.elapsed_time(x1) for x0, x1 in ev_o)
        mn, mo = tn[len(tn) // 2] * 1e-3, to[len(to) // 2] * 1e-3
        fl = 2.0 * M * N * Kd
        byts = (2.0 if dt == torch.bfloat16 else 4.0) * M * (Kd + N * (1 + extra + (1 if "gelu'" in name and "saved" in name else 0)))
        tot_new += mn * depth
        tot_old += mo * depth
        print(f"s{i} {name:41s} {M:7d} {N:5d} {Kd:5d} | {mn * 1e6:7.1f} ({tn[0] * 1e3:6.1f})  {mo * 1e6:7.1f} ({to[0] * 1e3:6.1f})  {mo / mn:5.2f}x | "
              f"{fl / mn / 1e12:6.0f}  {fl / mo / 1e12:6.0f} | {byts / mn / 1e12:5.2f} | {e_new:.1e} {e_old:.1e} {e_ab:.1e}", flush=True)
        del a, b
print(f"per step (launches x depth): new {tot_new * 1e3:.2f} ms, old {tot_old * 1e3:.2f} ms")
