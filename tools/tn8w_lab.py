#!/usr/bin/env python3
"""A/B of the bf16 weight-gradient launches of BASELINE config 3 (B=64): the eight-wave LDS-DMA kernel
(gemm_bf16_tn8w.hip; LayerNorm'd operands written by hwgat_ln_bwd_xn) against the kernels it replaces (LAB library with
HWGAT_TN8W=0: gemm_tn256_bf16_k / gemm_tn_bf16_k, LayerNorm in the loader).  For the two LayerNorm -> Linear pairs the
comparison is the PAIR of launches each way: { LayerNorm backward, dW } old vs { LayerNorm backward + xn, dW } new."""
import ctypes, importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["HWGAT_TN8W"] = "0"
hw = importlib.import_module("sl-hwgat_amd")
HF, L = hw.functional, hw._lib
dev = "cuda:0"
new = L.lib()
old = ctypes.CDLL(os.path.join(ROOT, "sl-hwgat_amd", "libhwgat_hip_lab.so"))
assert old.hwgat_is_lab_build() == 1, "not the LAB library: build it with `python sl-hwgat_amd/build.py --lab`"
for name, args in L._SIGS.items():
    fn = getattr(old, name)
    fn.argtypes, fn.restype = args, ctypes.c_int


class use:
    def __init__(self, h):
        self.h = h

    def __enter__(self):
        self.prev, L._lib = L._lib, self.h

    def __exit__(self, *a):
        L._lib = self.prev


B, T, K = 64, 128, 80
reps = int(os.environ.get("TN8W_REPS", "8"))
stages = [int(a) for a in os.environ.get("TN8W_STAGES", "1,2").split(",")]
dt = torch.bfloat16


def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record()
    return e0, e1


def med(ev):
    t = sorted(a.elapsed_time(b) for a, b in ev)
    return t[len(t) // 2] * 1e-3, t[0] * 1e-3


tot_n = tot_o = 0.0
for i in stages:
    d = 128 << i
    M = B * (T >> i) * K
    depth = (2, 2, 4)[i]
    g = torch.Generator(device=dev).manual_seed(i)
    act = lambda *s: torch.randn(*s, device=dev, generator=g).to(dt)
    x, dy, d3, d2, u2, res = act(M, d), act(M, d), act(M, 3 * d), act(M, 2 * d), act(M, 2 * d), act(M, d)
    gamma, beta = 1 + 0.2 * torch.randn(d, device=dev, generator=g), 0.2 * torch.randn(d, device=dev, generator=g)
    mean, rstd = HF.ln_stats(x, gamma, beta)
    dgm, dbt = torch.zeros(d, device=dev), torch.zeros(d, device=dev)
    cases = []
    for nm, Nn, Kk, A, Bm, ln in (("dWproj", d, d, dy, x, False), ("dW2   ", d, 2 * d, dy, u2, False),
                                  ("dW1   (LN pair)", 2 * d, d, d2, x, True), ("dWqkv (LN pair)", 3 * d, d, d3, x, True)):
        dW_n, dW_o = torch.zeros(Nn, Kk, device=dev), torch.zeros(Nn, Kk, device=dev)
        db_n, db_o = torch.zeros(Nn, device=dev), torch.zeros(Nn, device=dev)
        if ln:
            def f_new(A=A, Bm=Bm, dW=dW_n, db=db_n):
                _, xn = HF.ln_backward(dy, Bm, mean, rstd, gamma, res, dgm, dbt, beta=beta)
                HF.linear_tn(A, xn, dW, db)

            def f_old(A=A, Bm=Bm, dW=dW_o, db=db_o):
                HF.ln_backward(dy, Bm, mean, rstd, gamma, res, dgm, dbt)
                HF.linear_tn(A, Bm, dW, db, ln=(mean, rstd, gamma, beta))
        else:
            def f_new(A=A, Bm=Bm, dW=dW_n, db=db_n):
                HF.linear_tn(A, Bm, dW, db)

            def f_old(A=A, Bm=Bm, dW=dW_o, db=db_o):
                HF.linear_tn(A, Bm, dW, db)
        with use(new):
            f_new()
        with use(old):
            f_old()
        torch.cuda.synchronize()
        err = float((dW_n.double() - dW_o.double()).norm() / dW_o.double().norm())
        errb = float((db_n.double() - db_o.double()).norm() / db_o.double().norm())
        en, eo = [], []
        for _ in range(reps):
            with use(new):
                en.append(timed(f_new))
            with use(old):
                eo.append(timed(f_old))
        torch.cuda.synchronize()
        (mn, bn), (mo, bo) = med(en), med(eo)
        fl = 2.0 * M * Nn * Kk
        tot_n += mn * depth
        tot_o += mo * depth
        print(f"s{i} {nm:16s} M={M} N={Nn:5d} K={Kk:5d} | new {mn * 1e6:7.1f} ({bn * 1e6:6.1f}) us  old {mo * 1e6:7.1f} ({bo * 1e6:6.1f}) us  {mo / mn:5.2f}x | "
              f"new {fl / mn / 1e12:5.0f} TF (incl. LN pass where paired) | new vs old dW {err:.1e} db {errb:.1e}", flush=True)
print(f"per step (launches x depth, stages {stages}): new {tot_n * 1e3:.2f} ms, old {tot_o * 1e3:.2f} ms")
