#!/usr/bin/env python3
"""Per-launch rates of the fp32 NT linear kernels on the HWGAT shapes (B=64 config 2), every prologue / epilogue the
fused block uses.  Run once per kernel choice, e.g.
    HWGAT_NT256_MINK=100000 python tools/nt_lab.py     # 128x128 kernels only (round-1 path)
    python tools/nt_lab.py                             # default dispatch (256x256 one-wave-per-SIMD kernel where eligible)
Prints TFLOP/s per (stage, linear, variant) and the per-step total of the 32 NT launches."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hw = importlib.import_module("sl-hwgat_amd")
HF = hw.functional
dev = "cuda:0"
B, T, K = 64, 128, 80
stages = [int(a) for a in os.environ.get("NT_LAB_STAGES", "0,1,2").split(",")]
reps = int(os.environ.get("NT_LAB_REPS", "8"))
dt = torch.bfloat16 if os.environ.get("NT_LAB_DTYPE", "f32") == "bf16" else torch.float32


def bench(fn, n=reps):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


total = 0.0
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
    gamma, beta = rnd(d), rnd(d)
    mean, rstd = HF.ln_stats(x, gamma, beta)
    w_qkv32, w_p32, w1_32, w2_32 = (rnd(*s) * .05 for s in ((3 * d, d), (d, d), (2 * d, d), (d, 2 * d)))     # fp32 masters
    w_qkv, w_p, w1, w2 = (w.to(dt) for w in (w_qkv32, w_p32, w1_32, w2_32))
    w_qkv_t, w_p_t, w1_t, w2_t = (w.t().contiguous() for w in (w_qkv, w_p, w1, w2))
    b3, b1, b2 = rnd(3 * d), rnd(d), rnd(2 * d)
    o3, o1, o2, o2b = (torch.empty(M, 3 * d, device=dev, dtype=dt), torch.empty(M, d, device=dev, dtype=dt),
                       torch.empty(M, 2 * d, device=dev, dtype=dt), None)
    esz = 2 if dt == torch.bfloat16 else 4
    ln = (mean, rstd, gamma, beta)
    cases = [
        ("qkv   LN(folded) -> bias ", M, 3 * d, d, lambda: HF.linear_nt_ln(x, w_qkv32, b3, ln, out=o3)),
        ("proj  bias+drop+res      ", M, d, d, lambda: HF.linear_nt(x, w_p, b1, epi=HF.EPI_BIAS_DROP_RES, res=res, epi_seed=1, epi_p=.1, out=o1)),
        ("fc1   LN(folded) -> bias+gelu+drop, gelu' saved", M, 2 * d, d, lambda: HF.linear_nt_ln(x, w1_32, b2, ln, epi=HF.EPI_BIAS_GELU_DROP_G, epi_seed=2, epi_p=.1, out=o2)),
        ("fc2   bias+drop+res      ", M, d, 2 * d, lambda: HF.linear_nt(u2, w2, b1, epi=HF.EPI_BIAS_DROP_RES, res=res, epi_seed=3, epi_p=.1, out=o1)),
        ("d_h1  drop -> x gelu'    ", M, 2 * d, d, lambda: HF.linear_nt(x, w2_t, None, pro=HF.PRO_DROP, pro_seed=3, pro_p=.1, epi=HF.EPI_MUL_AUX, aux=aux, out=o2)),
        ("d_z   plain              ", M, d, 2 * d, lambda: HF.linear_nt(u2, w1_t, None, epi=HF.EPI_NONE, out=o1)),
        ("d_o   drop -> plain      ", M, d, d, lambda: HF.linear_nt(x, w_p_t, None, pro=HF.PRO_DROP, pro_seed=1, pro_p=.1, epi=HF.EPI_NONE, out=o1)),
        ("d_xn  plain              ", M, d, 3 * d, lambda: HF.linear_nt(y3, w_qkv_t, None, epi=HF.EPI_NONE, out=o1)),
    ]
    if os.environ.get("NT_LAB_STATS") == "1":       # the same two launches with the statistics (and merged-store) epilogue
        cases += [
            ("proj  +row statistics     ", M, d, d, lambda: HF.linear_nt(x, w_p, b1, epi=HF.EPI_BIAS_DROP_RES, res=res, epi_seed=1, epi_p=.1, stats=True)),
            ("fc2   +row statistics     ", M, d, 2 * d, lambda: HF.linear_nt(u2, w2, b1, epi=HF.EPI_BIAS_DROP_RES, res=res, epi_seed=3, epi_p=.1, stats=True)),
            ("fc2   +statistics, merged ", M, d, 2 * d, lambda: HF.linear_nt(u2, w2, b1, epi=HF.EPI_BIAS_DROP_RES, res=res, epi_seed=3, epi_p=.1, stats=True, merge=(T >> i, K))),
        ]
    if os.environ.get("NT_LAB_ABLATE") == "1":     # what each prologue / epilogue costs: the same shapes with pieces removed
        cases += [
            ("qkv   LN in the loader (pro 1)", M, 3 * d, d, lambda: HF.linear_nt(x, w_qkv, b3, pro=HF.PRO_LN, ln=ln, out=o3)),
            ("fc1   LN in the loader, pre-activation saved (epi 2)", M, 2 * d, d, lambda: HF.linear_nt(x, w1, b2, pro=HF.PRO_LN, ln=ln, epi=HF.EPI_BIAS_GELU_DROP, epi_seed=2, epi_p=.1, out=o2)),
            ("d_h1  drop -> gelu-bwd (epi 3)", M, 2 * d, d, lambda: HF.linear_nt(x, w2_t, None, pro=HF.PRO_DROP, pro_seed=3, pro_p=.1, epi=HF.EPI_GELU_BWD, aux=aux, epi_seed=2, epi_p=.1, out=o2)),
            ("qkv   (no LN) bias        ", M, 3 * d, d, lambda: HF.linear_nt(x, w_qkv, b3, out=o3)),
            ("fc1   LN -> bias only     ", M, 2 * d, d, lambda: HF.linear_nt(x, w1, b2, pro=HF.PRO_LN, ln=ln, out=o2)),
            ("fc1   (no LN) gelu+drop   ", M, 2 * d, d, lambda: HF.linear_nt(x, w1, b2, epi=HF.EPI_BIAS_GELU_DROP, epi_seed=2, epi_p=.1, out=o2)),
            ("fc1   (no LN) gelu, p=0   ", M, 2 * d, d, lambda: HF.linear_nt(x, w1, b2, epi=HF.EPI_BIAS_GELU_DROP, epi_seed=2, epi_p=0., out=o2)),
            ("fc1   (no LN) bias only   ", M, 2 * d, d, lambda: HF.linear_nt(x, w1, b2, out=o2)),
            ("d_h1  (no drop) gelu-bwd  ", M, 2 * d, d, lambda: HF.linear_nt(x, w2_t, None, epi=HF.EPI_GELU_BWD, aux=aux, epi_seed=2, epi_p=.1, out=o2)),
            ("d_h1  drop -> plain       ", M, 2 * d, d, lambda: HF.linear_nt(x, w2_t, None, pro=HF.PRO_DROP, pro_seed=3, pro_p=.1, epi=HF.EPI_NONE, out=o2)),
            ("d_h1  (no drop) plain     ", M, 2 * d, d, lambda: HF.linear_nt(x, w2_t, None, epi=HF.EPI_NONE, out=o2)),
            ("fc2   bias+res, p=0       ", M, d, 2 * d, lambda: HF.linear_nt(u2, w2, b1, epi=HF.EPI_BIAS_DROP_RES, res=res, epi_seed=3, epi_p=0., out=o1)),
        ]
    for name, m, n, k, fn in cases:
        t = bench(fn)
        fl = 2.0 * m * n * k
        extra = ("statistics" in name or "(no" in name or "only" in name or "p=0" in name or name.startswith("d_h1  drop -> plain")
                 or "(pro 1)" in name or "(epi 2)" in name or "(epi 3)" in name)
        if not extra:
            total += t * depth
        if extra and "statistics" not in name:
            print(f"stage {i} d={d:4d} {name} M={m} N={n:4d} K={k:4d}: {t * 1e6:8.1f} us  {fl / t / 1e12:6.1f} TF   (ablation)", flush=True)
            continue
        if "statistics" in name:
            print(f"stage {i} d={d:4d} {name} M={m} N={n:4d} K={k:4d}: {t * 1e6:8.1f} us  {fl / t / 1e12:6.1f} TF   (incl. zero-fill + finalize)", flush=True)
            continue
        mult = {"qkv ": 4, "proj": 3, "fc1 ": 5, "fc2 ": 4, "d_h1": 5, "d_z ": 3, "d_o ": 2, "d_xn": 4}[name[:4].ljust(4)]      # E-sized streams
        gb = mult * m * d * esz
        print(f"stage {i} d={d:4d} {name} M={m} N={n:4d} K={k:4d}: {t * 1e6:8.1f} us  {fl / t / 1e12:6.1f} TF  {gb / t / 1e12:5.2f} TB/s", flush=True)
print(f"NT launches per step (depths 2,2,4): {total * 1e3:.2f} ms")
