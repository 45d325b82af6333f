#!/usr/bin/env python3
"""Per-launch rates of the fp32 / bf16 weight-gradient (TN) kernels on the HWGAT shapes (B=64 config 2), with the
prologues the fused block backward uses (dropout mask on dY, LayerNorm on X).  NT_LAB_DTYPE=bf16 for config 3."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hw = importlib.import_module("sl-hwgat_amd")
HF = hw.functional
if os.environ.get("TN_LAB_LIB") == "lab":                    # the LAB library (build.py --lab): its environment switches are live
    import ctypes
    L = hw._lib
    L.lib()
    lab = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sl-hwgat_amd", "libhwgat_hip_lab.so"))
    assert lab.hwgat_is_lab_build() == 1, "not the LAB library: build it with `python sl-hwgat_amd/build.py --lab`"
    for name, args in L._SIGS.items():
        fn = getattr(lab, name)
        fn.argtypes, fn.restype = args, (ctypes.c_int64 if name.endswith("_ws_bytes") else ctypes.c_int)
    L._lib = lab
dev = "cuda:0"
B, T, K = 64, 128, 80
dt = torch.bfloat16 if os.environ.get("NT_LAB_DTYPE", "f32") == "bf16" else torch.float32


def bench(fn, n=8):
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


plain = os.environ.get("TN_LAB_PLAIN") == "1"               # no prologue at all: the bare kernel
total = 0.0
for i in [int(a) for a in os.environ.get("TN_LAB_STAGES", "0,1,2").split(",")]:
    d = 128 << i
    M = B * (T >> i) * K
    depth = (2, 2, 4)[i]
    g = torch.Generator(device=dev).manual_seed(i)
    act = lambda *s: torch.randn(*s, device=dev, generator=g).to(dt)
    x1, x2, x3 = act(M, d), act(M, 2 * d), act(M, 3 * d)
    gamma, beta = torch.randn(d, device=dev), torch.randn(d, device=dev)
    mean, rstd = HF.ln_stats(x1, gamma, beta)
    ln = (mean, rstd, gamma, beta)
    for name, A, Bm, kw in (("dWqkv  LN(x)  ", x3, x1, dict(ln=ln)), ("dWproj drop(dy)", x1, x1, dict(pro_seed=1, pro_p=.1)),
                            ("dW1    LN(y)  ", x2, x1, dict(ln=ln)), ("dW2    drop(dy)", x1, x2, dict(pro_seed=3, pro_p=.1))):
        N, Kd = A.shape[1], Bm.shape[1]
        dW, db = torch.zeros(N, Kd, device=dev), torch.zeros(N, device=dev)
        t = bench(lambda: HF.linear_tn(A, Bm, dW, db, **({} if plain else kw)))
        fl = 2.0 * M * N * Kd
        total += t * depth
        print(f"stage {i} d={d:4d} {name} M={M} N={N:4d} K={Kd:4d}: {t * 1e6:8.1f} us  {fl / t / 1e12:6.1f} TF  "
              f"{(A.numel() + Bm.numel()) * A.element_size() / t / 1e12:5.2f} TB/s", flush=True)
print(f"TN launches per step (depths 2,2,4): {total * 1e3:.2f} ms")
