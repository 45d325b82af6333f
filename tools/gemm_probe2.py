"""hand-written fp32 MFMA linears vs the library on the HWGAT shapes (B=64 config 2)."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hw = importlib.import_module("sl-hwgat_amd")
HF = hw.functional
dev = "cuda:0"
B, T, K = 64, 128, 80


def bench(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


for i, d in enumerate((128, 256, 512)):
    M = B * (T >> i) * K
    for name, N, Kd in (("qkv", 3 * d, d), ("proj", d, d), ("fc1", 2 * d, d), ("fc2", d, 2 * d)):
        x = torch.randn(M, Kd, device=dev)
        w = torch.randn(N, Kd, device=dev) * 0.02
        b = torch.zeros(N, device=dev)
        dy = torch.randn(M, N, device=dev)
        res = torch.randn(M, N, device=dev)
        out = torch.empty(M, N, device=dev)
        dW = torch.zeros(N, Kd, device=dev)
        db = torch.zeros(N, device=dev)
        fl = 2.0 * M * Kd * N
        t_lib = bench(lambda: torch.nn.functional.linear(x, w, b))
        t_nt = bench(lambda: HF.linear_nt(x, w, b, out=out))
        t_nt_f = bench(lambda: HF.linear_nt(x, w, b, epi=HF.EPI_BIAS_DROP_RES, res=res, epi_seed=1, epi_p=0.1, out=out))
        t_libdw = bench(lambda: dy.t() @ x)
        t_tn = bench(lambda: HF.linear_tn(dy, x, dW, db))
        print(f"{name:5s} M={M} K={Kd} N={N}: fwd lib {fl / t_lib / 1e12:6.1f} TF | nt {fl / t_nt / 1e12:6.1f} TF | "
              f"nt+drop+res {fl / t_nt_f / 1e12:6.1f} TF || dW lib {fl / t_libdw / 1e12:6.1f} TF | tn(+db) {fl / t_tn / 1e12:6.1f} TF",
              flush=True)
