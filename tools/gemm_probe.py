"""Library GEMM rates (torch -> rocBLAS/hipBLASLt) on the HWGAT linear shapes: the bar a
hand-written MFMA kernel has to beat.  Not part of the product path."""
import sys
import torch

dev = "cuda:0"
B, T, K = 64, 128, 80
shapes = []
for i, d in enumerate((128, 256, 512)):
    M = B * (T >> i) * K
    shapes += [(M, d, 3 * d, "qkv"), (M, d, d, "proj"), (M, d, 2 * d, "fc1"), (M, 2 * d, d, "fc2")]


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


for dt in (torch.float32, torch.bfloat16):
    for M, Kd, N, name in shapes:
        x = torch.randn(M, Kd, device=dev, dtype=dt)
        w = torch.randn(N, Kd, device=dev, dtype=dt) * 0.02
        b = torch.zeros(N, device=dev, dtype=dt)
        dy = torch.randn(M, N, device=dev, dtype=dt)
        t_f = bench(lambda: torch.nn.functional.linear(x, w, b))
        t_dx = bench(lambda: dy @ w)
        t_dw = bench(lambda: dy.t() @ x)
        fl = 2.0 * M * Kd * N
        by = (M * Kd + M * N) * x.element_size()
        print(f"{str(dt)[6:]:9s} {name:5s} M={M} K={Kd} N={N}: fwd {fl / t_f / 1e12:6.1f} TF ({by / t_f / 1e12:4.2f} TB/s)"
              f"  dx {fl / t_dx / 1e12:6.1f} TF  dW {fl / t_dw / 1e12:6.1f} TF", flush=True)
# plain copy bandwidth for reference
x = torch.empty(1 << 28, device=dev)
y = torch.empty_like(x)
t = bench(lambda: y.copy_(x))
print(f"copy 1 GiB: {2 * x.numel() * 4 / t / 1e12:.2f} TB/s")
