#!/usr/bin/env python3
"""Does the 256 MB Infinity Cache keep a tensor between the kernel that writes it and the kernel that reads it?

For working-set sizes S from 8 MiB to 2 GiB, three chains of plain torch kernels (vectorised elementwise / reduce):
  rd   : sum(a) repeated                      -- read-only, the same S bytes again and again
  rw   : a.add_(1) repeated                   -- reads and rewrites the same S bytes
  pc   : b = a + 1 ; c = b * 2 (ping-pong)    -- producer -> consumer, S written then S read by the next kernel
Prints effective bytes / second per chain and size.  HBM streams at ~4.8-5.3 TB/s through these kernels; a rate well above
that at small S means the Infinity Cache (or L2, S <= 32 MiB) served the reads.
    python tools/mall_probe.py > gpurun_out/mall_probe.txt
"""
import torch

dev = torch.device("cuda:0")


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


print(f"{'MiB':>6} {'rd TB/s':>9} {'rw TB/s':>9} {'pc TB/s':>9}")
for mib in (8, 16, 32, 64, 96, 128, 192, 256, 384, 512, 1024, 2048):
    n = mib * 1024 * 1024 // 2
    a = torch.zeros(n, device=dev, dtype=torch.bfloat16)
    b = torch.empty_like(a)
    c = torch.empty_like(a)
    reps = max(4, min(200, 4096 // mib))
    t_rd = timed(lambda: a.view(torch.int16).sum(dtype=torch.int32), reps)
    t_rw = timed(lambda: a.add_(1), reps)

    def pc():
        torch.add(a, 1, out=b)
        torch.mul(b, 2, out=c)
    t_pc = timed(pc, reps)
    S = n * 2
    print(f"{mib:6d} {S / t_rd / 1e12:9.2f} {2 * S / t_rw / 1e12:9.2f} {4 * S / t_pc / 1e12:9.2f}", flush=True)
    del a, b, c
