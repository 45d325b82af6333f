"""attainable fp32 MFMA rate on this box (register-only loops), burst vs sustained, both MFMA shapes"""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hw = importlib.import_module("sl-hwgat_amd"); L = hw._lib
out = torch.zeros(256, device="cuda:0")
def run(blocks, nacc, iters, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        L.call("hwgat_debug_mfma_peak", L.ptr(out), blocks, iters, nacc, L.stream())
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1e-3
    fl = reps * blocks * 4 * iters * 4 * abs(nacc) * 2.0 * 32 * 32 * 2
    return fl / t / 1e12, t * 1e3
for name, nacc in (("32x32x2", 4), ("16x16x4", -4)):
    for blocks in (256, 512):
        run(blocks, nacc, 2000, 1)
        b, tb = run(blocks, nacc, 10000, 1)
        s, ts = run(blocks, nacc, 20000, 30)
        print(f"{name} blocks={blocks}: burst {b:6.1f} TF ({tb:.1f} ms) | sustained {s:6.1f} TF ({ts:.0f} ms)", flush=True)
# bf16 32x32x16 (n_acc code 100 + NACC): 32768 flop per MFMA
def run_bf16(blocks, nacc, iters, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        L.call("hwgat_debug_mfma_peak", L.ptr(out), blocks, iters, 100 + nacc, L.stream())
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1e-3
    return reps * blocks * 4 * iters * 4 * nacc * 32768.0 / t / 1e12, t * 1e3
for blocks in (256, 512):
    for nacc in (4, 16):
        run_bf16(blocks, nacc, 2000, 1)
        b, tb = run_bf16(blocks, nacc, 10000, 1)
        s, ts = run_bf16(blocks, nacc, 20000, 30)
        print(f"bf16 32x32x16 nacc={nacc} blocks={blocks}: burst {b:7.1f} TF ({tb:.1f} ms) | sustained {s:7.1f} TF ({ts:.0f} ms)", flush=True)
