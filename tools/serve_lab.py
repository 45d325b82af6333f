#!/usr/bin/env python3
"""eval forward latency, eager vs one HIP-graph replay (sl-hwgat_amd/serve.py), HWGATE at the headline shape, B = 1 .. 64"""
import importlib, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hw = importlib.import_module("sl-hwgat_amd")
serve = importlib.import_module("sl-hwgat_amd.serve")
dev = torch.device("cuda:0")
torch.manual_seed(0)
dt = torch.bfloat16 if "bf16" in sys.argv[1:] else torch.float32
hp = hw.HWGATEParams({"src_len": 128, "num_class": 2002}, 2, dev, num_kps=80)
model = hw.Model(*hp.get_model_params()).to(dev).eval()
model.set_activation_dtype(dt)
for B in (1, 2, 4, 8, 16, 64):
    x = torch.rand(B, 128, 80, 2, device=dev)
    fast = serve.GraphedEval(model, x)
    with torch.no_grad():
        ref = model(x)
    assert torch.equal(fast(x), ref), B

    def clock(fn, n=30):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    with torch.no_grad():
        te = clock(lambda: model(x))
    tg = clock(lambda: fast(x))
    print(f"{dt} B={B:3d}: eager {te:7.3f} ms ({B / te * 1e3:8.1f} clips/s)   graph replay {tg:7.3f} ms ({B / tg * 1e3:8.1f} clips/s)   {te / tg:4.2f}x", flush=True)
