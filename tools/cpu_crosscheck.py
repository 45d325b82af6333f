#!/usr/bin/env python3
"""Development container only (needs /root/reference): is the oracle a fair stand-in for the reference as a CPU
baseline?  Times the reference `Model` and the oracle on the same cores, same clips of the benchmarked shape
(T=128, K=80, C=2, d0=128, 2002 classes), same variants bench.py's `cpu_baseline` uses.  SURVEY.md 8d wants the
restatement within +-20 % of the reference before its timing on the GPU box is trusted; results go to BASELINE.md."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from make_fixtures import import_reference, build_reference, forced_thresholds   # noqa: E402
from oracle import hwgat_oracle as O                                             # noqa: E402


def med(fn, n=3):
    fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[n // 2]


def main():
    Model, HWGATEParams, Loss = import_reference()
    crit = Loss()
    T, nW, C, d0, nc = 128, 5, 2, 128, 2002
    print("threads", torch.get_num_threads())
    for B in (2, 8):
        g = torch.Generator().manual_seed(7)
        x = torch.rand(B, T, nW * 16, C, generator=g)
        y = torch.randint(0, nc, (B,), generator=g)
        for name, drop, thr in (("eval", 0.0, None), ("train_drop0", 0.0, [0.5] * 8), ("train_drop0.1", 0.1, [0.5] * 8)):
            ref, hp, cfg = build_reference(Model, HWGATEParams, T=T, nW=nW, C=C, d0=d0, nc=nc, drop=drop, seed=1, wstd=0.02)
            ref.train(thr is not None)

            def ref_step():
                ref.zero_grad()
                if thr is None:
                    crit(ref(x), y).backward()
                else:
                    with forced_thresholds(thr):
                        crit(ref(x), y).backward()
            params = {k: v.requires_grad_(k not in ("B", "pos_encoder.pe")) for k, v in O.synth_params(1, weight_std=0.02, **cfg).items()}
            orc = O.OracleHWGAT(params, num_kps=nW * 16, temporal_dim=T, drop_rate=drop)

            def orc_step():
                for p in params.values():
                    p.grad = None
                O.smoothed_cross_entropy(orc.forward(x, thresholds=thr), y).backward()
            tr, to = med(ref_step), med(orc_step)
            print(f"B={B} {name:14s} reference {B / tr:6.3f} clips/s | oracle {B / to:6.3f} clips/s | oracle/reference {tr / to:5.2f}x", flush=True)


if __name__ == "__main__":
    main()
