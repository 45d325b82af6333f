#!/usr/bin/env python3
"""Generate golden vectors from the REFERENCE model (development container only).

Run:  python tests/golden/make_fixtures.py        (needs /root/reference)

What it does: imports `/root/reference/hwgat/models/HWGATE.py` as-is, loads a
deterministic parameter set (`oracle.hwgat_oracle.synth_params`, regenerated
by the tests, so the 40 MB state_dict is never stored), runs the reference on
seeded inputs and stores inputs + outputs as small `.npz` files next to this
script.  Nothing from the reference is copied: fixtures are data only.

`timm` (requirements.txt:7) is not installed here and there is no network.
HWGATE.py:4 needs from it only `trunc_normal_`, used solely by
`_init_weights` (HWGATE.py:335).  We alias it to torch's own
`nn.init.trunc_normal_`; initial values are overwritten by `synth_params`
before anything is recorded, so the alias cannot influence a fixture.
"""
import os
import sys
import types
from contextlib import contextmanager

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/hwgat"
sys.path.insert(0, ROOT)

from oracle import hwgat_oracle as O  # noqa: E402


def import_reference():
    for name in ("timm", "timm.models", "timm.models.layers"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["timm.models.layers"].trunc_normal_ = torch.nn.init.trunc_normal_
    sys.path.insert(0, REF)
    from models.HWGATE import Model                     # noqa
    from models.model_params import HWGATEParams        # noqa
    from losses.SmoothCrossEntropy import SmoothedCrossEntropyLoss  # noqa
    return Model, HWGATEParams, SmoothedCrossEntropyLoss


def build_reference(Model, HWGATEParams, *, T, nW, C, d0, nc, drop=0.0, seed=11, wstd=0.08):
    hp = HWGATEParams({"src_len": T, "num_class": nc}, C, torch.device("cpu"))
    hp.num_kps = nW * 16
    hp.embed_dim = d0
    hp.drop_rate = drop
    hp.edges = [hp.edges[0]] * nW            # the 4 shipped lists are identical
    hp.adj_mat = torch.tensor(hp.get_adj_mat(), dtype=torch.float32)
    model = Model(*hp.get_model_params())
    cfg = dict(kp_dim=C, temporal_dim=T, num_classes=nc, embed_dim=d0,
               depths=tuple(hp.depths), ff_ratio=hp.ff_ratio, use_pe=hp.pe,
               num_kps=hp.num_kps, tp=hp.temporal_patch_size)
    synth = O.synth_params(seed, weight_std=wstd, **cfg)
    res = model.load_state_dict(synth, strict=False)
    assert not res.unexpected_keys, res.unexpected_keys
    assert all(k.endswith("attn_mask") for k in res.missing_keys), res.missing_keys
    # the reference's own PE buffer must equal the oracle's table
    assert torch.equal(model.state_dict()["pos_encoder.pe"], synth["pos_encoder.pe"])
    return model, hp, cfg


@contextmanager
def forced_thresholds(values):
    """HWGATE.py:96 draws `torch.rand(1).item()`; feed it a fixed sequence."""
    real, it = torch.rand, iter(values)
    torch.rand = lambda *a, **k: torch.tensor([next(it)], dtype=torch.float32)
    try:
        yield
    finally:
        torch.rand = real


def sub(t):
    """strided sample of a (B,F,K,d) activation, small enough to commit"""
    return t[:, ::5, ::3, ::7].contiguous().numpy()


def probe_vectors(name, n, k=4):
    """k fixed +-1 vectors of length n, a function of the parameter name only (tests regenerate them)"""
    import zlib
    rs = np.random.RandomState(zlib.crc32(name.encode()) & 0x7FFFFFFF)
    return rs.randint(0, 2, size=(k, n)).astype(np.float64) * 2.0 - 1.0


def grad_digest(model):
    """per parameter: first 48 entries, (norm, sum), and 4 random +-1 projections of the WHOLE gradient
    (a misplaced or permuted entry anywhere changes a projection)"""
    out = {}
    for name, p in model.named_parameters():
        if p.grad is None:
            continue
        g = p.grad.detach().double().flatten()
        out["gh." + name] = g[:48].float().numpy()
        out["gn." + name] = np.array([g.norm().item(), g.sum().item()], dtype=np.float64)
        out["gp." + name] = probe_vectors(name, g.numel()) @ g.numpy()
    return out


def main():
    Model, HWGATEParams, Loss = import_reference()
    torch.manual_seed(1001)                                   # configs.py:55-59
    crit = Loss()

    # ---- F1/F2/F3/F4: config 1 (T32, nW2, C2, d0 128), B=2 -------------
    T, nW, C, d0, nc, B = 32, 2, 2, 128, 10, 2
    model, hp, cfg = build_reference(Model, HWGATEParams, T=T, nW=nW, C=C, d0=d0, nc=nc)
    g = torch.Generator().manual_seed(7)
    x = torch.rand(B, T, nW * 16, C, generator=g)
    y = torch.randint(0, nc, (B,), generator=g)
    fx = {"x": x.numpy(), "y": y.numpy(), "adj": hp.adj_mat.numpy(),
          "cfg": np.array([T, nW, C, d0, nc, B, 11])}
    for k, v in model.state_dict().items():
        if k.endswith("attn_mask"):
            fx["mask." + k] = v.numpy()

    taps = {}

    def hook(name):
        def fn(_m, _i, out):
            taps[name] = out.detach()
        return fn
    blk = 0
    for i, layer in enumerate(model.layers):
        for j, b in enumerate(layer.blocks):
            b.register_forward_hook(hook(f"block{blk}"))
            blk += 1
        if layer.downsample is not None:
            layer.downsample.register_forward_hook(hook(f"merge{i}"))
    model.pos_encoder.register_forward_hook(hook("pe"))

    model.eval()
    with torch.no_grad():
        logits = model(x)
        feat = model.forward_features(x)
    fx["eval.logits"], fx["eval.feat"] = logits.numpy(), feat.numpy()
    for k, v in taps.items():
        fx["eval." + k] = sub(v)
    fx["eval.block0.full"] = taps["block0"][0, :4].numpy()
    fx["eval.block1.full"] = taps["block1"][0, -4:].numpy()

    # F3: fwd+bwd in eval()
    model.zero_grad()
    loss = crit(model(x), y)
    loss.backward()
    fx["evalbwd.loss"] = np.array(loss.item())
    fx.update({"evalbwd." + k: v for k, v in grad_digest(model).items()})

    # F4: train mode, drop_rate=0, thresholds injected
    model.train()
    cases = {"mid": [0.5, 0.2, 0.35, 0.08, 0.6, 0.15, 0.045, 0.9],
             "lo": [0.001] * 8, "hi": [0.999] * 8}
    for tag, thr in cases.items():
        model.zero_grad()
        with forced_thresholds(thr):
            out = model(x)
        loss = crit(out, y)
        loss.backward()
        fx[f"train.{tag}.thr"] = np.array(thr, dtype=np.float32)
        fx[f"train.{tag}.logits"] = out.detach().numpy()
        fx[f"train.{tag}.loss"] = np.array(loss.item())
        fx.update({f"train.{tag}." + k: v for k, v in grad_digest(model).items()})
    np.savez_compressed(os.path.join(HERE, "cfg1.npz"), **fx)

    # ---- F5a: nW=5 (the J=67 mapping), T=16, B=2 -------------------------
    T, nW, C, d0, nc, B = 16, 5, 2, 128, 7, 2
    model, hp, cfg = build_reference(Model, HWGATEParams, T=T, nW=nW, C=C, d0=d0, nc=nc, seed=12)
    x = torch.rand(B, T, nW * 16, C, generator=g)
    y = torch.randint(0, nc, (B,), generator=g)
    model.eval()
    with torch.no_grad():
        lo = model(x)
    model.train()
    thr = [0.3, 0.1, 0.5, 0.2, 0.07, 0.4, 0.25, 0.6]
    model.zero_grad()
    with forced_thresholds(thr):
        lt = model(x)
    loss = crit(lt, y)
    loss.backward()
    fy = {"x": x.numpy(), "y": y.numpy(), "cfg": np.array([T, nW, C, d0, nc, B, 12]),
          "eval.logits": lo.numpy(), "train.thr": np.array(thr, dtype=np.float32),
          "train.logits": lt.detach().numpy(), "train.loss": np.array(loss.item())}
    fy.update({"train." + k: v for k, v in grad_digest(model).items()})
    np.savez_compressed(os.path.join(HERE, "nw5.npz"), **fy)

    # ---- F5b: C=3, d0=256 (head_dim 128), nW=7, T=8, B=1 -----------------
    T, nW, C, d0, nc, B = 8, 7, 3, 256, 5, 1
    model, hp, cfg = build_reference(Model, HWGATEParams, T=T, nW=nW, C=C, d0=d0, nc=nc,
                                     seed=13, wstd=0.05)
    x = torch.rand(B, T, nW * 16, C, generator=g)
    y = torch.randint(0, nc, (B,), generator=g)
    model.eval()
    model.zero_grad()
    lo = model(x)
    loss = crit(lo, y)
    loss.backward()
    fz = {"x": x.numpy(), "y": y.numpy(), "cfg": np.array([T, nW, C, d0, nc, B, 13]),
          "wstd": np.array(0.05), "eval.logits": lo.detach().numpy(),
          "eval.loss": np.array(loss.item())}
    fz.update({"eval." + k: v for k, v in grad_digest(model).items()})
    np.savez_compressed(os.path.join(HERE, "hd128.npz"), **fz)

    # ---- F6: the BENCHMARKED shapes, one clip each (BASELINE configs[1] and configs[4]) -------------
    # eval-mode fwd+bwd (deterministic in the reference) + one train-mode run with injected thresholds
    for fname, (T, nW, C, d0, nc, B, seed, wstd) in {
            "cfg2_clip.npz": (128, 5, 2, 128, 2002, 1, 14, 0.05),
            "cfg5_clip.npz": (256, 7, 3, 256, 2002, 1, 15, 0.04)}.items():
        model, hp, cfg = build_reference(Model, HWGATEParams, T=T, nW=nW, C=C, d0=d0, nc=nc, seed=seed, wstd=wstd)
        x = torch.rand(B, T, nW * 16, C, generator=g)
        y = torch.randint(0, nc, (B,), generator=g)
        taps = {}
        handles = []
        blk = 0
        for i, layer in enumerate(model.layers):
            for j, b in enumerate(layer.blocks):
                handles.append(b.register_forward_hook(
                    lambda _m, _i, out, k=blk: taps.__setitem__(f"block{k}", out.detach())))
                blk += 1
        model.eval()
        model.zero_grad()
        lo = model(x)
        loss = crit(lo, y)
        loss.backward()
        for h in handles:
            h.remove()
        fw = {"x": x.numpy(), "y": y.numpy(), "cfg": np.array([T, nW, C, d0, nc, B, seed]), "wstd": np.array(wstd),
              "eval.logits": lo.detach().numpy(), "eval.loss": np.array(loss.item())}
        for k, v in taps.items():
            fw["eval." + k] = v[:, ::9, ::7, ::11].contiguous().numpy()
        fw.update({"eval." + k: v for k, v in grad_digest(model).items()})
        model.train()
        thr = [0.3, 0.1, 0.5, 0.2, 0.07, 0.4, 0.25, 0.6]
        model.zero_grad()
        with forced_thresholds(thr):
            lt = model(x)
        loss = crit(lt, y)
        loss.backward()
        fw.update({"train.thr": np.array(thr, dtype=np.float32), "train.logits": lt.detach().numpy(),
                   "train.loss": np.array(loss.item())})
        fw.update({"train." + k: v for k, v in grad_digest(model).items()})
        np.savez_compressed(os.path.join(HERE, fname), **fw)

    # ---- part gather (dataTransform.py:426-455) --------------------------
    from dataTransform import WindowCreate
    raw = np.random.RandomState(3).rand(6, 29, 2)
    np.savez_compressed(os.path.join(HERE, "window_create.npz"), raw=raw,
                        out=WindowCreate(6)(raw).astype(np.float32))
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
