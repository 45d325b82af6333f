#!/usr/bin/env python3
"""Golden vectors for the HGATE sibling model, from the REFERENCE (development container only).

Run:  python tests/golden/make_fixtures_hgate.py        (needs /root/reference)

Imports `/root/reference/hwgat/models/HGATE.py` as-is (same `timm.trunc_normal_` alias as
make_fixtures.py: init only, overwritten before anything is recorded), loads the deterministic
`synth_params` set, runs seeded inputs and stores inputs + outputs.  Fixtures are data only.

  hgate_a.npz  T=128, B=4, C=2, d0=128, heads (2,4,8) -> head_dim 64; B*F is a multiple of 128 at
               every stage, so the backend's fused-linear path is the one under test
  hgate_b.npz  T=16, B=2, C=3, d0=128, heads (4,8,16) -> head_dim 32; ragged token counts
HGATE has no train-mode threshold, so with drop_rate 0 train() and eval() are the same function;
each fixture holds eval logits, strided activation taps and gradient digests of one backward pass.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/hwgat"
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from oracle import hwgat_oracle as O  # noqa: E402
from make_fixtures import grad_digest, sub  # noqa: E402


def import_reference():
    for name in ("timm", "timm.models", "timm.models.layers"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["timm.models.layers"].trunc_normal_ = torch.nn.init.trunc_normal_
    sys.path.insert(0, REF)
    from models.HGATE import Model                      # noqa
    from models.model_params import HGATEParams         # noqa
    from losses.SmoothCrossEntropy import SmoothedCrossEntropyLoss  # noqa
    return Model, HGATEParams, SmoothedCrossEntropyLoss


def build(Model, HGATEParams, *, T, C, d0, nc, heads, seed, wstd=0.08):
    hp = HGATEParams({"src_len": T, "num_class": nc}, C, torch.device("cpu"))
    hp.embed_dim = d0
    hp.num_heads = list(heads)
    hp.drop_rate = 0.0
    model = Model(*hp.get_model_params())
    cfg = dict(kp_dim=C, temporal_dim=T, num_classes=nc, embed_dim=d0, depths=tuple(hp.depths),
               ff_ratio=hp.ff_ratio, use_pe=hp.pe, num_kps=hp.num_kps, tp=hp.temporal_patch_size)
    synth = O.synth_params(seed, weight_std=wstd, **cfg)
    res = model.load_state_dict(synth, strict=False)
    assert not res.unexpected_keys, res.unexpected_keys
    assert all(k.endswith("attn_mask") for k in res.missing_keys), res.missing_keys
    return model, hp


def record(model, hp, x, y, crit, cfg_row):
    fx = {"x": x.numpy(), "y": y.numpy(), "adj": hp.adj_mat.numpy(), "cfg": np.array(cfg_row),
          "heads": np.array(hp.num_heads)}
    for k, v in model.state_dict().items():
        if k.endswith("attn_mask"):
            fx["mask." + k] = v.numpy().astype(np.uint8)
    taps = {}

    def hook(name):
        def fn(_m, _i, out):
            taps[name] = out.detach()
        return fn
    blk = 0
    for i, layer in enumerate(model.layers):
        for b in layer.blocks:
            b.register_forward_hook(hook(f"block{blk}"))
            blk += 1
        if layer.downsample is not None:
            layer.downsample.register_forward_hook(hook(f"merge{i}"))
    model.eval()
    with torch.no_grad():
        fx["eval.logits"] = model(x).numpy()
        fx["eval.feat"] = model.forward_features(x).numpy()
    for k, v in taps.items():
        fx["eval." + k] = sub(v)
    fx["eval.block1.full"] = taps["block1"][0, -2:].numpy()       # the wrapped block of a shifted layer
    model.zero_grad()
    loss = crit(model(x), y)
    loss.backward()
    fx["evalbwd.loss"] = np.array(loss.item())
    fx.update({"evalbwd." + k: v for k, v in grad_digest(model).items()})
    return fx


def main():
    Model, HGATEParams, Loss = import_reference()
    torch.manual_seed(1001)
    crit = Loss()
    g = torch.Generator().manual_seed(17)

    T, C, d0, nc, B, seed = 128, 2, 128, 10, 4, 21
    model, hp = build(Model, HGATEParams, T=T, C=C, d0=d0, nc=nc, heads=(2, 4, 8), seed=seed)
    x = torch.rand(B, T, hp.num_kps, C, generator=g)
    y = torch.randint(0, nc, (B,), generator=g)
    np.savez_compressed(os.path.join(HERE, "hgate_a.npz"),
                        **record(model, hp, x, y, crit, [T, hp.num_kps, C, d0, nc, B, seed]))

    T, C, d0, nc, B, seed = 16, 3, 128, 7, 2, 22
    model, hp = build(Model, HGATEParams, T=T, C=C, d0=d0, nc=nc, heads=(4, 8, 16), seed=seed)
    x = torch.rand(B, T, hp.num_kps, C, generator=g)
    y = torch.randint(0, nc, (B,), generator=g)
    np.savez_compressed(os.path.join(HERE, "hgate_b.npz"),
                        **record(model, hp, x, y, crit, [T, hp.num_kps, C, d0, nc, B, seed]))
    for f in sorted(os.listdir(HERE)):
        if f.startswith("hgate") and f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
