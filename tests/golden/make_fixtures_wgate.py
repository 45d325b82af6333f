#!/usr/bin/env python3
"""Golden vectors for the WGATE sibling model, from the REFERENCE (development container only).

Run:  python tests/golden/make_fixtures_wgate.py        (needs /root/reference)

Imports `/root/reference/hwgat/models/WGATE.py` as-is (same `timm.trunc_normal_` alias as
make_fixtures.py: init only, overwritten before anything is recorded), loads the deterministic
`oracle.wgat_oracle.synth_params` set, runs seeded inputs and stores inputs + outputs (data only).

  wgate_a.npz  T=32, K=64 (4 part windows), B=4, C=2, d=128, 8 heads (head_dim 16), 8 blocks;
               B*T*K is a multiple of 128: the backend's fused-linear path
  wgate_b.npz  T=6, K=48 (3 windows), B=3, C=3, d=128, 4 heads (head_dim 32), 2 blocks; ragged
WGATE has no train-only arithmetic besides dropout, so with drop_rate 0 train() == eval().
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

from oracle import wgat_oracle as OW  # noqa: E402
from make_fixtures import grad_digest  # noqa: E402


def import_reference():
    for name in ("timm", "timm.models", "timm.models.layers"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["timm.models.layers"].trunc_normal_ = torch.nn.init.trunc_normal_
    sys.path.insert(0, REF)
    from models.WGATE import Model                      # noqa
    from models.model_params import WGATEParams         # noqa
    from losses.SmoothCrossEntropy import SmoothedCrossEntropyLoss  # noqa
    return Model, WGATEParams, SmoothedCrossEntropyLoss


def build(Model, WGATEParams, *, T, nW, C, d0, nc, heads, depths, seed):
    hp = WGATEParams({"src_len": T, "num_class": nc}, C, torch.device("cpu"))
    hp.num_kps = nW * 16
    hp.embed_dim, hp.num_heads, hp.depths, hp.drop_rate = d0, heads, depths, 0.0
    hp.edges = [hp.edges[0]] * nW                # the 4 shipped lists are identical
    hp.adj_mat = torch.tensor(hp.get_adj_mat(), dtype=torch.float32)
    model = Model(*hp.get_model_params())
    cfg = dict(kp_dim=C, temporal_dim=T, num_classes=nc, embed_dim=d0, depths=depths, ff_ratio=hp.ff_ratio,
               use_pe=hp.pe)
    synth = OW.synth_params(seed, **cfg)
    res = model.load_state_dict(synth, strict=False)
    assert not res.unexpected_keys, res.unexpected_keys
    assert res.missing_keys == ["adj_mask"], res.missing_keys
    return model, hp


def sub(t):
    return t[:, ::3, ::5, ::7].contiguous().numpy()


def record(model, hp, x, y, crit, cfg_row):
    fx = {"x": x.numpy(), "y": y.numpy(), "cfg": np.array(cfg_row),
          "adj_w0": hp.adj_mat[0].numpy().astype(np.uint8),
          "adj_mask_w0_head": model.state_dict()["adj_mask"][0, :48, :48].numpy()}
    assert all(torch.equal(hp.adj_mat[0], hp.adj_mat[w]) for w in range(hp.adj_mat.shape[0]))
    taps = {}

    def hook(name):
        def fn(_m, _i, out):
            taps[name] = out.detach()
        return fn
    for i, layer in enumerate(model.layers):
        layer.register_forward_hook(hook(f"block{i}"))
    model.eval()
    with torch.no_grad():
        fx["eval.logits"] = model(x).numpy()
        fx["eval.feat"] = model.forward_features(x).numpy()
    for k, v in taps.items():
        fx["eval." + k] = sub(v)
    fx["eval.block0.full"] = taps["block0"][0, :3].numpy()        # first frames: the clipped band edge
    fx["eval.block0.tail"] = taps["block0"][0, -2:].numpy()
    model.zero_grad()
    loss = crit(model(x), y)
    loss.backward()
    fx["evalbwd.loss"] = np.array(loss.item())
    fx.update({"evalbwd." + k: v for k, v in grad_digest(model).items()})
    return fx


def main():
    Model, WGATEParams, Loss = import_reference()
    torch.manual_seed(1001)
    crit = Loss()
    g = torch.Generator().manual_seed(23)

    T, nW, C, d0, nc, B, heads, depths, seed = 32, 4, 2, 128, 10, 4, 8, 8, 31
    model, hp = build(Model, WGATEParams, T=T, nW=nW, C=C, d0=d0, nc=nc, heads=heads, depths=depths, seed=seed)
    x = torch.rand(B, T, nW * 16, C, generator=g)
    y = torch.randint(0, nc, (B,), generator=g)
    np.savez_compressed(os.path.join(HERE, "wgate_a.npz"),
                        **record(model, hp, x, y, crit, [T, nW, C, d0, nc, B, heads, depths, seed]))

    T, nW, C, d0, nc, B, heads, depths, seed = 6, 3, 3, 128, 7, 3, 4, 2, 32
    model, hp = build(Model, WGATEParams, T=T, nW=nW, C=C, d0=d0, nc=nc, heads=heads, depths=depths, seed=seed)
    x = torch.rand(B, T, nW * 16, C, generator=g)
    y = torch.randint(0, nc, (B,), generator=g)
    np.savez_compressed(os.path.join(HERE, "wgate_b.npz"),
                        **record(model, hp, x, y, crit, [T, nW, C, d0, nc, B, heads, depths, seed]))
    for f in sorted(os.listdir(HERE)):
        if f.startswith("wgate") and f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
