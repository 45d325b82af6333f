"""Shared helpers for the parity tests (oracle side only)."""
import os

import numpy as np
import torch

from oracle import hwgat_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_fixture(name):
    return dict(np.load(os.path.join(GOLDEN, name)))


def cfg_of(fx, **over):
    T, nW, C, d0, nc, B, seed = [int(v) for v in fx["cfg"]]
    cfg = dict(kp_dim=C, temporal_dim=T, num_classes=nc, embed_dim=d0,
               depths=(2, 2, 4), ff_ratio=2.0, use_pe=True, num_kps=nW * 16, tp=2)
    cfg.update(over)
    return cfg, seed, B


def oracle_from_fixture(fx, dtype=torch.float32):
    cfg, seed, _ = cfg_of(fx)
    wstd = float(fx["wstd"]) if "wstd" in fx else 0.08
    params = {k: v.to(dtype) for k, v in O.synth_params(seed, weight_std=wstd, **cfg).items()}
    model = O.OracleHWGAT(params, num_kps=cfg["num_kps"], temporal_dim=cfg["temporal_dim"])
    return model, params, cfg


def sub(t):
    return t[:, ::5, ::3, ::7]


def natural(t, d):
    """a block output in natural token order (B,F,K,d): the last block of a stage may hand it over already in the
    TemporalMerging layout (B,F/2,K,2d) (fc2 epilogue store, reference HWGATE.py:55-63) -- undo that for comparison"""
    if t.shape[-1] == d:
        return t
    B, f, K, d2 = t.shape
    assert d2 == 2 * d
    return t.view(B, f, K, 2, d).permute(0, 1, 3, 2, 4).reshape(B, 2 * f, K, d)


def rel_err(a, b):
    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def tie_free_threshold(p0, nominal, span=1.25):
    """A train-mode threshold (reference HWGATE.py:94-100) near `nominal` that no entry of the unmasked softmax `p0`
    (the fp64 oracle's, any shape) comes close to: the geometric centre of the widest relative gap between neighbouring
    values inside [nominal / span, nominal * span].  Returns (thr, margin) with margin = the relative distance from thr
    to the nearest probability.  With it the selector [p0 <= thr] cannot flip under rounding smaller than `margin`, so a
    kernel can be held to its arithmetic tolerance in train mode too."""
    lo, hi = nominal / span, nominal * span
    v = torch.as_tensor(p0, dtype=torch.float64).flatten()
    v = v[(v > lo) & (v < hi)].sort().values
    edges = torch.cat([torch.tensor([lo], dtype=torch.float64), v, torch.tensor([hi], dtype=torch.float64)])
    ratio = edges[1:] / edges[:-1]
    i = int(ratio.argmax())
    return float((edges[i] * edges[i + 1]).sqrt()), float(ratio[i].sqrt() - 1.0)


BF16_SELECTOR_BAND = 2.0 ** -6


def oracle_threshold_bracket(run, thresholds, band=BF16_SELECTOR_BAND):
    """Train-mode parity of a WHOLE bf16 model has a discontinuity no threshold can dodge: the selector [P0 <= thr] of
    reference HWGATE.py:94-100 is evaluated on scores formed from bf16 activations (relative error ~2^-8 per layer), so
    a probability within about `band` (relative) of the threshold may fall on either side, and with 10^5..10^7
    probabilities per block some always are that close.  Instead of waiving the assertions, the test BRACKETS the flips:
    `run(thresholds)` -> (logits, {name: grad}) is evaluated by the fp64 oracle at thr, thr * (1 + band) (every entry
    of the band kept) and thr * (1 - band) (every entry of the band dropped).  An implementation whose selector differs
    from the oracle's only inside the band must land within its arithmetic tolerance + the distance the band itself can
    move the result.  Returns (ref_out, ref_grads, width_out, {name: width}) with width = relative L2 distance between
    the two bracket ends; the caller adds 2 x width to its tolerance and asserts that width is small (else the test
    would have no teeth)."""
    ref_out, ref_g = run(list(thresholds))
    hi_out, hi_g = run([t * (1.0 + band) for t in thresholds])
    lo_out, lo_g = run([t * (1.0 - band) for t in thresholds])
    w_out = rel_err(hi_out, lo_out)
    w_g = {k: rel_err(hi_g[k], lo_g[k]) for k in ref_g}
    return ref_out, ref_g, w_out, w_g


def probe_vectors(name, n, k=4):
    """the k fixed +-1 vectors tests/golden/make_fixtures.py::probe_vectors projects a gradient on"""
    import zlib
    rs = np.random.RandomState(zlib.crc32(name.encode()) & 0x7FFFFFFF)
    return rs.randint(0, 2, size=(k, n)).astype(np.float64) * 2.0 - 1.0


def grad_digest_check(named_grads, fx, prefix, tol):
    """compare {name: grad} with the `gh.`/`gn.` digests stored in a fixture"""
    worst = 0.0
    n = 0
    for name, g in named_grads.items():
        key = prefix + "gh." + name
        if key not in fx:
            continue
        n += 1
        gd = g.detach().double().flatten().cpu()
        ref_norm, ref_sum = fx[prefix + "gn." + name]
        scale = max(ref_norm, 1e-12)
        e1 = abs(gd.norm().item() - ref_norm) / scale
        e2 = (gd[:48] - torch.from_numpy(fx[key]).double()).norm().item() / \
            max(np.linalg.norm(fx[key]), 1e-3 * scale / max(gd.numel(), 1) ** 0.5, 1e-30)
        worst = max(worst, e1, e2)
        assert e1 < tol, (name, "norm", e1)
        assert e2 < tol * 10, (name, "head", e2)
        pkey = prefix + "gp." + name
        if pkey in fx:          # +-1 projections of the WHOLE gradient: catches misplaced entries anywhere
            proj = probe_vectors(name, gd.numel()) @ gd.numpy()
            e3 = float(np.abs(proj - fx[pkey]).max()) / scale
            worst = max(worst, e3)
            assert e3 < tol * 10, (name, "projection", e3)
    assert n > 0
    return worst


def hgate_oracle_from_fixture(fx, dtype=torch.float32):
    """(OracleHGAT, params, cfg) of a tests/golden/hgate_*.npz fixture"""
    from oracle import hgat_oracle as OH
    T, K, C, d0, nc, B, seed = [int(v) for v in fx["cfg"]]
    cfg = dict(kp_dim=C, temporal_dim=T, num_classes=nc, embed_dim=d0, depths=(2, 2, 4), ff_ratio=2.0,
               use_pe=True, num_kps=K, tp=2)
    params = {k: v.to(dtype) for k, v in O.synth_params(seed, weight_std=0.08, **cfg).items()}
    model = OH.OracleHGAT(params, num_kps=K, temporal_dim=T, num_heads=[int(h) for h in fx["heads"]])
    return model, params, cfg


def wgate_oracle_from_fixture(fx, dtype=torch.float32):
    """(OracleWGAT, params, cfg) of a tests/golden/wgate_*.npz fixture"""
    from oracle import wgat_oracle as OW
    T, nW, C, d0, nc, B, heads, depths, seed = [int(v) for v in fx["cfg"]]
    cfg = dict(kp_dim=C, temporal_dim=T, num_classes=nc, embed_dim=d0, depths=depths, ff_ratio=2.0, use_pe=True)
    params = {k: v.to(dtype) for k, v in OW.synth_params(seed, **cfg).items()}
    model = OW.OracleWGAT(params, num_kps=nW * 16, temporal_dim=T, depths=depths, num_heads=heads)
    return model, params, dict(cfg, num_kps=nW * 16, num_heads=heads)
