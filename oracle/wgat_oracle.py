"""CPU oracle for the WGATE sibling model  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Functional restatement (plain torch CPU ops) of the arithmetic of the reference's
`hwgat/models/WGATE.py`: the windowed graph-attention ablation WITHOUT hierarchy.  An attention window is
one 16-joint body-part window over ALL T frames (T*16 tokens, WGATE.py:32-46); the (nW, T*16, T*16)
adjacency (model_params.py:209-228: same frame -> part graph, adjacent frames -> same joint, else 0)
enters as an ADDITIVE 0 / -10000 mask (WGATE.py:190, 97-100), there is no shift, no threshold, no
temporal merging; `depths` identical blocks at constant width.

This oracle evaluates the attention DENSELY over all T*16 keys, exactly as the reference does; the HIP
kernel exploits that exp(s - 10000 - max) underflows to exactly 0 in fp32.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this file.

Parity status: PINNED by `tests/golden/wgate_*.npz`, generated from the reference `Model` by
`tests/golden/make_fixtures_wgate.py` (tests/test_oracle_wgate_golden.py checks this file against them).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from .hwgat_oracle import (NEG_FILL, WINDOW, fourier_embed, gelu, layer_norm, part_adjacency,  # noqa: F401
                           sinusoid_table, smoothed_cross_entropy)


def band_adjacency(frames: int, n_windows: int, dtype=torch.float32) -> torch.Tensor:
    """(nW, T*16, T*16) block-tridiagonal 0/1 matrix (model_params.py:209-228)."""
    a, eye = part_adjacency(dtype), torch.eye(WINDOW, dtype=dtype)
    one = torch.zeros(frames, WINDOW, frames, WINDOW, dtype=dtype)
    for i in range(frames):
        one[i, :, i, :] = a
        if i + 1 < frames:
            one[i, :, i + 1, :] = eye
            one[i + 1, :, i, :] = eye
    one = one.reshape(frames * WINDOW, frames * WINDOW)
    return one.unsqueeze(0).repeat(n_windows, 1, 1).contiguous()


def additive_mask(adj: torch.Tensor) -> torch.Tensor:
    """WGATE.py:190: 0 -> -10000, 1 -> 0."""
    return adj.masked_fill(adj == 0, NEG_FILL).masked_fill(adj == 1, 0.0)


def to_windows(x: torch.Tensor) -> torch.Tensor:
    """(B,F,K,d) -> (B, nW, F*16, d); token t = frame*16 + joint (WGATE.py:32-46)."""
    B, F, K, d = x.shape
    nW = K // WINDOW
    return x.reshape(B, F, nW, WINDOW, d).transpose(1, 2).reshape(B, nW, F * WINDOW, d)


def from_windows(xw: torch.Tensor, frames: int) -> torch.Tensor:
    """inverse of `to_windows` (WGATE.py:50-65)."""
    B, nW, _, d = xw.shape
    return xw.reshape(B, nW, frames, WINDOW, d).transpose(1, 2).reshape(B, frames, nW * WINDOW, d)


def band_attention(q, k, v, mask, attn_keep=None):
    """MSA.forward core (WGATE.py:92-105) for q,k,v (B, nW, nH, T*16, hd); mask (nW, T*16, T*16) additive;
    attn_keep: None, or the attention-dropout factor (B, nW, nH, T*16, T*16) = mask / (1 - p) of nn.Dropout(attn_drop)
    on the probabilities (WGATE.py:81,103), injected."""
    hd = q.shape[-1]
    s = (q * hd ** -0.5) @ k.transpose(-2, -1)                 # :92-94
    s = s + mask[None, :, None]                                # :97-100
    p = torch.softmax(s, dim=-1)                               # :102
    a = p if attn_keep is None else p * attn_keep              # :103
    o = a @ v                                                  # :105
    B, nW, nH, n, _ = o.shape
    return o.transpose(2, 3).reshape(B, nW, n, nH * hd), p


class OracleWGAT:
    """Functional WGATE forward on a reference-keyed parameter dict (`layers.{i}.<...>`)."""

    def __init__(self, params: Dict[str, torch.Tensor], *, num_kps: int, temporal_dim: int, depths: int = 8,
                 num_heads: int = 8, use_pe: bool = True, adj: Optional[torch.Tensor] = None):
        self.p = params
        self.K, self.T, self.depths, self.heads, self.use_pe = num_kps, temporal_dim, depths, num_heads, use_pe
        dt = params["B"].dtype
        adj = adj if adj is not None else band_adjacency(temporal_dim, num_kps // WINDOW)
        self.mask = additive_mask(adj.to(dt))
        self.taps: Dict[str, torch.Tensor] = {}

    def block(self, x, i):                                     # PartAttentionBlock.forward, WGATE.py:150-160
        P, pre = self.p, f"layers.{i}."
        B, F, K, d = x.shape
        nH = self.heads
        hd = d // nH
        xw = to_windows(x)                                                     # :155
        xn = layer_norm(xw, P[pre + "norm1.weight"], P[pre + "norm1.bias"])    # :156
        qkv = xn @ P[pre + "attn.qkv.weight"].t() + P[pre + "attn.qkv.bias"]   # :88
        qkv = qkv.reshape(B, K // WINDOW, F * WINDOW, 3, nH, hd).permute(3, 0, 1, 4, 2, 5)
        o, prob = band_attention(qkv[0], qkv[1], qkv[2], self.mask)
        a = o @ P[pre + "attn.proj.weight"].t() + P[pre + "attn.proj.bias"]    # :106
        y = x + from_windows(a, F)                                             # :157-158
        h = layer_norm(y, P[pre + "norm2.weight"], P[pre + "norm2.bias"])
        h = gelu(h @ P[pre + "ff.fc1.weight"].t() + P[pre + "ff.fc1.bias"])
        h = h @ P[pre + "ff.fc2.weight"].t() + P[pre + "ff.fc2.bias"]
        return y + h, prob                                                     # :159

    def forward(self, x, tap: bool = False):
        P = self.p
        h = fourier_embed(x, P["B"])                                           # :241-243
        if self.use_pe:
            h = h + P["pos_encoder.pe"][:, :h.shape[1]]
        for i in range(self.depths):
            h, prob = self.block(h, i)
            if tap:
                self.taps[f"block{i}"] = h
        h = layer_norm(h, P["norm.weight"], P["norm.bias"])                    # :250
        feat = h.mean(dim=(1, 2))                                              # :251
        if tap:
            self.taps["feat"] = feat
        return feat @ P["head.weight"].t() + P["head.bias"]                    # :256


def param_shapes(*, kp_dim: int, temporal_dim: int, num_classes: int, embed_dim: int = 128, depths: int = 8,
                 ff_ratio: float = 2.0, use_pe: bool = True):
    """Ordered (name, shape) list of the reference `state_dict()` minus the derived `adj_mask` buffer."""
    d, hid = embed_dim, int(embed_dim * ff_ratio)
    out = [("B", (d // 2, kp_dim))]
    if use_pe:
        out.append(("pos_encoder.pe", (1, temporal_dim, 1, d)))
    for i in range(depths):
        pre = f"layers.{i}."
        out += [(pre + "norm1.weight", (d,)), (pre + "norm1.bias", (d,)),
                (pre + "attn.qkv.weight", (3 * d, d)), (pre + "attn.qkv.bias", (3 * d,)),
                (pre + "attn.proj.weight", (d, d)), (pre + "attn.proj.bias", (d,)),
                (pre + "norm2.weight", (d,)), (pre + "norm2.bias", (d,)),
                (pre + "ff.fc1.weight", (hid, d)), (pre + "ff.fc1.bias", (hid,)),
                (pre + "ff.fc2.weight", (d, hid)), (pre + "ff.fc2.bias", (d,))]
    out += [("norm.weight", (d,)), ("norm.bias", (d,)), ("head.weight", (num_classes, d)),
            ("head.bias", (num_classes,))]
    return out


def synth_params(seed: int, *, weight_std: float = 0.08, **cfg) -> Dict[str, torch.Tensor]:
    """Deterministic parameter set (numpy legacy MT19937 stream, one draw per tensor in `param_shapes`
    order; same conventions as hwgat_oracle.synth_params)."""
    import numpy as np
    rs = np.random.RandomState(seed)
    out: Dict[str, torch.Tensor] = {}
    for name, shape in param_shapes(**cfg):
        if name == "pos_encoder.pe":
            out[name] = sinusoid_table(cfg["temporal_dim"], cfg.get("embed_dim", 128))
            continue
        if name == "B":
            v = rs.standard_normal(shape) * 10.0
        elif name.endswith("norm1.weight") or name.endswith("norm2.weight") or name == "norm.weight":
            v = 1.0 + 0.1 * rs.standard_normal(shape)
        elif name.endswith(".bias"):
            v = 0.05 * rs.standard_normal(shape)
        else:
            v = weight_std * rs.standard_normal(shape)
        out[name] = torch.from_numpy(v.astype(np.float32))
    return out
