"""CPU oracle for the HWGAT hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A from-scratch, purely functional restatement (plain torch CPU ops, any float
dtype) of the arithmetic of the reference model `hwgat/models/HWGATE.py`.
Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg
may import this file; the product package (`sl-hwgat_amd/`) never does and
fails loudly when its HIP library is missing.

Parity status: PINNED.  `tests/golden/make_fixtures.py` imports the reference
`Model` in the development container and stores its outputs; the CPU test
suite checks this oracle against those fixtures (tests/test_oracle_golden.py).

Everything works on a flat `params` dict that uses the reference's
`state_dict()` key names, so reference checkpoints plug in
directly.  All activations stay in the natural token order (B, F, K, d):
window partition / reverse / roll are pure index permutations
(HWGATE.py:30-47, 197-215) and are expressed here as views.

Reference lines restated by each function are cited in its docstring.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import torch

NEG_FILL = -10000.0          # HWGATE.py:110
WINDOW = 16                  # model_params.py:254
LN_EPS = 1e-5                # nn.LayerNorm default (HWGATE.py:162,166,327)

# 25-edge part graph over the 16 slots of a part window
# (3 head, 3 arm, 10 hand slots) -- model_params.py:261-287; the four lists
# there are identical, so one table serves every part window.
PART_EDGES = (
    (0, 1), (0, 2), (0, 3), (3, 4), (4, 5), (5, 6), (6, 7), (6, 8), (8, 9),
    (8, 10), (6, 10), (10, 11), (10, 12), (6, 12), (12, 13), (12, 14),
    (14, 15), (6, 14), (7, 9), (9, 11), (11, 13), (13, 15), (7, 15), (7, 11),
    (7, 13),
)


# ---------------------------------------------------------------- adjacency
def part_adjacency(dtype=torch.float32) -> torch.Tensor:
    """16x16 symmetric part graph with unit diagonal (model_params.py:394-400)."""
    a = torch.eye(WINDOW, dtype=dtype)
    for i, j in PART_EDGES:
        a[i, j] = 1
        a[j, i] = 1
    return a


def window_adjacency(n_windows: int, tp: int = 2, dtype=torch.float32) -> torch.Tensor:
    """(nW, tp*16, tp*16) block matrix: same frame -> part graph, adjacent
    frames -> identity (same joint), else zero (model_params.py:373-392)."""
    a = part_adjacency(dtype)
    eye = torch.eye(WINDOW, dtype=dtype)
    zero = torch.zeros(WINDOW, WINDOW, dtype=dtype)
    rows = []
    for i in range(tp):
        rows.append(torch.cat([a if i == j else (eye if abs(i - j) == 1 else zero)
                               for j in range(tp)], dim=1))
    one = torch.cat(rows, dim=0)
    return one.unsqueeze(0).repeat(n_windows, 1, 1).contiguous()


def shift_mask(frames: int, n_windows: int, tp: int = 2, shift: int = 1,
               dtype=torch.float32) -> torch.Tensor:
    """Swin-style 0/1 mask of an odd block, (f*nW, tp*16, tp*16)
    (HWGATE.py:169-187).  Frame labels: 0 for all but the last tp frames,
    1 for frames [F-tp, F-shift), 2 for the last `shift` frames; a pair of
    tokens may attend iff their labels are equal."""
    label = torch.zeros(frames, dtype=dtype)
    label[frames - tp:frames - shift] = 1
    label[frames - shift:] = 2
    f = frames // tp
    # token t = tp_idx*16 + j of window (fi, wi) sits in frame fi*tp + tp_idx
    tok = label.view(f, tp, 1).expand(f, tp, WINDOW).reshape(f, tp * WINDOW)
    m = (tok.unsqueeze(1) == tok.unsqueeze(2)).to(dtype)          # (f, 32, 32)
    return m.unsqueeze(1).expand(f, n_windows, tp * WINDOW, tp * WINDOW) \
            .reshape(f * n_windows, tp * WINDOW, tp * WINDOW).contiguous()


# ---------------------------------------------------------------- embedding
def sinusoid_table(max_len: int, d_model: int, dtype=torch.float32) -> torch.Tensor:
    """(1, max_len, 1, d) table of HWGATE.py:16-22 (computed in fp32 like the
    reference, then cast)."""
    pe = torch.zeros(max_len, d_model)
    pos = torch.arange(0, max_len).unsqueeze(1)
    div = torch.exp(torch.arange(0, d_model, 2) * -(math.log(10000.0) / d_model))
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe.view(1, max_len, 1, d_model).to(dtype)


def fourier_embed(x: torch.Tensor, bmat: torch.Tensor) -> torch.Tensor:
    """HWGATE.py:343-344: cat[sin, cos]((2*pi*x) @ B^T)."""
    proj = (2.0 * math.pi * x) @ bmat.t()
    return torch.cat([proj.sin(), proj.cos()], dim=-1)


# ---------------------------------------------------------------- pieces
def layer_norm(x, w, b):
    return torch.nn.functional.layer_norm(x, (x.shape[-1],), w, b, LN_EPS)


def to_windows(x: torch.Tensor, tp: int = 2) -> torch.Tensor:
    """(B,F,K,d) -> (B, f, nW, tp*16, d) view-copy; token t = tp_idx*16 + j
    (HWGATE.py:30-36)."""
    B, F, K, d = x.shape
    f, nW = F // tp, K // WINDOW
    return x.reshape(B, f, tp, nW, WINDOW, d).permute(0, 1, 3, 2, 4, 5) \
            .reshape(B, f, nW, tp * WINDOW, d)


def from_windows(xw: torch.Tensor, tp: int = 2) -> torch.Tensor:
    """inverse of `to_windows` (HWGATE.py:39-47)."""
    B, f, nW, _, d = xw.shape
    return xw.reshape(B, f, nW, tp, WINDOW, d).permute(0, 1, 3, 2, 4, 5) \
             .reshape(B, f * tp, nW * WINDOW, d)


def window_attention(q, k, v, adj, smask=None, thr: Optional[float] = None, attn_keep=None):
    """Steps 1-8 of MSA.forward (HWGATE.py:89-114) for already-projected
    q,k,v of shape (B, f, nW, nH, 32, hd).

    adj: (nW, 32, 32) 0/1; smask: (f, nW, 32, 32) 0/1 or None;
    thr: None for eval mode, else the train-mode probability threshold;
    attn_keep: None, or the attention-dropout factor (B, f, nW, nH, 32, 32) = mask / (1 - p)
    that nn.Dropout(attn_drop) applies to the probabilities (HWGATE.py:78,112) -- injected,
    like the thresholds, so that parity does not depend on a random stream.
    Returns o (B, f, nW, 32, nH*hd) and the final (undropped) probabilities."""
    hd = q.shape[-1]
    s = (q * hd ** -0.5) @ k.transpose(-2, -1)                 # :89-91
    if thr is not None:                                        # :94-100
        p0 = torch.softmax(s.detach(), dim=-1)
        keep = (~(p0 > torch.tensor(thr, dtype=p0.dtype))).to(s.dtype)
        s = s * keep
    if smask is not None:                                      # :102-104
        s = s * smask[None, :, :, None]
    s = s * adj[None, None, :, None]                           # :106-108
    s = s.masked_fill(s == 0, NEG_FILL)                        # :110
    p = torch.softmax(s, dim=-1)                               # :111
    a = p if attn_keep is None else p * attn_keep              # :112
    o = a @ v                                                  # :114
    B, f, nW, nH, n, _ = o.shape
    return o.permute(0, 1, 2, 4, 3, 5).reshape(B, f, nW, n, nH * hd), p


def gelu(x):
    return torch.nn.functional.gelu(x)        # exact-erf, nn.GELU default


# ---------------------------------------------------------------- model
class OracleHWGAT:
    """Functional HWGAT forward driven by a reference-keyed parameter dict.

    Train-mode PARITY is defined with dropout disabled and thresholds injected
    (SURVEY.md §7 hard parts), so every test leaves `drop_rate` at 0.  A non-zero
    `drop_rate` applies `torch.nn.functional.dropout` at the reference's four
    sites (HWGATE.py:27, :116, :133, :135) and exists only so that `bench.py`'s
    CPU baseline can time the reference's DEFAULT train mode (drop 0.1), whose
    cost on a CPU is dominated by the mask generation."""

    def __init__(self, params: Dict[str, torch.Tensor], *, num_kps: int,
                 temporal_dim: int, depths: Sequence[int] = (2, 2, 4),
                 num_heads: Sequence[int] = (2, 4, 8), tp: int = 2,
                 use_pe: bool = True, adj: Optional[torch.Tensor] = None,
                 drop_rate: float = 0.0):
        self.drop_rate = float(drop_rate)
        self.p = params
        self.K, self.T, self.tp = num_kps, temporal_dim, tp
        self.depths, self.heads = list(depths), list(num_heads)
        self.use_pe = use_pe
        self.nW = num_kps // WINDOW
        dt = params["B"].dtype
        self.adj = (adj if adj is not None else window_adjacency(self.nW, tp)).to(dt)
        self.taps: Dict[str, torch.Tensor] = {}

    def _drop(self, t):
        return torch.nn.functional.dropout(t, self.drop_rate, True) if self.drop_rate > 0.0 else t

    # one PartAttentionBlock (HWGATE.py:189-221) in natural token order
    def block(self, x, i, j, nH, thr):
        P, pre = self.p, f"layers.{i}.blocks.{j}."
        B, F, K, d = x.shape
        f, nW, tp = F // self.tp, self.nW, self.tp
        shifted = (j % 2 == 1)
        xs = torch.roll(x, shifts=-1, dims=1) if shifted else x           # :197-200
        xw = to_windows(xs, tp)                                           # :201
        xn = layer_norm(xw, P[pre + "norm1.weight"], P[pre + "norm1.bias"])   # :203
        qkv = xn @ P[pre + "attn.qkv.weight"].t() + P[pre + "attn.qkv.bias"]  # :86
        hd = d // nH
        qkv = qkv.reshape(B, f, nW, tp * WINDOW, 3, nH, hd).permute(4, 0, 1, 2, 5, 3, 6)
        sm = None
        if shifted:
            sm = shift_mask(F, nW, tp, 1, x.dtype).view(f, nW, tp * WINDOW, tp * WINDOW)
        o, prob = window_attention(qkv[0], qkv[1], qkv[2], self.adj, sm, thr)
        a = self._drop(o @ P[pre + "attn.proj.weight"].t() + P[pre + "attn.proj.bias"])   # :115-116
        a = from_windows(a, tp)                                           # :207
        if shifted:
            a = torch.roll(a, shifts=1, dims=1)                           # :210-211
        y = x + a                                                         # :217
        h = layer_norm(y, P[pre + "norm2.weight"], P[pre + "norm2.bias"])
        h = self._drop(gelu(h @ P[pre + "ff.fc1.weight"].t() + P[pre + "ff.fc1.bias"]))   # :131-133
        h = self._drop(h @ P[pre + "ff.fc2.weight"].t() + P[pre + "ff.fc2.bias"])         # :134-135
        return y + h, prob                                                # :219

    def forward(self, x, thresholds: Optional[List[float]] = None, tap: bool = False):
        """x: (B,T,K,C).  thresholds: None = eval mode; else one float per
        block in execution order (train mode, HWGATE.py:94-100)."""
        P = self.p
        h = fourier_embed(x, P["B"])                                      # :343-345
        if tap:
            self.taps["embed"] = h
        if self.use_pe:
            h = self._drop(h + P["pos_encoder.pe"][:, :h.shape[1]])       # :26-27
        if tap:
            self.taps["pe"] = h
        blk = 0
        for i, depth in enumerate(self.depths):
            for j in range(depth):
                thr = None if thresholds is None else thresholds[blk]
                h, prob = self.block(h, i, j, self.heads[i], thr)
                if tap:
                    self.taps[f"block{blk}"] = h
                    self.taps[f"prob{blk}"] = prob
                blk += 1
            if i < len(self.depths) - 1:                                  # :55-63
                B, F, K, d = h.shape
                h = h.reshape(B, F // self.tp, self.tp, K, d).transpose(2, 3) \
                     .reshape(B, F // self.tp, K, self.tp * d)
                if tap:
                    self.taps[f"merge{i}"] = h
        h = layer_norm(h, P["norm.weight"], P["norm.bias"])               # :353
        feat = h.mean(dim=(1, 2))                                         # :354
        if tap:
            self.taps["feat"] = feat
        return feat @ P["head.weight"].t() + P["head.bias"]               # :359


def smoothed_cross_entropy(logits, target, eps: float = 0.01):
    """losses/SmoothCrossEntropy.py:35-39."""
    lp = torch.log_softmax(logits, dim=-1)
    nll = -lp.gather(-1, target.unsqueeze(1)).squeeze(1)
    return ((1.0 - eps) * nll + eps * (-lp.mean(dim=-1))).mean()


# ------------------------------------------------- deterministic parameters
def param_shapes(*, kp_dim: int, temporal_dim: int, num_classes: int,
                 embed_dim: int = 128, depths=(2, 2, 4), ff_ratio: float = 2.0,
                 use_pe: bool = True, num_kps: int = 64, tp: int = 2):
    """Ordered (name, shape) list equal to the reference `state_dict()`
    minus the derived `attn_mask` buffers."""
    out = [("B", (embed_dim // 2, kp_dim))]
    if use_pe:
        out.append(("pos_encoder.pe", (1, temporal_dim, 1, embed_dim)))
    for i, depth in enumerate(depths):
        d = embed_dim * 2 ** i
        hid = int(d * ff_ratio)
        for j in range(depth):
            pre = f"layers.{i}.blocks.{j}."
            out += [(pre + "norm1.weight", (d,)), (pre + "norm1.bias", (d,)),
                    (pre + "attn.qkv.weight", (3 * d, d)), (pre + "attn.qkv.bias", (3 * d,)),
                    (pre + "attn.proj.weight", (d, d)), (pre + "attn.proj.bias", (d,)),
                    (pre + "norm2.weight", (d,)), (pre + "norm2.bias", (d,)),
                    (pre + "ff.fc1.weight", (hid, d)), (pre + "ff.fc1.bias", (hid,)),
                    (pre + "ff.fc2.weight", (d, hid)), (pre + "ff.fc2.bias", (d,))]
    dl = embed_dim * 2 ** (len(depths) - 1)
    out += [("norm.weight", (dl,)), ("norm.bias", (dl,)),
            ("head.weight", (num_classes, dl)), ("head.bias", (num_classes,))]
    return out


def synth_params(seed: int, *, weight_std: float = 0.08, **cfg) -> Dict[str, torch.Tensor]:
    """Deterministic, platform-independent parameter set (numpy legacy
    MT19937 stream, one draw per tensor in `param_shapes` order).  Used so
    that fixtures do not have to carry a 40 MB state_dict: the fixture
    generator loads these into the reference model, the tests regenerate
    them.  Weights are larger than the reference's init (std .02) on purpose:
    they make the softmax non-uniform, which makes parity checks sharper."""
    import numpy as np
    rs = np.random.RandomState(seed)
    out: Dict[str, torch.Tensor] = {}
    for name, shape in param_shapes(**cfg):
        if name == "pos_encoder.pe":
            out[name] = sinusoid_table(cfg["temporal_dim"], cfg.get("embed_dim", 128))
            continue
        if name == "B":
            v = rs.standard_normal(shape) * 10.0          # HWGATE.py:297-298
        elif name.endswith("norm1.weight") or name.endswith("norm2.weight") or name == "norm.weight":
            v = 1.0 + 0.1 * rs.standard_normal(shape)
        elif name.endswith(".bias"):
            v = 0.05 * rs.standard_normal(shape)
        else:
            v = weight_std * rs.standard_normal(shape)
        out[name] = torch.from_numpy(v.astype(np.float32))
    return out
