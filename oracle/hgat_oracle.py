"""CPU oracle for the HGATE sibling model  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Functional restatement (plain torch CPU ops) of the arithmetic of the reference's
`hwgat/models/HGATE.py`: the hierarchical graph-attention ablation WITHOUT body-part windows.  An
attention block is `tp` consecutive frames x ALL K joints (HGATE.py:30-37), the adjacency is one
(tp*K, tp*K) matrix shared by every block (HGATE.py:100-102, model_params.py:460-483) and there is no
train-mode threshold drop (HGATE.py:84-108 has no `self.training` branch).  Everything else -- Fourier
embedding, PE, LayerNorms, FFN, TemporalMerging, final norm / pool / head -- is the HWGATE arithmetic
and is imported from `hwgat_oracle`.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this file.

Parity status: PINNED by `tests/golden/hgate_*.npz`, generated from the reference `Model` by
`tests/golden/make_fixtures_hgate.py` (tests/test_oracle_golden.py checks this file against them).
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import torch

from .hwgat_oracle import (NEG_FILL, fourier_embed, gelu, layer_norm, param_shapes,  # noqa: F401
                           sinusoid_table, smoothed_cross_entropy, synth_params)

# 29-joint skeleton of HGATEParams (model_params.py:424-457): 9 upper-body joints, then two 10-joint
# hands with the same internal graph, rooted at joints 9 and 19 and hung from the wrists 7 and 8.
BODY_EDGES = ((2, 0), (1, 0), (0, 3), (0, 4), (3, 5), (4, 6), (5, 7), (6, 8), (7, 9), (8, 19))
HAND_EDGES = ((0, 1), (0, 2), (2, 3), (2, 4), (4, 5), (0, 4), (4, 6), (0, 6), (6, 7), (6, 8), (0, 8), (8, 9))
HAND_ROOTS = (9, 19)
NUM_JOINTS = 29


def skeleton_edges():
    return list(BODY_EDGES) + [(r + a, r + b) for r in HAND_ROOTS for a, b in HAND_EDGES]


def skeleton_adjacency(dtype=torch.float32) -> torch.Tensor:
    """29x29 symmetric joint graph with unit diagonal (model_params.py:478-483)."""
    a = torch.eye(NUM_JOINTS, dtype=dtype)
    for i, j in skeleton_edges():
        a[i, j] = 1
        a[j, i] = 1
    return a


def block_adjacency(joint_adj: Optional[torch.Tensor] = None, tp: int = 2, dtype=torch.float32) -> torch.Tensor:
    """(tp*K, tp*K): same frame -> joint graph, adjacent frames -> same joint only, else nothing
    (model_params.py:460-476)."""
    a = skeleton_adjacency(dtype) if joint_adj is None else joint_adj.to(dtype)
    K = a.shape[0]
    eye, zero = torch.eye(K, dtype=dtype), torch.zeros(K, K, dtype=dtype)
    rows = [torch.cat([a if i == j else (eye if abs(i - j) == 1 else zero) for j in range(tp)], dim=1)
            for i in range(tp)]
    return torch.cat(rows, dim=0)


def block_shift_mask(frames: int, K: int, tp: int = 2, shift: int = 1, dtype=torch.float32) -> torch.Tensor:
    """0/1 mask of an odd block, (f, tp*K, tp*K) (HGATE.py:154-172): frame labels 0 / 1 / 2 for
    [0, F-tp) / [F-tp, F-shift) / [F-shift, F); tokens attend iff labels are equal."""
    label = torch.zeros(frames, dtype=dtype)
    label[frames - tp:frames - shift] = 1
    label[frames - shift:] = 2
    f = frames // tp
    tok = label.view(f, tp, 1).expand(f, tp, K).reshape(f, tp * K)    # token t = tp_idx*K + joint
    return (tok.unsqueeze(1) == tok.unsqueeze(2)).to(dtype)


def block_attention(q, k, v, adj, smask=None, attn_keep=None):
    """MSA.forward core (HGATE.py:91-107) for q,k,v of shape (B, f, nH, tp*K, hd);
    adj (tp*K, tp*K) 0/1; smask (f, tp*K, tp*K) 0/1 or None; attn_keep: None, or the attention-dropout factor
    (B, f, nH, tp*K, tp*K) = mask / (1 - p) that nn.Dropout(attn_drop) applies to the probabilities (HGATE.py:78,106) --
    injected, so that parity does not depend on a random stream.  Returns o (B, f, tp*K, nH*hd), probs."""
    hd = q.shape[-1]
    s = (q * hd ** -0.5) @ k.transpose(-2, -1)                 # :91-93
    if smask is not None:
        s = s * smask[None, :, None]                           # :96-98
    s = s * adj                                                # :100-102
    s = s.masked_fill(s == 0, NEG_FILL)                        # :104
    p = torch.softmax(s, dim=-1)                               # :105
    a = p if attn_keep is None else p * attn_keep              # :106
    o = a @ v                                                  # :108
    B, f, nH, n, _ = o.shape
    return o.transpose(2, 3).reshape(B, f, n, nH * hd), p


class OracleHGAT:
    """Functional HGATE forward on a reference-keyed parameter dict (same key names as HWGATE)."""

    def __init__(self, params: Dict[str, torch.Tensor], *, num_kps: int = NUM_JOINTS, temporal_dim: int,
                 depths: Sequence[int] = (2, 2, 4), num_heads: Sequence[int] = (2, 4, 8), tp: int = 2,
                 use_pe: bool = True, adj: Optional[torch.Tensor] = None):
        self.p = params
        self.K, self.T, self.tp = num_kps, temporal_dim, tp
        self.depths, self.heads = list(depths), list(num_heads)
        self.use_pe = use_pe
        dt = params["B"].dtype
        self.adj = (adj if adj is not None else block_adjacency(None, tp)).to(dt)
        assert self.adj.shape == (tp * num_kps, tp * num_kps)
        self.taps: Dict[str, torch.Tensor] = {}

    # one GraphAttentionBlock (HGATE.py:175-213) in natural token order
    def block(self, x, i, j, nH):
        P, pre = self.p, f"layers.{i}.blocks.{j}."
        B, F, K, d = x.shape
        f, tp = F // self.tp, self.tp
        shifted = (j % 2 == 1)
        xs = torch.roll(x, shifts=-1, dims=1) if shifted else x                # :184-188
        xb = xs.reshape(B, f, tp * K, d)                                       # :190 (block_partition)
        xn = layer_norm(xb, P[pre + "norm1.weight"], P[pre + "norm1.bias"])    # :192
        qkv = xn @ P[pre + "attn.qkv.weight"].t() + P[pre + "attn.qkv.bias"]   # :86
        hd = d // nH
        qkv = qkv.reshape(B, f, tp * K, 3, nH, hd).permute(3, 0, 1, 4, 2, 5)
        sm = block_shift_mask(F, K, tp, 1, x.dtype) if shifted else None
        o, prob = block_attention(qkv[0], qkv[1], qkv[2], self.adj, sm)
        a = o @ P[pre + "attn.proj.weight"].t() + P[pre + "attn.proj.bias"]    # :109
        a = a.reshape(B, F, K, d)                                              # :196 (block_reverse)
        if shifted:
            a = torch.roll(a, shifts=1, dims=1)                                # :199-202
        y = x + a                                                              # :204
        h = layer_norm(y, P[pre + "norm2.weight"], P[pre + "norm2.bias"])
        h = gelu(h @ P[pre + "ff.fc1.weight"].t() + P[pre + "ff.fc1.bias"])
        h = h @ P[pre + "ff.fc2.weight"].t() + P[pre + "ff.fc2.bias"]
        return y + h, prob                                                     # :206

    def forward(self, x, tap: bool = False):
        """x: (B,T,K,C) -> logits.  Train mode without dropout is the same function (no threshold)."""
        P = self.p
        h = fourier_embed(x, P["B"])                                           # :329-331
        if self.use_pe:
            h = h + P["pos_encoder.pe"][:, :h.shape[1]]
        if tap:
            self.taps["pe"] = h
        blk = 0
        for i, depth in enumerate(self.depths):
            for j in range(depth):
                h, prob = self.block(h, i, j, self.heads[i])
                if tap:
                    self.taps[f"block{blk}"] = h
                    self.taps[f"prob{blk}"] = prob
                blk += 1
            if i < len(self.depths) - 1:                                       # :49-62
                B, F, K, d = h.shape
                h = h.reshape(B, F // self.tp, self.tp, K, d).transpose(2, 3).reshape(B, F // self.tp, K, self.tp * d)
                if tap:
                    self.taps[f"merge{i}"] = h
        h = layer_norm(h, P["norm.weight"], P["norm.bias"])                    # :339
        feat = h.mean(dim=(1, 2))                                              # :340
        if tap:
            self.taps["feat"] = feat
        return feat @ P["head.weight"].t() + P["head.bias"]                    # :345
