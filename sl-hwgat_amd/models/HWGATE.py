"""HWGAT (hierarchical windowed graph attention) -- MI355X-native backend.

Drop-in for the reference's `hwgat/models/HWGATE.py`: class `Model` takes the
same positional 16-tuple (`HWGATEParams.get_model_params()`), has the same
`forward(x: (B,T,K,C)) -> (B,num_classes)`, is an `nn.Module` that takes part
in autograd, and exposes the same `state_dict()` keys/shapes,
so reference checkpoints load here and vice versa.

Inside, nothing is shared with the reference's formulation: activations stay
in natural (B,F,K,d) order for the whole network, window partition / reverse /
roll never materialise, and the attention core, LayerNorms, embedding, merging
and final pooling run as hand-written gfx950 kernels (libhwgat_hip.so).  There
is no CPU path: constructing on / moving to a CPU device works (parameters are
ordinary tensors) but `forward` needs an MI355X.
"""
import math
from typing import List, Optional

import torch
from torch import nn

from .. import functional as HF
from ..block import fused_block

_SUPPORTED_WIDTHS = (128, 256, 512, 1024)


class _Slot(nn.Module):
    """parameter container (no forward of its own)"""


def _sinusoid(max_len, d):
    pe = torch.zeros(max_len, d)
    pos = torch.arange(0, max_len).unsqueeze(1)
    div = torch.exp(torch.arange(0, d, 2) * -(math.log(10000.0) / d))
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe.view(1, max_len, 1, d)


def _last_slot_mask(frames, n_windows):
    """value of the reference's `attn_mask` buffer (HWGATE.py:169-187): all
    ones except the last temporal slot, which is block-diagonal per frame."""
    f = frames // 2
    m = torch.ones(f, n_windows, 32, 32)
    blk = torch.zeros(32, 32)
    blk[:16, :16] = 1
    blk[16:, 16:] = 1
    m[f - 1] = blk
    return m.view(f * n_windows, 32, 32)


class Model(nn.Module):
    _attn_kind = "win"          # part-window attention (hwgat_win_attn_*); HGATE overrides with "blk", WGATE with "band"

    def __init__(self, kp_dim=26, num_kps=64, temporal_dim=256, num_classes=1000, embed_dim=64,
                 temporal_patch_size=4, pe=False, depths=[2, 2, 6, 2], num_heads=[2, 4, 8, 16],
                 window_size=16, adj_mat=None, drop_rate=0., attn_drop_rate=0., ff_ratio=4.,
                 norm_layer=nn.LayerNorm, device=None) -> None:
        super().__init__()
        if temporal_patch_size != 2:
            # the reference's TemporalMerging doubles the width per stage, which is only
            # consistent with temporal_patch_size == 2 (HWGATE.py:61 vs :312)
            raise NotImplementedError("HWGAT HIP backend supports temporal_patch_size == 2")
        if window_size != 16:
            raise NotImplementedError("HWGAT HIP backend supports window_size == 16")
        if not 0.0 <= float(attn_drop_rate) < 1.0:
            raise ValueError("attn_drop_rate must be in [0, 1)")
        if norm_layer is not nn.LayerNorm:
            raise NotImplementedError("norm_layer must be nn.LayerNorm")
        assert num_kps % window_size == 0, "window size and number of kps are incompatible"
        n_stage = len(depths)
        assert temporal_dim % (2 ** n_stage) == 0, "temporal dimension must be divisible by 2**stages"
        assert embed_dim % 2 == 0
        self.kp_dim, self.num_kps, self.temporal_dim = kp_dim, num_kps, temporal_dim
        self.num_classes, self.embed_dim, self.pe = num_classes, embed_dim, pe
        self.depths, self.num_heads = list(depths), list(num_heads)
        self.drop_rate, self.ff_ratio = float(drop_rate), ff_ratio
        self.attn_drop_rate = float(attn_drop_rate)          # nn.Dropout on the attention probabilities (HWGATE.py:78,112)
        self.num_layers = n_stage
        self.num_features = int(embed_dim * 2 ** (n_stage - 1))
        self.n_windows = num_kps // 16

        self.B = nn.Parameter(torch.normal(0.0, 1.0, (embed_dim // 2, kp_dim)) * 10, requires_grad=False)
        if pe:
            self.pos_encoder = _Slot()
            self.pos_encoder.register_buffer("pe", _sinusoid(temporal_dim, embed_dim))

        self.layers = nn.ModuleList()
        for i in range(n_stage):
            d = embed_dim * 2 ** i
            if d not in _SUPPORTED_WIDTHS or d % num_heads[i] or (d // num_heads[i]) not in (32, 64, 128):
                raise NotImplementedError(
                    f"stage width {d} / heads {num_heads[i]} not supported by the HIP kernels: widths must be in "
                    f"{_SUPPORTED_WIDTHS} (the linears tile their output in 128- / 256-column blocks and the LayerNorm "
                    f"row maps exist for these widths; embed_dim = 64, the reference constructor's default that no "
                    f"reference config uses, would need 64-column instantiations -- INTEGRATION.md section 6) and "
                    f"head_dim in (32, 64, 128)")
            stage = _Slot()
            stage.blocks = nn.ModuleList()
            for j in range(depths[i]):
                blk = _Slot()
                blk.norm1 = nn.LayerNorm(d)
                blk.attn = _Slot()
                blk.attn.qkv = nn.Linear(d, 3 * d)
                blk.attn.proj = nn.Linear(d, d)
                blk.norm2 = nn.LayerNorm(d)
                blk.ff = _Slot()
                blk.ff.fc1 = nn.Linear(d, int(d * ff_ratio))
                blk.ff.fc2 = nn.Linear(int(d * ff_ratio), d)
                blk.register_buffer("attn_mask", _last_slot_mask(temporal_dim // 2 ** i, self.n_windows)
                                    if j % 2 == 1 else None)
                stage.blocks.append(blk)
            self.layers.append(stage)
        self.norm = nn.LayerNorm(self.num_features)
        self.head = nn.Linear(self.num_features, num_classes) if num_classes > 0 else nn.Identity()

        for m in self.modules():                       # reference HWGATE.py:333-340
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=.02)
                nn.init.zeros_(m.bias)

        if adj_mat is None:
            adj_mat = torch.ones(self.n_windows, 32, 32)
        self.adj_mat = adj_mat
        # compact bit rows derived from adjacency + shift structure; not part of state_dict
        self.register_buffer("_mask_bits", HF.mask_bits(adj_mat), persistent=False)
        self.part_index: Optional[torch.Tensor] = None           # set by use_part_table()
        self.activation_dtype = torch.float32
        self.threshold_override: Optional[List[float]] = None    # tests: inject train thresholds
        self._drop_calls = 0
        # the dropout seed lives on the DEVICE: {step counter, base seed of the step, initial seed, rank salt}; every
        # seeded kernel adds word 1 to its (host, per-site) seed when it runs, see include/hwgat_hip.h "dropout seeds"
        self.register_buffer("_seed_state", torch.zeros(4, dtype=torch.int32), persistent=False)
        self.device_seed_counter = False                          # True: a captured train step advances the counter itself
        self.deterministic_eval = True                            # eval(): fixed-order sums, bit-reproducible logits
        self.deterministic_train = False                          # train(): the same for the whole step (slower: no float atomics anywhere)
        if device is not None:
            self.to(device)

    # ------------------------------------------------------------ options
    def use_part_table(self, index: torch.Tensor):
        """accept raw (B,T,J,C) keypoints and gather joints on the device
        (replaces the host-side WindowCreate transform)."""
        assert index.numel() == self.num_kps
        self.register_buffer("_part_index", index.to(torch.int32).to(self.B.device), persistent=False)
        self.part_index = self._part_index
        return self

    def set_activation_dtype(self, dtype):
        assert dtype in (torch.float32, torch.bfloat16)
        self.activation_dtype = dtype
        return self

    # ------------------------------------------------------------ forward
    attn_drop_rate = 0.0      # set by every constructor (HWGATE / HGATE / WGATE)

    def _site_seeds(self, k):
        """four dropout-SITE seeds of block k (host integers that never change): proj, fc1, fc2 outputs
        (HWGATE.py:116,133,135) and the attention probabilities (HWGATE.py:112).  A kernel hashes with
        site seed + the base seed of the step, which it reads from `_seed_state[1]` on the device."""
        return [((k * 4 + s) * HF.SEED_SITE) & 0xFFFFFFFF for s in range(4)]

    def _seeds(self, k):
        """the four EFFECTIVE seeds of block k for the step whose counter is `_drop_calls` (host mirror of the device
        word: site seed + base; what hwgat_dropout_mask_f32 needs to reproduce a mask in a test)"""
        # rank_salt: data-parallel ranks share torch's seed (identical initial weights) but must not share
        # dropout masks; dist.broadcast_parameters() sets it to the rank
        base = HF.seed_base_value(torch.initial_seed(), self._drop_calls, getattr(self, "rank_salt", 0))
        return [(base + s) & 0xFFFFFFFF for s in self._site_seeds(k)]

    def _seed_base(self):
        """the 1-element device view the kernels read the step's base seed from"""
        return self._seed_state[1:2]

    def _next_step_seed(self):
        """once per train-mode forward.  Eager: the host counter goes up and the four state words are rewritten from host
        integers (kernel arguments -- no copy, no sync).  `device_seed_counter` (a captured train step, train.GraphedTrainStep):
        the device increments its own counter, so a graph replay draws fresh masks; the host counter is then only a mirror
        that the step object keeps in step."""
        if self.device_seed_counter:
            HF.seed_advance(self._seed_state)
        else:
            self._drop_calls += 1
            HF.seed_set(self._seed_state, self._drop_calls, torch.initial_seed(), getattr(self, "rank_salt", 0))

    deterministic_train = False

    def _deterministic(self):
        """bit-reproducible arithmetic for this call: eval() by default (`deterministic_eval`); train() on request
        (`deterministic_train = True`: fixed-order row statistics, pooled sum and parameter gradients -- the reference's
        single-device training repeats itself bit for bit with fixed seeds, this is the mode that does the same)"""
        return bool(self.deterministic_train if self.training else self.deterministic_eval)

    def block_list(self):
        """every PartAttentionBlock container in execution order (what functional.weight_prep derives the copies of)"""
        return [blk for st in self.layers for blk in st.blocks]

    def _block(self, h, blk, n_heads, shifted, thr, k, hand):
        """one PartAttentionBlock (HWGATE.py:189-221) = one fused autograd node (block.fused_block).  `hand` is the
        HandOver of THIS forward call: what the previous block's epilogues produced for this one (LayerNorm statistics of
        h, the carrier of the dropout-masked gradient) goes in, what this block produces for the next one comes out --
        explicit values held in a local of forward_features, nothing stored on the module.  Returns the block output
        (B,F,K,d) -- or, for the last block of a stage when the fc2 epilogue can do it, already in the TemporalMerging
        layout (B,F/2,K,2d) (forward_features checks the shape)."""
        p = self.drop_rate if self.training else 0.0
        h = h.contiguous()
        have = hand.stats if hand.of is h else None
        carrier, up = (hand.carrier, hand.up) if (hand.of is h and hand.carrier is not None) else (None, None)
        want, merge = hand.plan.get(k, (False, False))
        seeds = self._site_seeds(k)
        out, st, oc = fused_block(h, thr, blk, self._mask_bits, n_heads, shifted, p, seeds, self._attn_kind,
                                  stats=have, want_stats=want, merge_out=merge, return_stats=True,
                                  carrier=carrier, up=up, carry_out=(want or k == hand.last_block) and not merge,
                                  return_carrier=True, book=hand.book, deterministic=hand.deterministic,
                                  attn_p=self.attn_drop_rate if self.training else 0.0,
                                  prep=hand.prep.per_block[k] if hand.prep is not None else None,
                                  seed_base=hand.seed_base, deterministic_backward=hand.deterministic and self.training)
        hand.of, hand.stats, hand.carrier, hand.up = out, st, oc, ((seeds[2], p) if oc is not None else None)
        return out

    def _embed(self, x):
        if x.dim() != 4 or x.shape[1] != self.temporal_dim or x.shape[3] != self.kp_dim:
            raise ValueError(f"expected (B,{self.temporal_dim},K,{self.kp_dim}) keypoints, got {tuple(x.shape)}")
        idx = None
        if x.shape[2] != self.num_kps:
            if self.part_index is None:
                raise ValueError(f"got {x.shape[2]} joints, model has {self.num_kps} slots and no part table")
            idx = self.part_index
        x = x.contiguous().float()
        pe = self.pos_encoder.pe.view(self.temporal_dim, self.embed_dim) if self.pe else None
        if self.training:
            self._next_step_seed()
        p_pe = self.drop_rate if (self.training and self.pe) else 0.0     # Dropout lives in PositionalEncoding
        return HF.embed(x, idx, self.B, pe, self.num_kps, self.activation_dtype, p_pe, self._site_seeds(63)[0],
                        seed_base=self._seed_base() if self.training else None)

    def forward_features(self, x):
        h = self._embed(x)
        n_blocks = sum(len(st.blocks) for st in self.layers)
        hand = HF.HandOver(last_block=n_blocks - 1, deterministic=self._deterministic())
        # every derived copy of the block weights this call needs (LayerNorm folds, bf16 copies, transposes for the backward)
        hand.prep = HF.weight_prep(self, self.block_list(), self.activation_dtype, torch.is_grad_enabled())
        hand.seed_base = self._seed_base() if self.training else None
        kk = 0
        for i, stage in enumerate(self.layers):          # every block but the last feeds a LayerNorm; stage ends merge
            for j in range(len(stage.blocks)):
                hand.plan[kk] = (kk < n_blocks - 1, j == len(stage.blocks) - 1 and i < self.num_layers - 1)
                kk += 1
        k = 0
        for i, stage in enumerate(self.layers):
            for j, blk in enumerate(stage.blocks):
                thr = None
                if self.training and self._attn_kind == "win":     # HGATE has no threshold drop
                    if self.threshold_override is not None:
                        thr = torch.full((1,), float(self.threshold_override[k]), device=x.device)
                    else:
                        thr = torch.rand(1, device=x.device)      # device RNG, no host sync
                h = self._block(h, blk, self.num_heads[i], j % 2 == 1, thr, k, hand)
                k += 1
            if i < self.num_layers - 1 and h.shape[-1] == self.embed_dim * 2 ** i:
                h = HF.temporal_merge(h)                  # the fc2 epilogue could not store merged (ragged M)
        if hand.of is h and hand.carrier is not None:
            return HF.ln_mean_pool(h, self.norm.weight, self.norm.bias, carrier=hand.carrier, up=hand.up, book=hand.book,
                                   deterministic=hand.deterministic, seed_base=hand.seed_base)
        return HF.ln_mean_pool(h, self.norm.weight, self.norm.bias, deterministic=hand.deterministic)

    def forward(self, x):
        feat = self.forward_features(x)
        return self.head(feat)
