"""WGATE (windowed graph attention WITHOUT hierarchy) -- MI355X-native backend.

Drop-in for the reference's `hwgat/models/WGATE.py`: class `Model` takes the same
positional 15-tuple (`WGATEParams.get_model_params()`: integer `depths` / `num_heads`, no temporal patch
size), has the same `forward(x: (B,T,K,C)) -> (B,num_classes)` and the same `state_dict()` keys / shapes
-- `layers.{i}.<...>` directly (no stages) and the (nW, T*16, T*16) additive `adj_mask` buffer of
WGATE.py:190-196 -- so checkpoints interchange.

An attention window is one 16-joint part window over ALL T frames.  The reference forms the dense
(T*16)^2 scores and adds 0 / -10000; its adjacency is block-tridiagonal over frames, so every masked
probability is exactly 0 in fp32 and `hwgat_band_attn_fwd/bwd` (csrc/band_attn.hip) only ever touch the
three neighbouring key frames.  Embedding, LayerNorms, fused linears and the final norm + pool are the
kernels HWGATE uses, unchanged.  `adj_mask` is kept only for the state_dict contract; the kernels read
the (nW,16) bit rows derived from `adj_mat`.
"""
import torch
from torch import nn

from .. import functional as HF
from . import HWGATE as _base


class Model(_base.Model):
    _attn_kind = "band"

    def __init__(self, kp_dim=26, num_kps=64, temporal_dim=256, num_classes=1000, embed_dim=64, pe=False,
                 depths=16, num_heads=8, window_size=16, ff_ratio=4., adj_mat=None, drop_rate=0.,
                 attn_drop_rate=0., norm_layer=nn.LayerNorm, device=None) -> None:
        nn.Module.__init__(self)
        if window_size != 16:
            raise NotImplementedError("WGATE HIP backend supports window_size == 16")
        if not 0.0 <= float(attn_drop_rate) < 1.0:
            raise ValueError("attn_drop_rate must be in [0, 1)")
        self.attn_drop_rate = float(attn_drop_rate)          # nn.Dropout on the attention probabilities (WGATE.py:81,103)
        if norm_layer is not nn.LayerNorm:
            raise NotImplementedError("norm_layer must be nn.LayerNorm")
        if adj_mat is None:
            raise NotImplementedError("WGATE needs its (nW, T*16, T*16) adjacency (the reference dereferences it too)")
        assert num_kps % window_size == 0, "window size and number of kps are incompatible"
        d = embed_dim
        if d not in _base._SUPPORTED_WIDTHS or d % num_heads or (d // num_heads) not in (16, 32):
            raise NotImplementedError(f"width {d} / heads {num_heads} not supported by the HIP kernels "
                                      "(band attention: head_dim 16 or 32)")
        self.kp_dim, self.num_kps, self.temporal_dim = kp_dim, num_kps, temporal_dim
        self.num_classes, self.embed_dim, self.pe = num_classes, embed_dim, pe
        self.depths, self.num_heads = int(depths), int(num_heads)
        self.drop_rate, self.ff_ratio = float(drop_rate), ff_ratio
        self.window_size = window_size
        self.num_features = embed_dim
        self.n_windows = num_kps // 16

        rows = HF.band_mask_rows(adj_mat, temporal_dim)        # validates the structure the kernel relies on
        self.adj_mask_name = "adj_mask"                        # WGATE.py:190-196: 0 -> -10000, 1 -> 0
        self.register_buffer("adj_mask", adj_mat.to(torch.float32).masked_fill(adj_mat == 0, float(-10000))
                             .masked_fill(adj_mat == 1, float(0)))
        self.B = nn.Parameter(torch.normal(0.0, 1.0, (embed_dim // 2, kp_dim)) * 10, requires_grad=False)
        if pe:
            self.pos_encoder = _base._Slot()
            self.pos_encoder.register_buffer("pe", _base._sinusoid(temporal_dim, embed_dim))
        self.layers = nn.ModuleList()
        for _ in range(self.depths):
            blk = _base._Slot()
            blk.norm1 = nn.LayerNorm(d)
            blk.attn = _base._Slot()
            blk.attn.qkv = nn.Linear(d, 3 * d)
            blk.attn.proj = nn.Linear(d, d)
            blk.norm2 = nn.LayerNorm(d)
            blk.ff = _base._Slot()
            blk.ff.fc1 = nn.Linear(d, int(d * ff_ratio))
            blk.ff.fc2 = nn.Linear(int(d * ff_ratio), d)
            self.layers.append(blk)
        self.norm = nn.LayerNorm(d)
        self.head = nn.Linear(d, num_classes) if num_classes > 0 else nn.Identity()

        for m in self.modules():                       # reference WGATE.py:229-236
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=.02)
                nn.init.zeros_(m.bias)

        self.adj_mat = adj_mat
        self.register_buffer("_mask_bits", rows, persistent=False)
        self.part_index = None
        self.activation_dtype = torch.float32
        self.threshold_override = None
        self._drop_calls = 0
        self.register_buffer("_seed_state", torch.zeros(4, dtype=torch.int32), persistent=False)   # see HWGATE.Model
        self.device_seed_counter = False
        self.deterministic_eval = True
        if device is not None:
            self.to(device)

    def block_list(self):
        return list(self.layers)

    def forward_features(self, x):
        h = self._embed(x)
        hand = HF.HandOver(last_block=self.depths - 1, deterministic=self._deterministic())
        hand.prep = HF.weight_prep(self, self.block_list(), self.activation_dtype, torch.is_grad_enabled())
        hand.seed_base = self._seed_base() if self.training else None
        for k in range(self.depths):                   # every block but the last feeds the next block's LayerNorm
            hand.plan[k] = (k < self.depths - 1, False)
        for k, blk in enumerate(self.layers):          # PartAttentionBlock.forward, WGATE.py:150-160
            h = self._block(h, blk, self.num_heads, False, None, k, hand)
        if hand.of is h and hand.carrier is not None:
            return HF.ln_mean_pool(h, self.norm.weight, self.norm.bias, carrier=hand.carrier, up=hand.up, book=hand.book,
                                   deterministic=hand.deterministic, seed_base=hand.seed_base)
        return HF.ln_mean_pool(h, self.norm.weight, self.norm.bias, deterministic=hand.deterministic)
