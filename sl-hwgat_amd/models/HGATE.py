"""HGATE (hierarchical graph attention WITHOUT body-part windows) -- MI355X-native backend.

Drop-in for the reference's `hwgat/models/HGATE.py`: class `Model` takes the same
positional 15-tuple (`HGATEParams.get_model_params()`, no `window_size`), has the same
`forward(x: (B,T,K,C)) -> (B,num_classes)` and the same `state_dict()` keys / shapes -- including the odd
blocks' `attn_mask` buffers of shape (F/2, 2K, 2K) (HGATE.py:154-172) -- so checkpoints interchange.

Relative to HWGATE only the attention core differs: a block is 2 frames x ALL K joints (K <= 32; 29 in
HGATEParams), one (2K, 2K) adjacency for every block, and no train-mode threshold (HGATE.py:84-108).
It runs on `hwgat_blk_attn_fwd/bwd` (csrc/blk_attn.hip); embedding, LayerNorms, the fused linears,
TemporalMerging and the final norm + pool are the kernels HWGATE uses, unchanged.
"""
import torch
from torch import nn

from .. import functional as HF
from . import HWGATE as _base


def _last_block_mask(frames, n_joints):
    """value of the reference's `attn_mask` buffer (HGATE.py:154-172): ones, except the last block,
    where only same-frame pairs are allowed."""
    f, K = frames // 2, n_joints
    m = torch.ones(f, 2 * K, 2 * K)
    blk = torch.zeros(2 * K, 2 * K)
    blk[:K, :K] = 1
    blk[K:, K:] = 1
    m[f - 1] = blk
    return m


class Model(_base.Model):
    _attn_kind = "blk"

    def __init__(self, kp_dim=26, num_kps=64, temporal_dim=256, num_classes=1000, embed_dim=64,
                 temporal_patch_size=4, pe=False, depths=[2, 2, 6, 2], num_heads=[2, 4, 8, 16],
                 adj_mat=None, drop_rate=0., attn_drop_rate=0., ff_ratio=4., norm_layer=nn.LayerNorm,
                 device=None) -> None:
        nn.Module.__init__(self)
        if temporal_patch_size != 2:
            raise NotImplementedError("HGATE HIP backend supports temporal_patch_size == 2")
        if not 0.0 <= float(attn_drop_rate) < 1.0:
            raise ValueError("attn_drop_rate must be in [0, 1)")
        self.attn_drop_rate = float(attn_drop_rate)          # nn.Dropout on the attention probabilities (HGATE.py:78,106)
        if norm_layer is not nn.LayerNorm:
            raise NotImplementedError("norm_layer must be nn.LayerNorm")
        if not 1 <= num_kps <= 32:
            raise NotImplementedError("HGATE HIP backend supports at most 32 joints per frame (2 x 32-row MFMA tiles)")
        n_stage = len(depths)
        assert temporal_dim % (2 ** n_stage) == 0, "temporal dimension must be divisible by 2**stages"
        assert embed_dim % 2 == 0
        self.kp_dim, self.num_kps, self.temporal_dim = kp_dim, num_kps, temporal_dim
        self.num_classes, self.embed_dim, self.pe = num_classes, embed_dim, pe
        self.depths, self.num_heads = list(depths), list(num_heads)
        self.drop_rate, self.ff_ratio = float(drop_rate), ff_ratio
        self.num_layers = n_stage
        self.num_features = int(embed_dim * 2 ** (n_stage - 1))

        self.B = nn.Parameter(torch.normal(0.0, 1.0, (embed_dim // 2, kp_dim)) * 10, requires_grad=False)
        if pe:
            self.pos_encoder = _base._Slot()
            self.pos_encoder.register_buffer("pe", _base._sinusoid(temporal_dim, embed_dim))

        self.layers = nn.ModuleList()
        for i in range(n_stage):
            d = embed_dim * 2 ** i
            if d not in _base._SUPPORTED_WIDTHS or d % num_heads[i] or (d // num_heads[i]) not in (32, 64):
                raise NotImplementedError(
                    f"stage width {d} / heads {num_heads[i]} not supported by the HIP kernels: widths must be in "
                    f"{_base._SUPPORTED_WIDTHS} (the linears tile their output in 128- / 256-column blocks and the LayerNorm "
                    f"row maps exist for these widths; embed_dim = 64, the reference constructor's default that no "
                    f"reference config uses, would need 64-column instantiations -- INTEGRATION.md section 6) and "
                    f"head_dim in (32, 64)")
            stage = _base._Slot()
            stage.blocks = nn.ModuleList()
            for j in range(depths[i]):
                blk = _base._Slot()
                blk.norm1 = nn.LayerNorm(d)          # registration order of HGATE.py:150-174 (state_dict key order)
                blk.norm2 = nn.LayerNorm(d)
                blk.ff = _base._Slot()
                blk.ff.fc1 = nn.Linear(d, int(d * ff_ratio))
                blk.ff.fc2 = nn.Linear(int(d * ff_ratio), d)
                blk.register_buffer("attn_mask", _last_block_mask(temporal_dim // 2 ** i, num_kps)
                                    if j % 2 == 1 else None)
                blk.attn = _base._Slot()
                blk.attn.qkv = nn.Linear(d, 3 * d)
                blk.attn.proj = nn.Linear(d, d)
                stage.blocks.append(blk)
            self.layers.append(stage)
        self.norm = nn.LayerNorm(self.num_features)
        self.head = nn.Linear(self.num_features, num_classes) if num_classes > 0 else nn.Identity()

        for m in self.modules():                       # reference HGATE.py:317-324
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=.02)
                nn.init.zeros_(m.bias)

        if adj_mat is None:
            adj_mat = torch.ones(2 * num_kps, 2 * num_kps)
        self.adj_mat = adj_mat
        self.register_buffer("_mask_bits", HF.blk_mask_bits(adj_mat, num_kps), persistent=False)
        self.part_index = None                         # HGATE consumes the raw joints: no part table
        self.activation_dtype = torch.float32
        self.threshold_override = None
        self._drop_calls = 0
        self.register_buffer("_seed_state", torch.zeros(4, dtype=torch.int32), persistent=False)   # see HWGATE.Model
        self.device_seed_counter = False
        self.deterministic_eval = True
        if device is not None:
            self.to(device)

    def use_part_table(self, index):
        raise NotImplementedError("HGATE takes the raw (B,T,K,C) joints; there are no part windows to gather")
