"""TEST-ONLY second formulation of a PartAttentionBlock: torch library GEMMs (rocBLAS / hipBLASLt) + torch GELU /
dropout around the standalone HIP LayerNorm and attention kernels.  The product (`sl-hwgat_amd`) has exactly one path
-- the fused hand-written linears -- and no fallback; this file exists so that the whole-model fixtures can be checked
through a second, independent arrangement of the arithmetic (reference hwgat/models/HWGATE.py:189-221)."""
import types

import torch.nn.functional as tF


def use_library_linears(model):
    """replace `model._block` by the library-GEMM formulation (returns the model)"""
    hw = __import__("importlib").import_module("sl-hwgat_amd")
    HF = hw.functional

    def linear(x, lin):
        if x.dtype == lin.weight.dtype:
            return tF.linear(x, lin.weight, lin.bias)
        return tF.linear(x, lin.weight.to(x.dtype), lin.bias.to(x.dtype))

    def drop(self, x):
        return tF.dropout(x, self.drop_rate, True) if (self.training and self.drop_rate > 0.0) else x

    def _block(self, h, blk, n_heads, shifted, thr, k, hand):
        xn = HF.layer_norm(h, blk.norm1.weight, blk.norm1.bias)
        qkv = linear(xn, blk.attn.qkv)
        if self._attn_kind == "win":
            o = HF.window_attention(qkv, self._mask_bits, thr, n_heads, shifted)
        elif self._attn_kind == "blk":
            o = HF.block_attention(qkv, self._mask_bits, n_heads, shifted)
        else:
            o = HF.band_attention(qkv, self._mask_bits, n_heads)
        y = h + drop(self, linear(o, blk.attn.proj))
        z = HF.layer_norm(y, blk.norm2.weight, blk.norm2.bias)
        u = drop(self, tF.gelu(linear(z, blk.ff.fc1)))
        return y + drop(self, linear(u, blk.ff.fc2))

    model._block = types.MethodType(_block, model)
    return model
