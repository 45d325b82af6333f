// HGATE block attention: geometry and unit decoding shared by the 32x32-tile kernels (blk_attn.hip) and the four-wave
// kernels on 16x16 tiles (blk_attn_bf16.hip, blk_attn_f32.hip).  Reference: hwgat/models/HGATE.py:30-47,184-207 (block partition / roll), :84-108.
#pragma once
#include "common.h"

namespace blk {

struct BlkGeom {
    int F, KJ, nH, f, d, shift;
};

struct BUnit {
    int64_t base[2];        // token index of joint 0 of frame A / frame B
    int head, mrow;         // head index, first row of this unit's mask variant
};
__device__ __forceinline__ BUnit decode_bunit(const BlkGeom& g, int u) {
    BUnit r;
    const int n = u / g.nH;
    r.head = u - n * g.nH;
    const int fi = n % g.f;
    const int b = n / g.f;
    int fa = 2 * fi + g.shift, fb = fa + 1;          // torch.roll(x, -shift) (HGATE.py:186): shifted[t] = x[(t+shift) % F]
    if (fa >= g.F) fa -= g.F;
    if (fb >= g.F) fb -= g.F;
    r.base[0] = ((int64_t)b * g.F + fa) * g.KJ;
    r.base[1] = ((int64_t)b * g.F + fb) * g.KJ;
    r.mrow = (g.shift && fi == g.f - 1) ? 64 : 0;    // the last shifted block straddles the clip ends (HGATE.py:158-172)
    return r;
}

}  // namespace blk

// bf16 storage, head_dim 64: 16x16x32 bf16 MFMA tiles, one workgroup of 4 waves per unit (blk_attn_bf16.hip)
// (drop_seed, drop_p, seed_base): attention dropout, p = 0 for none
int hwgat_launch_blk_fwd_b16(const void* qkv, void* o, const uint32_t* maskbits, int B, int F, int KJ, int nH, int shifted,
                             uint32_t drop_seed, float drop_p, const uint32_t* seed_base, hipStream_t st);
int hwgat_launch_blk_bwd_b16(const void* qkv, const void* dO, void* dqkv, const uint32_t* maskbits, int B, int F, int KJ,
                             int nH, int shifted, uint32_t drop_seed, float drop_p, const uint32_t* seed_base, hipStream_t st);
// fp32 storage and arithmetic, head_dim 64: 16x16x4 fp32 MFMA tiles, one workgroup of 4 waves per unit (blk_attn_f32.hip)
int hwgat_launch_blk_fwd_f32(const void* qkv, void* o, const uint32_t* maskbits, int B, int F, int KJ, int nH, int shifted,
                             uint32_t drop_seed, float drop_p, const uint32_t* seed_base, hipStream_t st);
int hwgat_launch_blk_bwd_f32(const void* qkv, const void* dO, void* dqkv, const uint32_t* maskbits, int B, int F, int KJ,
                             int nH, int shifted, uint32_t drop_seed, float drop_p, const uint32_t* seed_base, hipStream_t st);
