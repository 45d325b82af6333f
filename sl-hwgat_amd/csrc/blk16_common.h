// HGATE block attention on 16x16 MFMA tiles, head_dim 64, four waves per unit: what the bf16 kernels (blk_attn_bf16.hip) and the
// fp32 kernels (blk_attn_f32.hip) share -- the masks / fill / softmax numerators of one query's 64 key slots in the
// "lane = query, 4 x 4 registers = keys 16 kt + 4 g + r" layout both get from S^T = K Q^T, and the attention-dropout factors.
#pragma once
#include "blk_common.h"
#include "attn16_common.h"
#include "fused_ops.h"            // the dropout hash (attention dropout, HGATE.py:78,106)

namespace {
using namespace blk;

constexpr float SCALE = 0.125f;                          // float(64 ** -0.5), HGATE.py:79,91

// masks (HGATE.py:96-104) + the "== 0 -> -10000" fill + softmax numerators over the 64 key slots of one query; pad key slots
// are no keys at all.  s[kt][r] = raw score of key slot 16 kt + 4g + r on entry, exp(scaled, masked score - row max) on
// exit; returns the row sum, `nz` = bit 4 kt + r set where the logit was kept (the gradient flows).
__device__ __forceinline__ float masked_exp64(f32x4v (&s)[4], uint32_t mb0, uint32_t mb1, int gq, int KJ, uint32_t& nz) {
    nz = 0;
    float m = -3.0e38f;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = 16 * (kt & 1) + 4 * gq + r;
            const bool vis = ((kt >> 1 ? mb1 : mb0) >> j) & 1u;
            float v = vis ? s[kt][r] * SCALE : 0.f;
            if (v == 0.f) v = -10000.f; else nz |= 1u << (4 * kt + r);          // HGATE.py:104
            if (j >= KJ) { v = -3.0e38f; nz &= ~(1u << (4 * kt + r)); }
            s[kt][r] = v;
            m = __builtin_fmaxf(m, v);
        }
    m = xg_max(m);
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            s[kt][r] = __builtin_amdgcn_exp2f((s[kt][r] - m) * LOG2E);
            sum += s[kt][r];
        }
    return xg_sum(sum);
}

// attention dropout: keep[kt][r] = 1/(1-p) or 0 for P[query slot][key slot 16 kt + 4g + r] of unit u = element
// ((u * N2 + q) * N2 + key) of the reference's (B f, nH, N2, N2) attention tensor, N2 = 2 KJ, token = frame * KJ + joint
__device__ __forceinline__ void blk_keep16(f32x4v (&k)[4], const AttnDrop& ad, int u, int q_slot, int gq, int KJ) {
    const uint32_t thresh = drop_thresh(ad.p);
    const float scale = 1.0f / (1.0f - ad.p);
    const int qj = q_slot & 31;
    const uint64_t row = ((uint64_t)u * (2 * KJ) + (q_slot >> 5) * KJ + (qj < KJ ? qj : KJ - 1)) * (2 * KJ);
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = 16 * (kt & 1) + 4 * gq + r;
            k[kt][r] = j < KJ ? drop_keep(ad.seed, row + (kt >> 1) * KJ + j, thresh, scale) : 0.f;
        }
}

}  // namespace
