// WGATE band attention: geometry and unit decoding shared by the fp32 kernels (band_attn.hip) and the bf16-MFMA
// kernels (band_attn_bf16.hip).  Reference: hwgat/models/WGATE.py:32-65 (window partition), :87-108 (attention core).
#pragma once
#include "common.h"
#include "fused_ops.h"            // the dropout hash (attention dropout, WGATE.py:81,103)

namespace band {

struct BandGeom {
    int F, K, nW, nH, d, seg, n_seg;      // seg = query frames per unit, n_seg = segments per clip
};

template <int HD> __device__ __forceinline__ constexpr float band_scale() {
    return HD == 16 ? 0.25f : 0.17677669529663687f;            // float(head_dim ** -0.5), WGATE.py:79,92
}

struct BandUnit {
    int64_t tok0;          // token index of (clip, frame 0, first joint of the window)
    int head, w, f0, f1;   // query frames [f0, f1)
    int bw;                // clip * nW + window: row block of the reference's (B nW, nH, T 16, T 16) attention tensor
};
__device__ __forceinline__ BandUnit decode_band(const BandGeom& g, int u) {
    BandUnit r;
    const int sgi = u % g.n_seg;
    int t = u / g.n_seg;
    r.head = t % g.nH;
    t /= g.nH;
    r.bw = t;
    r.w = t % g.nW;
    const int b = t / g.nW;
    r.tok0 = (int64_t)b * g.F * g.K + r.w * 16;
    r.f0 = sgi * g.seg;
    r.f1 = min(g.F, r.f0 + g.seg);
    return r;
}

// attention dropout: keep[t] = the four factors 1/(1-p) or 0 of P[query = (frame f, joint lr)][key = (frame f-1+t, joint 4g + r)]
// for head `head` of clip-window `bw`: element ((bw nH + head) T16 + f 16 + lr) T16 + (f-1+t) 16 + 4g + r of the reference's
// dense (B nW, nH, T16, T16) attention tensor, T16 = F 16 (WGATE.py:89-103).  Tiles outside the clip are masked anyway.
__device__ __forceinline__ void band_keep(f32x4 (&k)[3], const AttnDrop& ad, int bw, int nH, int head, int F, int f, int lr, int gq) {
    const uint32_t thresh = drop_thresh(ad.p);
    const float scale = 1.0f / (1.0f - ad.p);
    const uint64_t T16 = (uint64_t)F * 16;
    const uint64_t row = (((uint64_t)bw * nH + head) * T16 + (uint64_t)f * 16 + lr) * T16;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int kf = min(max(f - 1 + t, 0), F - 1);
        k[t] = drop_keep4(ad.seed, row + (uint64_t)kf * 16 + 4 * gq, thresh, scale);
    }
}

inline bool band_ok(int B, int F, int nW, int nH, int hd) {
    return B > 0 && F > 0 && nW > 0 && nH > 0 && (hd == 16 || hd == 32);
}


}  // namespace band

// bf16 storage: 16x16x16 bf16 MFMA tiles, LDS-transposed column operands (band_attn_bf16.hip)
// (drop_seed, drop_p, seed_base): attention dropout, p = 0 for none
int hwgat_launch_band_fwd_b16(const void* qkv, void* o, const uint64_t* maskrows, int B, int F, int nW, int nH, int hd,
                              uint32_t drop_seed, float drop_p, const uint32_t* seed_base, hipStream_t st);
int hwgat_launch_band_bwd_b16(const void* qkv, const void* dO, void* dqkv, const uint64_t* maskrows, int B, int F, int nW,
                              int nH, int hd, uint32_t drop_seed, float drop_p, const uint32_t* seed_base, hipStream_t st);
// fp32 storage and arithmetic, head_dim 16: 16x16x4 fp32 MFMA tiles, workgroup-staged whole-row tiles (band_attn_f32.hip)
int hwgat_launch_band_fwd_f32(const void* qkv, void* o, const uint64_t* maskrows, int B, int F, int nW, int nH,
                              uint32_t drop_seed, float drop_p, const uint32_t* seed_base, hipStream_t st);
int hwgat_launch_band_bwd_f32(const void* qkv, const void* dO, void* dqkv, const uint64_t* maskrows, int B, int F, int nW,
                              int nH, uint32_t drop_seed, float drop_p, const uint32_t* seed_base, hipStream_t st);
