// WGATE band attention: geometry and unit decoding shared by the fp32 kernels (band_attn.hip) and the bf16-MFMA
// kernels (band_attn_bf16.hip).  Reference: hwgat/models/WGATE.py:32-65 (window partition), :87-108 (attention core).
#pragma once
#include "common.h"

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
};
__device__ __forceinline__ BandUnit decode_band(const BandGeom& g, int u) {
    BandUnit r;
    const int sgi = u % g.n_seg;
    int t = u / g.n_seg;
    r.head = t % g.nH;
    t /= g.nH;
    r.w = t % g.nW;
    const int b = t / g.nW;
    r.tok0 = (int64_t)b * g.F * g.K + r.w * 16;
    r.f0 = sgi * g.seg;
    r.f1 = min(g.F, r.f0 + g.seg);
    return r;
}

inline bool band_ok(int B, int F, int nW, int nH, int hd) {
    return B > 0 && F > 0 && nW > 0 && nH > 0 && (hd == 16 || hd == 32);
}


}  // namespace band

// bf16 storage: 16x16x16 bf16 MFMA tiles, LDS-transposed column operands (band_attn_bf16.hip)
int hwgat_launch_band_fwd_b16(const void* qkv, void* o, const uint64_t* maskrows, int B, int F, int nW, int nH, int hd,
                              hipStream_t st);
int hwgat_launch_band_bwd_b16(const void* qkv, const void* dO, void* dqkv, const uint64_t* maskrows, int B, int F, int nW,
                              int nH, int hd, hipStream_t st);
