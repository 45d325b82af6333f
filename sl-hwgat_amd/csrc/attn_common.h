// Device helpers shared by the window (win_attn.hip) and block (blk_attn.hip) attention kernels:
// element <-> float conversion, the row-per-lane LDS tile products and the softmax exponential.
#pragma once
#include "common.h"

namespace {

template <int HD> __device__ __forceinline__ constexpr float qk_scale() {
    // float(head_dim ** -0.5), HWGATE.py:75,89
    return HD == 16 ? 0.25f : HD == 32 ? 0.17677669529663687f : HD == 64 ? 0.125f : 0.08838834764831845f;
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// 16-byte raw chunk -> floats written to LDS (optionally scaled)
template <typename T> struct chunk;
template <> struct chunk<float> {
    __device__ static __forceinline__ void to_lds(float* dst, u32x4 raw, float s) {
        f32x4 v = {__uint_as_float(raw.x) * s, __uint_as_float(raw.y) * s,
                   __uint_as_float(raw.z) * s, __uint_as_float(raw.w) * s};
        *reinterpret_cast<f32x4*>(dst) = v;
    }
};
template <> struct chunk<bf16_t> {
    __device__ static __forceinline__ void to_lds(float* dst, u32x4 raw, float s) {
        f32x4 a = {__uint_as_float(raw.x << 16) * s, __uint_as_float(raw.x & 0xffff0000u) * s,
                   __uint_as_float(raw.y << 16) * s, __uint_as_float(raw.y & 0xffff0000u) * s};
        f32x4 b = {__uint_as_float(raw.z << 16) * s, __uint_as_float(raw.z & 0xffff0000u) * s,
                   __uint_as_float(raw.w << 16) * s, __uint_as_float(raw.w & 0xffff0000u) * s};
        reinterpret_cast<f32x4*>(dst)[0] = a;
        reinterpret_cast<f32x4*>(dst)[1] = b;
    }
};

// NT consecutive elements <-> floats (NT = 1, 2 or 4)
template <typename T, int NT> __device__ __forceinline__ void load_nt(const T* p, float (&v)[NT]) {
    if constexpr (sizeof(T) == 4) {
        if constexpr (NT == 4) { f32x4 t = *reinterpret_cast<const f32x4*>(p); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
        else if constexpr (NT == 2) { f32x2 t = *reinterpret_cast<const f32x2*>(p); v[0] = t.x; v[1] = t.y; }
        else v[0] = *p;
    } else {
        if constexpr (NT == 4) { bf16x4 t = *reinterpret_cast<const bf16x4*>(p); v[0] = (float)t.x; v[1] = (float)t.y; v[2] = (float)t.z; v[3] = (float)t.w; }
        else if constexpr (NT == 2) { bf16x2 t = *reinterpret_cast<const bf16x2*>(p); v[0] = (float)t.x; v[1] = (float)t.y; }
        else v[0] = (float)*p;
    }
}
template <typename T, int NT> __device__ __forceinline__ void store_nt(T* p, const float (&v)[NT]) {
    if constexpr (sizeof(T) == 4) {
        if constexpr (NT == 4) { f32x4 t = {v[0], v[1], v[2], v[3]}; *reinterpret_cast<f32x4*>(p) = t; }
        else if constexpr (NT == 2) { f32x2 t = {v[0], v[1]}; *reinterpret_cast<f32x2*>(p) = t; }
        else *p = v[0];
    } else {
        if constexpr (NT == 4) { bf16x4 t = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]}; *reinterpret_cast<bf16x4*>(p) = t; }
        else if constexpr (NT == 2) { bf16x2 t = {(bf16_t)v[0], (bf16_t)v[1]}; *reinterpret_cast<bf16x2*>(p) = t; }
        else *p = (bf16_t)v[0];
    }
}

// swap with the partner lane that holds the other 16 keys of the same query
__device__ __forceinline__ float partner(float v) { return __shfl_xor(v, 32, 64); }

// softmax exponential: hardware v_exp_f32 (2^x) on x*log2(e).  Arguments are <= 0; the relative error is
// ~|x| * 4e-8 (2e-6 at x = -30, below which the term is < 1e-13 of the row sum anyway), two instructions
// instead of libm's ~20 -- the softmax VALU work otherwise competes with the MFMAs for the wave's time.
__device__ __forceinline__ float sm_exp(float x) { return __expf(x); }

// D(32x32) += X Y^T for two row-per-lane LDS tiles X, Y of [32][LDW] floats.
// result lane (j = lane&31, hh), reg r  ->  D[crow(r,hh)][j]  (rows index X).
template <int HD, int LDW>
__device__ __forceinline__ f32x16 tile_xyT(const float* X, const float* Y, int lq, int hh) {
    f32x16 acc = {0.f};
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const float* xr = X + lq * LDW + 4 * hh;
    const float* yr = Y + lq * LDW + 4 * hh;
#pragma unroll
    for (int m = 0; m < HD / 8; ++m) {
        const f32x4 xf = *reinterpret_cast<const f32x4*>(xr + 8 * m);
        const f32x4 yf = *reinterpret_cast<const f32x4*>(yr + 8 * m);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xf.x, yf.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xf.y, yf.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xf.z, yf.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xf.w, yf.w, acc, 0, 0, 0);
    }
    return acc;
}

__device__ __forceinline__ void lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// acc(32 x HD) = A(32x32) . Y where the A operand value for k-step r is a[r]
// (row = lane&31, k = crow(r,hh)) and Y is an LDS tile [32][LDW] read in the
// B layout (lane (c,hh) reads Y[crow(r,hh)][c*NT .. +NT-1]).
template <int HD, int LDW, bool ZERO = true>
__device__ __forceinline__ void tile_ay(const float (&a)[16], const float* Y, int lq, int hh,
                                        f32x16 (&acc)[HD / 32]) {
    constexpr int NT = HD / 32;
    if constexpr (ZERO) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[nt][i] = 0.f;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float y[NT];
        const float* yp = Y + crow(r, hh) * LDW + lq * NT;
        if constexpr (NT == 4) { f32x4 t = *reinterpret_cast<const f32x4*>(yp); y[0] = t.x; y[1] = t.y; y[2] = t.z; y[3] = t.w; }
        else if constexpr (NT == 2) { f32x2 t = *reinterpret_cast<const f32x2*>(yp); y[0] = t.x; y[1] = t.y; }
        else y[0] = *yp;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r], y[nt], acc[nt], 0, 0, 0);
    }
}

}  // namespace
