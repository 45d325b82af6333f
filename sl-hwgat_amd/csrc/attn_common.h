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
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

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

// streaming forms (every byte of q, k, v, dO is read once and every byte of o, dq, dk, dv written once by whole 256-byte
// rows): nontemporal hints keep them out of the way of each other in L2
template <typename T, int NT> __device__ __forceinline__ void load_nt_s(const T* p, float (&v)[NT]) {
    if constexpr (sizeof(T) == 4 && NT == 4) { f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p)); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
    else if constexpr (sizeof(T) == 4 && NT == 2) { f32x2 t = __builtin_nontemporal_load(reinterpret_cast<const f32x2*>(p)); v[0] = t.x; v[1] = t.y; }
    else if constexpr (sizeof(T) == 2 && NT == 4) {
        const u32x2 t = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(p));
        v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u); v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
    } else if constexpr (sizeof(T) == 2 && NT == 2) {
        const uint32_t t = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(p));
        v[0] = __uint_as_float(t << 16); v[1] = __uint_as_float(t & 0xffff0000u);
    } else load_nt<T, NT>(p, v);
}
template <typename T, int NT> __device__ __forceinline__ void store_nt_s(T* p, const float (&v)[NT]) {
    if constexpr (sizeof(T) == 4 && NT == 4) { f32x4 t = {v[0], v[1], v[2], v[3]}; __builtin_nontemporal_store(t, reinterpret_cast<f32x4*>(p)); }
    else if constexpr (sizeof(T) == 4 && NT == 2) { f32x2 t = {v[0], v[1]}; __builtin_nontemporal_store(t, reinterpret_cast<f32x2*>(p)); }
    else if constexpr (sizeof(T) == 2 && NT == 4) {
        const bf16x4 t = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
        __builtin_nontemporal_store(__builtin_bit_cast(u32x2, t), reinterpret_cast<u32x2*>(p));
    } else if constexpr (sizeof(T) == 2 && NT == 2) {
        const bf16x2 t = {(bf16_t)v[0], (bf16_t)v[1]};
        __builtin_nontemporal_store(__builtin_bit_cast(uint32_t, t), reinterpret_cast<uint32_t*>(p));
    } else store_nt<T, NT>(p, v);
}
template <typename T> __device__ __forceinline__ u32x4 load16_s(const T* p) {
    return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
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

// ---------------------------------------------------------------- bf16 storage: the same products on
// v_mfma_f32_32x32x16_bf16.  With fp32 MFMAs a (window, head) unit of head_dim 64 costs 64 + 32 NT matrix
// instructions of 64 cycles each whatever the storage type, so halving the bytes (bf16 activations) halved the
// fraction of the HBM roof these kernels reach (0.49 / 0.45 forward / backward on config 3).  bf16 tiles keep the raw
// 16-byte chunks in LDS (no expansion to fp32), and every 32x32xHD product is HD / 16 instructions of 32 cycles.
// Softmax, masks, threshold and all accumulation stay fp32; P and dS are rounded to bf16 only as MFMA operands
// (what torch.autocast does with the softmax output in front of its bf16 matmul).
template <typename T> struct tile_of;                  // element type of the LDS tiles, row padding, is Q stored pre-scaled
template <> struct tile_of<float> { using E = float; static constexpr int PAD = 4; static constexpr bool QSCALED = true; };
template <> struct tile_of<bf16_t> { using E = bf16_t; static constexpr int PAD = 8; static constexpr bool QSCALED = false; };

__device__ __forceinline__ void raw_to_lds(float* dst, u32x4 raw, float s, float) { chunk<float>::to_lds(dst, raw, s); }
__device__ __forceinline__ void raw_to_lds(bf16_t* dst, u32x4 raw, float, bf16_t) { *reinterpret_cast<u32x4*>(dst) = raw; }

// four consecutive scratch elements (8-byte aligned fp32 pairs / 4-byte aligned bf16 pairs)
__device__ __forceinline__ void put4(float* dst, const float* v) {
    f32x2 a = {v[0], v[1]}, b = {v[2], v[3]};
    reinterpret_cast<f32x2*>(dst)[0] = a; reinterpret_cast<f32x2*>(dst)[1] = b;
}
__device__ __forceinline__ void put4(bf16_t* dst, const float* v) {
    bf16x2 a = {(bf16_t)v[0], (bf16_t)v[1]}, b = {(bf16_t)v[2], (bf16_t)v[3]};
    reinterpret_cast<bf16x2*>(dst)[0] = a; reinterpret_cast<bf16x2*>(dst)[1] = b;
}

__device__ __forceinline__ bf16x8 pack_bf16x8(const float* v) {
    bf16x8 f = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3], (bf16_t)v[4], (bf16_t)v[5], (bf16_t)v[6], (bf16_t)v[7]};
    return f;
}

// D(32x32) = X Y^T for two row-per-lane bf16 tiles [32][LDB]; same result layout as the fp32 form
template <int HD, int LDB>
__device__ __forceinline__ f32x16 tile_xyT(const bf16_t* X, const bf16_t* Y, int lq, int hh) {
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const bf16_t* xr = X + lq * LDB + 8 * hh;
    const bf16_t* yr = Y + lq * LDB + 8 * hh;
#pragma unroll
    for (int m = 0; m < HD / 16; ++m) {
        const bf16x8 xf = *reinterpret_cast<const bf16x8*>(xr + 16 * m);
        const bf16x8 yf = *reinterpret_cast<const bf16x8*>(yr + 16 * m);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf, yf, acc, 0, 0, 0);
    }
    return acc;
}

// the NT consecutive elements a lane holds of one row of the B-side matrix (V, K, Q or dO): raw bf16 pairs
template <int NT> struct brow {
    static constexpr int W = NT >= 2 ? NT / 2 : 1;
    uint32_t w[W];
    __device__ __forceinline__ void load(const bf16_t* p) {
        if constexpr (NT == 1) w[0] = *reinterpret_cast<const uint16_t*>(p);
        else if constexpr (NT == 2) w[0] = *reinterpret_cast<const uint32_t*>(p);
        else { const u32x2 t = *reinterpret_cast<const u32x2*>(p); w[0] = t.x; w[1] = t.y; }
    }
    __device__ __forceinline__ void load_s(const bf16_t* p) {       // streaming form (see load_nt_s)
        if constexpr (NT == 1) w[0] = *reinterpret_cast<const uint16_t*>(p);
        else if constexpr (NT == 2) w[0] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(p));
        else { const u32x2 t = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(p)); w[0] = t.x; w[1] = t.y; }
    }
};
// B operand of one k16 step for column tile nt: element j of the lane = rows[j], column nt of the lane's NT
template <int NT>
__device__ __forceinline__ bf16x8 bfrag(const brow<NT>* rows, int nt) {
    const int wd = nt >> 1;
    const uint32_t sel = (nt & 1) ? 0x07060302u : 0x05040100u;     // high or low halves of (rows[2d+1], rows[2d])
    u32x4 f;
    f.x = __builtin_amdgcn_perm(rows[1].w[wd], rows[0].w[wd], sel);
    f.y = __builtin_amdgcn_perm(rows[3].w[wd], rows[2].w[wd], sel);
    f.z = __builtin_amdgcn_perm(rows[5].w[wd], rows[4].w[wd], sel);
    f.w = __builtin_amdgcn_perm(rows[7].w[wd], rows[6].w[wd], sel);
    return __builtin_bit_cast(bf16x8, f);
}
// acc(32 x 32 NT) (+)= A . B with the A operand in the accumulator layout (lane = row, a[r] <-> k = crow(r,hh)) and
// the 16 B rows of the lane (rows[r] = row crow(r,hh) of B, the lane's NT columns): two k16 steps of 8 registers
template <int NT>
__device__ __forceinline__ void mfma_ab(const float (&a)[16], const brow<NT> (&rows)[16], f32x16 (&acc)[NT]) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const bf16x8 af = pack_bf16x8(a + 8 * s);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfrag<NT>(rows + 8 * s, nt), acc[nt], 0, 0, 0);
    }
}

// tile_ay with a bf16 B-side tile [32][LDB]
template <int HD, int LDB, bool ZERO = true>
__device__ __forceinline__ void tile_ay(const float (&a)[16], const bf16_t* Y, int lq, int hh, f32x16 (&acc)[HD / 32]) {
    constexpr int NT = HD / 32;
    if constexpr (ZERO) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[nt][i] = 0.f;
    }
    brow<NT> rows[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) rows[r].load(Y + crow(r, hh) * LDB + lq * NT);
    mfma_ab<NT>(a, rows, acc);
}

}  // namespace
