// 16x16x32 bf16 MFMA tile helpers shared by the head_dim-64 bf16 attention kernels that stage their operands as images of
// 128-byte rows in LDS (blk_attn_bf16.hip, win_attn_bf16.hip): operand packing, cross-lane reductions, the row swizzle,
// row / column operand reads.
#pragma once
#include "common.h"

namespace {

typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
typedef __attribute__((address_space(3))) u32x2v lds_u32x2;
typedef __attribute__((address_space(3))) u32x4v lds_u32x4;
typedef __attribute__((address_space(3))) void* lds_void;

constexpr int HD = 64, RB = 128;                         // head_dim; bytes per slot row of an image
constexpr float LOG2E = 1.4426950408889634f;

__device__ __forceinline__ f32x4v mfma32(u32x4v a, u32x4v b, f32x4v c) {
    // D(16x16) += A(16x32) B(32x16): lane l supplies A[i = l&15][k = 8 (l>>4) + e], B[k = 8 (l>>4) + e][j = l&15], e = 0..7;
    // register r of lane l is D[i = 4 (l>>4) + r][j = l&15]
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ uint32_t pk2(float a, float b) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{a, b}, bf16x2));
}
__device__ __forceinline__ u32x2v to_bf(const f32x4v& v) { return u32x2v{pk2(v.x, v.y), pk2(v.z, v.w)}; }

__device__ __forceinline__ float xg_max(float v) {              // over the 4 lanes l, l^16, l^32, l^48, without LDS
    u32x2v r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __builtin_fmaxf(__uint_as_float(r.x), __uint_as_float(r.y));
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __builtin_fmaxf(__uint_as_float(r.x), __uint_as_float(r.y));
}
__device__ __forceinline__ float xg_sum(float v) {
    u32x2v r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r.x) + __uint_as_float(r.y);
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r.x) + __uint_as_float(r.y);
}

// 16-byte chunk c (0..7) of row `row` of an image
__device__ __forceinline__ int xr(int row) {
    const int h = (row >> 1) & 7;
    return ((h & 3) << 1) | (h >> 2);
}
__device__ __forceinline__ uint32_t chunk_off(int row, int c) { return row * RB + ((c ^ xr(row)) << 4); }

__device__ __forceinline__ void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void wg_barrier() { asm volatile("s_barrier" ::: "memory"); }
__device__ __forceinline__ void wait_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// row operand (16 bytes: channels 32 kc + 8 g .. + 7) of slot row `row`, k-step kc
__device__ __forceinline__ u32x4v row_op(const char* img, int row, int kc, int gq) {
    return *(const lds_u32x4*)(img + chunk_off(row, 4 * kc + gq));
}
// column operand over the 32 slot rows of half h (k = slot 32 h + 16 (e >> 2) + 4 g + (e & 3)), channels 16 ct + (l & 15)
__device__ __forceinline__ u32x4v col_op(const char* img, int h, int ct, int lr, int gq) {
    const int r0 = 32 * h + 4 * gq + (lr >> 2), c = 2 * ct + ((lr & 3) >> 1), half = (lr & 1) * 8;
    const u32x2v lo = __builtin_bit_cast(u32x2v, __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(img + chunk_off(r0, c) + half)));
    const u32x2v hi = __builtin_bit_cast(u32x2v, __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(img + chunk_off(r0 + 16, c) + half)));
    return u32x4v{lo.x, lo.y, hi.x, hi.y};
}

}  // namespace
