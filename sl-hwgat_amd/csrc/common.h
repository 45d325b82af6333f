// Shared device helpers for the HWGAT gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "../../include/hwgat_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16_t;
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

#define HWGAT_WAVE 64

// The product library reads NO environment variables and keeps no mutable global state.  The A/B switches of the
// kernel lab (tools/*_lab.py: alternative tile kernels, split counts, ...) exist only in a `python sl-hwgat_amd/build.py
// --lab` build (-DHWGAT_LAB, a separate libhwgat_hip_lab.so); in the shipped build lab_env() is a constant null pointer
// and every branch behind it folds away.  hwgat_is_lab_build() tells a tool which of the two it loaded.
#ifdef HWGAT_LAB
#include <stdlib.h>
inline const char* lab_env(const char* name) { return getenv(name); }
#else
constexpr const char* lab_env(const char*) { return nullptr; }
#endif

// Dropout seeds.  Every `*_seed` argument of the C ABI is a SITE seed (a host integer, fixed for a model / block / dropout
// site); the seed a kernel hashes with is  site_seed + *seed_base  where `seed_base` is a DEVICE word read when the kernel
// runs (NULL = 0).  A train step captured in a HIP graph therefore replays with fresh masks: hwgat_seed_advance, a node of
// the same graph, rewrites the word before the kernels that read it.  Resolved once at kernel entry (a scalar load).
__device__ __forceinline__ uint32_t seed_base_of(const uint32_t* seed_base) { return seed_base ? *seed_base : 0u; }
#define HWGAT_RESOLVE_SEEDS2(p) do { const uint32_t sb__ = seed_base_of((p).seed_base); (p).pro_seed += sb__; (p).epi_seed += sb__; } while (0)
#define HWGAT_RESOLVE_SEED1(p) do { (p).pro_seed += seed_base_of((p).seed_base); } while (0)

// epilogue of the dW kernels: plain store into the split's image (deterministic mode) or a float atomic onto dW
#define HWGAT_TN_ACC(det, base, idx, v) do { if (det) (base)[idx] = (v); else atomicAdd((base) + (idx), (v)); } while (0)

// workspace of a deterministic dW launch: `cap` zero-filled images of N K floats at `dw`, `cap` x N floats at `db`
struct DetWs { float* dw; float* db; int cap; };
// out[i] += sum over the `cap` images ws[s * stride + i], s ascending (unused images are zero): gemm_f32.hip
// (`stride` floats between images; out[i] for i < count)
int hwgat_tn_det_reduce(const float* ws, float* out, int cap, int64_t stride, int64_t count, hipStream_t st);

#define HWGAT_LAUNCH_CHECK()                          \
    do {                                              \
        hipError_t e__ = hipGetLastError();           \
        return e__ == hipSuccess ? 0 : (int)e__;      \
    } while (0)

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// row index of accumulator register r of v_mfma_f32_32x32x2_f32 for lane half h
__device__ __forceinline__ constexpr int crow(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

template <typename T> struct io;
template <> struct io<float> {
    static constexpr int EPV = 4;                       // elements per 16-byte vector
    __device__ static __forceinline__ void load4(const float* p, float (&v)[4]) {
        f32x4 t = *reinterpret_cast<const f32x4*>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
    __device__ static __forceinline__ void store4(float* p, const float (&v)[4]) {
        f32x4 t = {v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(p) = t;
    }
    __device__ static __forceinline__ float ld(const float* p) { return *p; }
    __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct io<bf16_t> {
    static constexpr int EPV = 8;
    __device__ static __forceinline__ void load4(const bf16_t* p, float (&v)[4]) {
        bf16x4 t = *reinterpret_cast<const bf16x4*>(p);
        v[0] = (float)t.x; v[1] = (float)t.y; v[2] = (float)t.z; v[3] = (float)t.w;
    }
    __device__ static __forceinline__ void store4(bf16_t* p, const float (&v)[4]) {
        bf16x4 t = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
        *reinterpret_cast<bf16x4*>(p) = t;
    }
    __device__ static __forceinline__ float ld(const bf16_t* p) { return (float)*p; }
    __device__ static __forceinline__ void st(bf16_t* p, float v) { *p = (bf16_t)v; }
};

// Sum over aligned groups of W lanes (W = 8, 16 or 32) with DPP modifiers on plain vector adds -- no LDS crossbar
// (`__shfl_xor` is a ds_bpermute: an LDS round trip per step, ruinous in an exposed epilogue).  A DPP "row" is 16 lanes.
template <int W>
__device__ __forceinline__ float group_sum(float v) {
    auto dpp = [](float x, auto ctrl) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xf, 0xf, true));
    };
    v += dpp(v, std::integral_constant<int, 0xB1>{});            // quad_perm [1,0,3,2]
    v += dpp(v, std::integral_constant<int, 0x4E>{});            // quad_perm [2,3,0,1]
    v += dpp(v, std::integral_constant<int, 0x141>{});           // row_half_mirror: lanes 0-7 / 8-15 of a row reversed
    if constexpr (W >= 16) v += dpp(v, std::integral_constant<int, 0x140>{});   // row_mirror
    if constexpr (W >= 32) v += __shfl_xor(v, 16, 64);
    return v;
}

// butterfly all-reduce over the low `W` lanes-groups (W = 32 or 64)
template <int W>
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = W / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
