// argument block of the bf16 NT linear kernels (gemm_bf16.hip, gemm_bf16_nt256.hip)
#pragma once
#include "common.h"

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct NtArgsB {
    const bf16_t* A; const bf16_t* W; const float* bias; bf16_t* C;
    bf16_t* C2; const bf16_t* res; const bf16_t* aux;
    const float* mean; const float* rstd; const float* gamma; const float* beta;
    int64_t M; int N, K;
    uint32_t pro_seed, epi_seed; float pro_p, epi_p;
    int64_t row0;          // RAGGED tail launches: global index of this launch's first row (dropout hash)
    // EPI_BIAS_DROP_RES only: row sums / sums of squares of the (bf16-rounded) output and the TemporalMerging store,
    // exactly as NtArgs in gemm_f32.h
    float* stat_sum; float* stat_sq;
    int mg_F, mg_K;
    const uint32_t* seed_base;   // device word added to the site seeds at kernel entry (NULL = 0), see common.h
};

// output row m -> row of the merged (B, F/2, K, 2N) tensor for a lane that walks rows m, m + step, ... of one tile:
// ONE pair of 32-bit divisions, then carries (see MergeWalk in gemm_f32.h)
struct MergeWalk {
    int k, f; int64_t b; int F, K, step;
    __device__ __forceinline__ void start(int64_t m, int F_, int K_, int step_) {
        F = F_; K = K_; step = step_;
        const uint32_t mm = (uint32_t)m, fr = mm / (uint32_t)K_;
        k = (int)(mm - fr * (uint32_t)K_);
        const uint32_t bb = fr / (uint32_t)F_;
        f = (int)(fr - bb * (uint32_t)F_);
        b = bb;
    }
    __device__ __forceinline__ int64_t mrow() const { return (b * (F >> 1) + (f >> 1)) * K + k; }
    __device__ __forceinline__ int64_t off(int N) const { return mrow() * (2 * (int64_t)N) + (int64_t)(f & 1) * N; }
    __device__ __forceinline__ void next() {
        k += step;
        const bool wk = k >= K;
        k -= wk ? K : 0;
        f += wk ? 1 : 0;
        const bool wf = f >= F;
        f -= wf ? F : 0;
        b += wf ? 1 : 0;
    }
};

__device__ __forceinline__ void unpack8(u32x4 r, float (&v)[8]) {
    v[0] = __uint_as_float(r.x << 16); v[1] = __uint_as_float(r.x & 0xffff0000u);
    v[2] = __uint_as_float(r.y << 16); v[3] = __uint_as_float(r.y & 0xffff0000u);
    v[4] = __uint_as_float(r.z << 16); v[5] = __uint_as_float(r.z & 0xffff0000u);
    v[6] = __uint_as_float(r.w << 16); v[7] = __uint_as_float(r.w & 0xffff0000u);
}
__device__ __forceinline__ uint32_t pack2(float a, float b) {
    bf16x2 t = {(bf16_t)a, (bf16_t)b};
    return *reinterpret_cast<uint32_t*>(&t);
}
__device__ __forceinline__ u32x4 pack8(const float (&v)[8]) {
    u32x4 r = {pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7])};
    return r;
}

// 256x256 C tile, 4 waves x (128x128), one wave per SIMD (gemm_bf16_nt256.hip); needs M % 256 == N % 256 == K % 64 == 0
// dW / db launch record of the bf16 weight-gradient kernels (gemm_tn_bf16_k, gemm_tn256_bf16_k)
struct TnArgsB {
    const bf16_t* A; const bf16_t* B; float* dW; float* db;
    const float* mean; const float* rstd; const float* gamma; const float* beta;
    int64_t M; int N, K;
    int n_split; int64_t rows_per_split;
    uint32_t pro_seed; float pro_p;
    int64_t row0;          // see NtArgsB
    const uint32_t* seed_base;   // device word added to the site seeds at kernel entry (NULL = 0), see common.h
    // deterministic accumulation (hwgat_linear_tn_*_det): det_dw != NULL -> the block of M split s stores its partial dW tile
    // PLAINLY into image s of a zero-filled workspace (det_dw + s N K) and its partial bias gradient into det_db + s N
    // instead of adding them to dW / db with float atomics; tn_det_reduce_k then adds the images in split order.
    // det_cap = images the workspace holds (a launcher whose split count exceeds it returns HWGAT_ESHAPE).
    float* det_dw; float* det_db; int det_cap;
};

// defined in gemm_bf16_tn256.hip: 256x256 dW tiles; N % 256 == K % 256 == M % 32 == 0
int hwgat_launch_tn256_bf16(TnArgsB a, hipStream_t st);

int hwgat_launch_nt256_bf16(const NtArgsB& a, int pro, int epi, hipStream_t st);

// defined in gemm_bf16_tn8w.hip: 256x256 dW tiles on eight waves with LDS-DMA operand streaming; plain operands only
bool hwgat_tn8w_bf16_takes(int64_t M, int N, int K, float pro_p, const float* mean);
int hwgat_launch_tn8w_bf16(TnArgsB a, hipStream_t st, float* ws = nullptr);   // ws: hwgat_tn8w_bf16_ws_floats() floats, or NULL (atomics)
int64_t hwgat_tn8w_bf16_ws_floats(int64_t M, int N, int K);

// defined in gemm_bf16_nt8w.hip: the same tile on eight waves with LDS-DMA operand streaming and a register epilogue;
// M % 256 == N % 256 == K % 128 == 0, no A-side prologue (PRO_NONE / PRO_LN_FOLD)
bool hwgat_nt8w_bf16_takes(const NtArgsB& a, int pro, int epi);
int hwgat_launch_nt8w_bf16(const NtArgsB& a, int pro, int epi, hipStream_t st);
