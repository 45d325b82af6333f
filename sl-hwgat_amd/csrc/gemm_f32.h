// argument block of the fp32 NT linear kernels (gemm_f32.hip)
#pragma once
#include "common.h"

struct NtArgs {
    const float* A; const float* W; const float* bias; float* C;
    float* C2; const float* res; const float* aux;
    const float* mean; const float* rstd; const float* gamma; const float* beta;
    int64_t M; int N, K;
    uint32_t pro_seed, epi_seed; float pro_p, epi_p;
    int64_t row0;          // RAGGED tail launches: global index of this launch's first row (dropout hash)
    // EPI_BIAS_DROP_RES only (the two launches whose output feeds a LayerNorm): per-row sum / sum of squares of the
    // OUTPUT accumulated into stat_sum / stat_sq (zeroed by the caller; hwgat_ln_finalize turns them into mean / rstd),
    // and, with mg_K > 0, the output stored in the TemporalMerging layout (HWGATE.py:55-63): row (b, f, k) of width N
    // goes to row (b, f/2, k), columns (f & 1) N .. of a (B, F/2, K, 2N) tensor; statistics are then per MERGED row.
    float* stat_sum; float* stat_sq;
    int mg_F, mg_K;
    const uint32_t* seed_base;   // device word added to the site seeds at kernel entry (NULL = 0), see common.h
};

// destination of output row m under the merged store: row (b, f, k) -> merged row (b, f/2, k), column offset (f & 1) N.
// For a lane that walks rows m, m + step, m + 2 step, ... of one tile: ONE pair of 32-bit divisions
// (rows of a launch are < 2^31), then carries.  step < K.
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

struct TnArgs {
    const float* A; const float* B; float* dW; float* db;
    const float* mean; const float* rstd; const float* gamma; const float* beta;
    int64_t M; int N, K;
    int n_split; int64_t rows_per_split;
    uint32_t pro_seed; float pro_p;
    int64_t row0;          // see NtArgs
    const uint32_t* seed_base;   // device word added to the site seeds at kernel entry (NULL = 0), see common.h
    // deterministic accumulation (hwgat_linear_tn_*_det): det_dw != NULL -> the block of M split s stores its partial dW tile
    // PLAINLY into image s of a zero-filled workspace (det_dw + s N K) and its partial bias gradient into det_db + s N
    // instead of adding them to dW / db with float atomics; tn_det_reduce_k then adds the images in split order.
    // det_cap = images the workspace holds (a launcher whose split count exceeds it returns HWGAT_ESHAPE).
    float* det_dw; float* det_db; int det_cap;
};


// 256x256 dW tile, 4 waves x (128x128), one wave per SIMD, pinned MFMA/memory interleave (gemm_f32_tn256.hip);
// needs N % 256 == K % 256 == 0; fills in n_split / rows_per_split itself
// ws / ws_floats: optional workspace for the slab form (partial tiles + fixed-order reduction instead of global atomics)
int hwgat_launch_tn256(TnArgs a, hipStream_t st, float* ws = nullptr, int64_t ws_floats = 0);
int64_t hwgat_tn256_ws_floats(int64_t M, int N, int K);

// 256x256 C tile, 4 waves x (128x128), one wave per SIMD, pinned MFMA/memory interleave (gemm_f32_nt256.hip);
// needs M % 256 == N % 256 == K % 32 == 0; same prologues / epilogues as gemm_nt_k
int hwgat_launch_nt256(const NtArgs& a, int pro, int epi, hipStream_t st);

