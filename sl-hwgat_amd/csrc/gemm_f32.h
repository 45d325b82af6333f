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
};

// destination of output row m under the merged store (frames F, tokens per frame K, row width N): merged row index
// and element offset of its first column
__device__ __forceinline__ void merge_row(int64_t m, int F, int K, int N, int64_t& mrow, int64_t& off) {
    const int64_t fr = m / K;                 // b * F + f
    const int k = (int)(m - fr * K);
    const int64_t b = fr / F;
    const int f = (int)(fr - b * F);
    mrow = (b * (F >> 1) + (f >> 1)) * K + k;
    off = mrow * (2 * (int64_t)N) + (int64_t)(f & 1) * N;
}

struct TnArgs {
    const float* A; const float* B; float* dW; float* db;
    const float* mean; const float* rstd; const float* gamma; const float* beta;
    int64_t M; int N, K;
    int n_split; int64_t rows_per_split;
    uint32_t pro_seed; float pro_p;
    int64_t row0;          // see NtArgs
};


// 256x256 dW tile, 4 waves x (128x128), one wave per SIMD, pinned MFMA/memory interleave (gemm_f32_tn256.hip);
// needs N % 256 == K % 256 == 0; fills in n_split / rows_per_split itself
int hwgat_launch_tn256(TnArgs a, hipStream_t st);

// 256x256 C tile, 4 waves x (128x128), one wave per SIMD, pinned MFMA/memory interleave (gemm_f32_nt256.hip);
// needs M % 256 == N % 256 == K % 32 == 0; same prologues / epilogues as gemm_nt_k
int hwgat_launch_nt256(const NtArgs& a, int pro, int epi, hipStream_t st);
