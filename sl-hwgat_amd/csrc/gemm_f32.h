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
};



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
