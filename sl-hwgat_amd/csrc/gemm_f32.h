// argument block of the fp32 NT linear kernels (gemm_f32.hip)
#pragma once
#include "common.h"

struct NtArgs {
    const float* A; const float* W; const float* bias; float* C;
    float* C2; const float* res; const float* aux;
    const float* mean; const float* rstd; const float* gamma; const float* beta;
    int64_t M; int N, K;
    uint32_t pro_seed, epi_seed; float pro_p, epi_p;
};


