// Per-step weight preparation for the HWGAT blocks on gfx950: every derived copy of the fp32 master weights that the
// fused linears consume -- the activation-dtype copy (bf16 activations), the transposed copy for the dX launches, and the
// LayerNorm-folded copy with its row sums (hwgat_ln_fold) -- for ALL blocks of the model in ONE launch, driven by a table
// in device memory.  Before this, a train step issued 16 fold + 32 transpose + 48 cast launches of ~5 us each: nothing at
// the 102 ms fp32 step, but 0.5 ms of a 10-17 ms bf16 step and 96 of its ~400 host-side launches.
//
// Reference: nothing to replace -- torch keeps no such copies; these are operands of the kernels that stand in for
// hwgat/models/HWGATE.py:86,115,131-135 and their backward.
#include "common.h"

namespace {

constexpr int OP_COPY = 0, OP_TRANSPOSE = 1, OP_FOLD = 2;

template <typename T>
__global__ __launch_bounds__(256) void weight_prep_k(const hwgat_prep_entry* __restrict__ table, int n) {
    __shared__ float tile[32][33];
    // entry of this workgroup: the last one whose first_block <= blockIdx.x (n is a few dozen: linear scan by every thread)
    int e = 0;
    while (e + 1 < n && table[e + 1].first_block <= (int)blockIdx.x) ++e;
    const hwgat_prep_entry en = table[e];
    const int b = blockIdx.x - en.first_block;
    const int N = en.N, K = en.K;
    T* out = (T*)en.out;
    if (en.op == OP_FOLD) {                                   // (W o gamma, s, c) of hwgat_ln_fold: 4 weight rows per workgroup
        const int row = b * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
        if (row >= N) return;
        const float* w = en.W + (int64_t)row * K;
        T* wf = out + (int64_t)row * K;
        float ss = 0.f, cc = 0.f;
        for (int k = lane; k < K; k += 64) {
            const float wv = w[k];
            const T r = (T)(wv * en.gamma[k]);
            wf[k] = r;
            ss += (float)r;
            cc = fmaf(en.beta[k], wv, cc);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { ss += __shfl_xor(ss, o); cc += __shfl_xor(cc, o); }
        if (lane == 0) { en.s[row] = ss; en.c[row] = cc + (en.bias ? en.bias[row] : 0.f); }
        return;
    }
    const int tiles_k = (K + 31) / 32;
    const int r0 = (b / tiles_k) * 32, c0 = (b % tiles_k) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    if (en.op == OP_COPY) {
        for (int i = ty; i < 32; i += 8)
            if (r0 + i < N && c0 + tx < K) out[(int64_t)(r0 + i) * K + c0 + tx] = (T)en.W[(int64_t)(r0 + i) * K + c0 + tx];
        return;
    }
    for (int i = ty; i < 32; i += 8)                          // OP_TRANSPOSE: out[k][n] = W[n][k]
        if (r0 + i < N && c0 + tx < K) tile[i][tx] = en.W[(int64_t)(r0 + i) * K + c0 + tx];
    __syncthreads();
    for (int i = ty; i < 32; i += 8)
        if (c0 + i < K && r0 + tx < N) out[(int64_t)(c0 + i) * N + r0 + tx] = (T)tile[tx][i];
}

}  // namespace

extern "C" int hwgat_weight_prep(const hwgat_prep_entry* table, int n, int total_blocks, int dtype, void* stream) {
    if (!table || n <= 0 || total_blocks <= 0) return HWGAT_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == HWGAT_F32) weight_prep_k<float><<<total_blocks, 256, 0, st>>>(table, n);
    else if (dtype == HWGAT_BF16) weight_prep_k<bf16_t><<<total_blocks, 256, 0, st>>>(table, n);
    else return HWGAT_EDTYPE;
    HWGAT_LAUNCH_CHECK();
}
