// LayerNorm forward/backward and the fused final LayerNorm + token mean-pool for
// HWGAT on gfx950.  Reference: nn.LayerNorm at hwgat/models/HWGATE.py:162,166
// (per block, applied at :203 and :219) and :327 (final, applied at :353) with
// the AvgPool1d over all f*K tokens at :354.
//
// All of these are pure HBM-streaming kernels: a row of d in {128..1024} values
// is owned by LPR = min(64, d/4) lanes, each lane holding d/(4*LPR) 16-byte
// chunks, so every global access is a full 16 B/lane coalesced vector; statistics
// are two-pass in registers (mean, then centred variance) in fp32.
#include "common.h"
#include "fused_ops.h"

namespace {

constexpr float LN_EPS = 1e-5f;

template <int D> struct RowMap {
    static constexpr int NCH = D / 4;                         // 4-element chunks per row
    static constexpr int LPR = NCH < 64 ? NCH : 64;           // lanes per row
    static constexpr int RPW = 64 / LPR;                      // rows per wave pass
    static constexpr int CPL = NCH / LPR;                     // chunks per lane
};

typedef uint32_t u32x2w __attribute__((ext_vector_type(2)));
// (nontemporal hints: every activation row is read once and written once per kernel)
template <typename T, int D>
__device__ __forceinline__ void load_row(const T* row, int sub, float (&v)[RowMap<D>::CPL][4]) {
#pragma unroll
    for (int c = 0; c < RowMap<D>::CPL; ++c) {
        if constexpr (sizeof(T) == 4) {
            const f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(row + (c * RowMap<D>::LPR + sub) * 4));
            v[c][0] = t.x; v[c][1] = t.y; v[c][2] = t.z; v[c][3] = t.w;
        } else {
            const u32x2w t = __builtin_nontemporal_load(reinterpret_cast<const u32x2w*>(row + (c * RowMap<D>::LPR + sub) * 4));
            v[c][0] = __uint_as_float(t.x << 16); v[c][1] = __uint_as_float(t.x & 0xffff0000u);
            v[c][2] = __uint_as_float(t.y << 16); v[c][3] = __uint_as_float(t.y & 0xffff0000u);
        }
    }
}
template <typename T, int D>
__device__ __forceinline__ void store_row(T* row, int sub, const float (&v)[RowMap<D>::CPL][4]) {
#pragma unroll
    for (int c = 0; c < RowMap<D>::CPL; ++c) {
        if constexpr (sizeof(T) == 4) {
            const f32x4 t = {v[c][0], v[c][1], v[c][2], v[c][3]};
            __builtin_nontemporal_store(t, reinterpret_cast<f32x4*>(row + (c * RowMap<D>::LPR + sub) * 4));
        } else {
            const bf16x4 t = {(bf16_t)v[c][0], (bf16_t)v[c][1], (bf16_t)v[c][2], (bf16_t)v[c][3]};
            __builtin_nontemporal_store(__builtin_bit_cast(u32x2w, t), reinterpret_cast<u32x2w*>(row + (c * RowMap<D>::LPR + sub) * 4));
        }
    }
}
template <int D>
__device__ __forceinline__ void load_vec(const float* p, int sub, float (&v)[RowMap<D>::CPL][4]) {
#pragma unroll
    for (int c = 0; c < RowMap<D>::CPL; ++c) io<float>::load4(p + (c * RowMap<D>::LPR + sub) * 4, v[c]);
}

template <int D>
__device__ __forceinline__ void row_stats(const float (&v)[RowMap<D>::CPL][4], float& mean, float& rstd) {
    using M = RowMap<D>;
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < M::CPL; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) s += v[c][e];
    mean = wave_sum<M::LPR>(s) * (1.0f / D);
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < M::CPL; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float t = v[c][e] - mean; q += t * t; }
    rstd = rsqrtf(wave_sum<M::LPR>(q) * (1.0f / D) + LN_EPS);
}

// ------------------------------------------------------------------ forward
template <typename T, int D>
__global__ __launch_bounds__(256) void ln_fwd_k(const T* __restrict__ x, const float* __restrict__ gamma,
                                                const float* __restrict__ beta, T* __restrict__ y,
                                                float* __restrict__ mean_o, float* __restrict__ rstd_o,
                                                int64_t N) {
    using M = RowMap<D>;
    const int lane = threadIdx.x & 63;
    const int sub = lane % M::LPR, rsub = lane / M::LPR;
    float g[M::CPL][4], b[M::CPL][4];
    load_vec<D>(gamma, sub, g);
    load_vec<D>(beta, sub, b);
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwave = (int64_t)gridDim.x * 4;
    for (int64_t r0 = wave * M::RPW; r0 < N; r0 += nwave * M::RPW) {
        const int64_t r = r0 + rsub;
        if (r >= N) continue;                      // only when N % RPW != 0 (whole row group idle)
        float v[M::CPL][4];
        load_row<T, D>(x + r * D, sub, v);
        float mean, rstd;
        row_stats<D>(v, mean, rstd);
        if (y != nullptr) {                        // y == NULL: statistics only (fused consumers normalise on the fly)
#pragma unroll
            for (int c = 0; c < M::CPL; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e) v[c][e] = (v[c][e] - mean) * rstd * g[c][e] + b[c][e];
            store_row<T, D>(y + r * D, sub, v);
        }
        if (sub == 0) { mean_o[r] = mean; rstd_o[r] = rstd; }
    }
}

// ------------------------------------------------------------------ backward
// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma
// MASK: also write dxm = dx * dropout-keep(seed, element index) -- the gradient this tensor's PRODUCER needs in front of
// its dropout (dX and dW of the linear whose output was dropped).  The hash is evaluated once here, in an HBM-bound
// kernel whose vector units idle, instead of in every tile of the two GEMM loaders that consume the masked gradient
// (8 evaluations per element at stage 2, none of them hidden behind the MFMAs).
// XN: also write LN(x) = xhat * gamma + beta (the layer input of the Linear behind this LayerNorm) for that Linear's
// weight-gradient launch, which then takes both operands plain (LDS-DMA, gemm_bf16_tn8w.hip) instead of normalising x in
// its loaders: xhat is in registers here anyway, the pass is HBM-bound with idle vector units.
// DET: dgamma / dbeta point at a workspace of gridDim.x images of 2 D floats; block b stores its column sums plainly into
// image b (no atomics) and hwgat_tn_det_reduce adds the images in block order: bit-reproducible parameter gradients.
template <typename T, int D, bool RES, bool MASK, bool XN = false, bool DET = false>
__global__ __launch_bounds__(256) void ln_bwd_k(const T* __restrict__ dy, const T* __restrict__ x,
                                                const float* __restrict__ mean_i,
                                                const float* __restrict__ rstd_i,
                                                const float* __restrict__ gamma, const T* __restrict__ dres,
                                                T* __restrict__ dx, float* __restrict__ dgamma,
                                                float* __restrict__ dbeta, int64_t N, T* __restrict__ dxm,
                                                uint32_t mseed, float mp, const float* __restrict__ beta = nullptr,
                                                T* __restrict__ xn = nullptr, const uint32_t* __restrict__ sbase = nullptr) {
    using M = RowMap<D>;
    if constexpr (MASK) mseed += seed_base_of(sbase);
    __shared__ float red[2][4][D];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int sub = lane % M::LPR, rsub = lane / M::LPR;
    float g[M::CPL][4], dg[M::CPL][4], db[M::CPL][4];
    load_vec<D>(gamma, sub, g);
    float bt[M::CPL][4];
    if constexpr (XN) load_vec<D>(beta, sub, bt);
#pragma unroll
    for (int c = 0; c < M::CPL; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) { dg[c][e] = 0.f; db[c][e] = 0.f; }
    const int64_t wave = (int64_t)blockIdx.x * 4 + wv;
    const int64_t nwave = (int64_t)gridDim.x * 4;
    for (int64_t r0 = wave * M::RPW; r0 < N; r0 += nwave * M::RPW) {
        const int64_t r = r0 + rsub;
        if (r >= N) continue;
        float xv[M::CPL][4], dv[M::CPL][4];
        load_row<T, D>(x + r * D, sub, xv);
        load_row<T, D>(dy + r * D, sub, dv);
        const float mean = mean_i[r], rstd = rstd_i[r];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int c = 0; c < M::CPL; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float xh = (xv[c][e] - mean) * rstd;
                const float gg = dv[c][e] * g[c][e];
                dg[c][e] += dv[c][e] * xh;
                db[c][e] += dv[c][e];
                xv[c][e] = xh;
                dv[c][e] = gg;
                s1 += gg;
                s2 += gg * xh;
            }
        s1 = wave_sum<M::LPR>(s1) * (1.0f / D);
        s2 = wave_sum<M::LPR>(s2) * (1.0f / D);
        if constexpr (XN) {
            float nv[M::CPL][4];
#pragma unroll
            for (int c = 0; c < M::CPL; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e) nv[c][e] = xv[c][e] * g[c][e] + bt[c][e];
            store_row<T, D>(xn + r * D, sub, nv);
        }
        if constexpr (RES) {
            float rv[M::CPL][4];
            load_row<T, D>(dres + r * D, sub, rv);
#pragma unroll
            for (int c = 0; c < M::CPL; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e) dv[c][e] = rstd * (dv[c][e] - s1 - xv[c][e] * s2) + rv[c][e];
        } else {
#pragma unroll
            for (int c = 0; c < M::CPL; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e) dv[c][e] = rstd * (dv[c][e] - s1 - xv[c][e] * s2);
        }
        store_row<T, D>(dx + r * D, sub, dv);
        if constexpr (MASK) {
            const uint32_t th = drop_thresh(mp);
            const float sc = 1.0f / (1.0f - mp);
#pragma unroll
            for (int c = 0; c < M::CPL; ++c) {
                const f32x4 k = drop_keep4(mseed, (uint64_t)(r * D + (c * M::LPR + sub) * 4), th, sc);
                dv[c][0] *= k.x; dv[c][1] *= k.y; dv[c][2] *= k.z; dv[c][3] *= k.w;
            }
            store_row<T, D>(dxm + r * D, sub, dv);
        }
    }
    // fold the RPW row groups of a wave, then the 4 waves, then one atomic per column per block
    if constexpr (M::RPW == 2) {
#pragma unroll
        for (int c = 0; c < M::CPL; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                dg[c][e] += __shfl_xor(dg[c][e], 32, 64);
                db[c][e] += __shfl_xor(db[c][e], 32, 64);
            }
    }
    if (rsub == 0) {
#pragma unroll
        for (int c = 0; c < M::CPL; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                red[0][wv][(c * M::LPR + sub) * 4 + e] = dg[c][e];
                red[1][wv][(c * M::LPR + sub) * 4 + e] = db[c][e];
            }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < D; i += 256) {
        const float sg = red[0][0][i] + red[0][1][i] + red[0][2][i] + red[0][3][i];
        const float sb = red[1][0][i] + red[1][1][i] + red[1][2][i] + red[1][3][i];
        if constexpr (DET) {
            dgamma[(int64_t)blockIdx.x * (2 * D) + i] = sg;
            dgamma[(int64_t)blockIdx.x * (2 * D) + D + i] = sb;
        } else {
            atomicAdd(dgamma + i, sg);
            atomicAdd(dbeta + i, sb);
        }
    }
}

// ------------------------------------------------------------------ LN + pool
// xhat_sum[b][c] += sum over this wave's tokens of (x - mean) * rstd
template <typename T, int D>
__global__ __launch_bounds__(256) void lnpool_fwd_k(const T* __restrict__ x, float* __restrict__ xhat_sum,
                                                    float* __restrict__ mean_o, float* __restrict__ rstd_o,
                                                    int n_tok, int chunks, float* __restrict__ partial) {
    using M = RowMap<D>;
    __shared__ float red[4][D];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int sub = lane % M::LPR, rsub = lane / M::LPR;
    const int b = blockIdx.x / chunks, ch = blockIdx.x % chunks;
    const int per = (n_tok + chunks - 1) / chunks;
    const int t0 = ch * per, t1 = min(n_tok, t0 + per);
    float acc[M::CPL][4];
#pragma unroll
    for (int c = 0; c < M::CPL; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[c][e] = 0.f;
    for (int t = t0 + wv * M::RPW + rsub; t < t1; t += 4 * M::RPW) {
        const int64_t r = (int64_t)b * n_tok + t;
        float v[M::CPL][4];
        load_row<T, D>(x + r * D, sub, v);
        float mean, rstd;
        row_stats<D>(v, mean, rstd);
#pragma unroll
        for (int c = 0; c < M::CPL; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[c][e] += (v[c][e] - mean) * rstd;
        if (sub == 0) { mean_o[r] = mean; rstd_o[r] = rstd; }
    }
    if constexpr (M::RPW == 2) {
#pragma unroll
        for (int c = 0; c < M::CPL; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[c][e] += __shfl_xor(acc[c][e], 32, 64);
    }
    if (rsub == 0) {
#pragma unroll
        for (int c = 0; c < M::CPL; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) red[wv][(c * M::LPR + sub) * 4 + e] = acc[c][e];
    }
    __syncthreads();
    // `partial` (deterministic form): every block stores its own sum, lnpool_reduce_k adds the chunks of a clip in index
    // order -- the same bits on every run; otherwise one fp32 atomic per block and channel (order varies run to run)
    for (int i = threadIdx.x; i < D; i += 256) {
        const float v = red[0][i] + red[1][i] + red[2][i] + red[3][i];
        if (partial) partial[(int64_t)blockIdx.x * D + i] = v;
        else atomicAdd(xhat_sum + (int64_t)b * D + i, v);
    }
}

// xhat_sum[b][i] = sum over the clip's chunks, in chunk order (fixed summation order)
__global__ void lnpool_reduce_k(const float* __restrict__ partial, float* __restrict__ xhat_sum, int chunks, int d) {
    const int b = blockIdx.x;
    for (int i = threadIdx.x; i < d; i += blockDim.x) {
        float s = 0.f;
        for (int c = 0; c < chunks; ++c) s += partial[((int64_t)b * chunks + c) * d + i];
        xhat_sum[(int64_t)b * d + i] = s;
    }
}

// the upstream gradient of every token of clip b is the same vector g[b] (fp32)
template <typename T, int D>
__global__ __launch_bounds__(256) void lnpool_bwd_k(const float* __restrict__ g, const T* __restrict__ x,
                                                    const float* __restrict__ mean_i,
                                                    const float* __restrict__ rstd_i, T* __restrict__ dx,
                                                    int n_tok, int chunks, T* __restrict__ dxm, uint32_t mseed,
                                                    float mp, const uint32_t* __restrict__ sbase) {
    using M = RowMap<D>;
    mseed += seed_base_of(sbase);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int sub = lane % M::LPR, rsub = lane / M::LPR;
    const int b = blockIdx.x / chunks, ch = blockIdx.x % chunks;
    const int per = (n_tok + chunks - 1) / chunks;
    const int t0 = ch * per, t1 = min(n_tok, t0 + per);
    float gv[M::CPL][4];
    load_vec<D>(g + (int64_t)b * D, sub, gv);
    float s1 = 0.f;
#pragma unroll
    for (int c = 0; c < M::CPL; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) s1 += gv[c][e];
    s1 = wave_sum<M::LPR>(s1) * (1.0f / D);
    for (int t = t0 + wv * M::RPW + rsub; t < t1; t += 4 * M::RPW) {
        const int64_t r = (int64_t)b * n_tok + t;
        float v[M::CPL][4];
        load_row<T, D>(x + r * D, sub, v);
        const float mean = mean_i[r], rstd = rstd_i[r];
        float s2 = 0.f;
#pragma unroll
        for (int c = 0; c < M::CPL; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[c][e] = (v[c][e] - mean) * rstd; s2 += gv[c][e] * v[c][e]; }
        s2 = wave_sum<M::LPR>(s2) * (1.0f / D);
#pragma unroll
        for (int c = 0; c < M::CPL; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[c][e] = rstd * (gv[c][e] - s1 - v[c][e] * s2);
        store_row<T, D>(dx + r * D, sub, v);
        if (dxm != nullptr) {                                   // see ln_bwd_k<.., MASK>
            const uint32_t th = drop_thresh(mp);
            const float sc = 1.0f / (1.0f - mp);
#pragma unroll
            for (int c = 0; c < M::CPL; ++c) {
                const f32x4 k = drop_keep4(mseed, (uint64_t)(r * D + (c * M::LPR + sub) * 4), th, sc);
                v[c][0] *= k.x; v[c][1] *= k.y; v[c][2] *= k.z; v[c][3] *= k.w;
            }
            store_row<T, D>(dxm + r * D, sub, v);
        }
    }
}

inline int ln_grid(int64_t N, int rpw) {
    const int64_t need = (N + 4 * rpw - 1) / (4 * rpw);
    return (int)(need < 2048 ? (need < 1 ? 1 : need) : 2048);
}

template <typename T>
int ln_fwd_t(const void* x, const float* gm, const float* bt, void* y, float* mean, float* rstd, int64_t N,
             int d, hipStream_t st) {
#define GO(D) ln_fwd_k<T, D><<<ln_grid(N, RowMap<D>::RPW), 256, 0, st>>>((const T*)x, gm, bt, (T*)y, mean, rstd, N)
    switch (d) {
        case 128: GO(128); break;
        case 256: GO(256); break;
        case 512: GO(512); break;
        case 1024: GO(1024); break;
        default: return HWGAT_ESHAPE;
    }
#undef GO
    HWGAT_LAUNCH_CHECK();
}
template <typename T, bool RES, bool MASK = false, bool XN = false, bool DET = false>
int ln_bwd_t(const void* dy, const void* x, const float* mean, const float* rstd, const float* gm,
             const void* dres, void* dx, float* dg, float* db, int64_t N, int d, hipStream_t st,
             void* dxm = nullptr, uint32_t mseed = 0, float mp = 0.f, const float* bt = nullptr, void* xn = nullptr,
             const uint32_t* sbase = nullptr, float* det_ws = nullptr) {
    // DET: the kernel writes per-block images into det_ws (1024 x 2 d floats), then two fixed-order reductions
    int grid = 0;
#define GO(D)                                                                                                        \
    grid = ln_grid(N, RowMap<D>::RPW) < 1024 ? ln_grid(N, RowMap<D>::RPW) : 1024;                                    \
    ln_bwd_k<T, D, RES, MASK, XN, DET><<<grid, 256, 0, st>>>((const T*)dy, (const T*)x, mean, rstd, gm, (const T*)dres, (T*)dx, \
                                          DET ? det_ws : dg, DET ? det_ws : db, N, (T*)dxm, mseed, mp, bt, (T*)xn, sbase)
    switch (d) {
        case 128: GO(128); break;
        case 256: GO(256); break;
        case 512: GO(512); break;
        case 1024: GO(1024); break;
        default: return HWGAT_ESHAPE;
    }
#undef GO
    if constexpr (DET) {
        int rc = hwgat_tn_det_reduce(det_ws, dg, grid, 2 * (int64_t)d, d, st);
        if (rc) return rc;
        return hwgat_tn_det_reduce(det_ws + d, db, grid, 2 * (int64_t)d, d, st);
    }
    HWGAT_LAUNCH_CHECK();
}
inline int pool_chunks(int B, int n_tok) {
    int c = 2048 / (B > 0 ? B : 1);
    if (c < 1) c = 1;
    const int maxc = (n_tok + 7) / 8;
    return c < maxc ? c : (maxc < 1 ? 1 : maxc);
}

// (sum, sum of squares) of the rows of a d-wide tensor, as accumulated by the producing linear's epilogue
// (hwgat_linear_nt_f32_ex) -> (mean, rstd) in place.  Biased variance like nn.LayerNorm; E[x^2] - mean^2 in fp32 is
// accurate to ~1e-7 (1 + mean^2 / var), ample for activations whose mean is not orders of magnitude above their spread.
__global__ void ln_finalize_k(float* __restrict__ sum_mean, float* __restrict__ sq_rstd, int64_t n, float inv_d) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float mean = sum_mean[i] * inv_d;
        const float var = fmaxf(sq_rstd[i] * inv_d - mean * mean, 0.f);
        sum_mean[i] = mean;
        sq_rstd[i] = rsqrtf(var + LN_EPS);
    }
}

// LayerNorm folded into the linear that consumes it (PRO_LN_FOLD, fused_ops.h):
//   LN(x) W^T + b = rstd_m (x (W o gamma)^T - mean_m s) + c,   s_n = sum_k W'[n,k],  c_n = b_n + sum_k beta_k W[n,k]
// One wave per output row n: W' = W o gamma stored in the activation dtype, s summed over the STORED (rounded)
// values so that the mean term cancels exactly against the product the MFMA forms, c from the fp32 weights.
template <typename T>
__global__ __launch_bounds__(256) void ln_fold_k(const float* __restrict__ W, const float* __restrict__ bias,
                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                 T* __restrict__ Wf, float* __restrict__ s, float* __restrict__ c, int N, int K) {
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (n >= N) return;
    const float* w = W + (int64_t)n * K;
    T* wf = Wf + (int64_t)n * K;
    float ss = 0.f, cc = 0.f;
    for (int k = lane; k < K; k += 64) {
        const float wv = w[k];
        const T r = (T)(wv * gamma[k]);
        wf[k] = r;
        ss += (float)r;
        cc = fmaf(beta[k], wv, cc);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { ss += __shfl_xor(ss, o); cc += __shfl_xor(cc, o); }
    if (lane == 0) { s[n] = ss; c[n] = cc + (bias ? bias[n] : 0.f); }
}

}  // namespace

extern "C" int hwgat_ln_fold(const float* W, const float* bias, const float* gamma, const float* beta, int N, int K,
                             void* Wf, float* s, float* c, int dtype, void* stream) {
    if (!W || !gamma || !beta || !Wf || !s || !c || N <= 0 || K <= 0) return HWGAT_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int grid = (N + 3) / 4;
    if (dtype == HWGAT_F32) ln_fold_k<float><<<grid, 256, 0, st>>>(W, bias, gamma, beta, (float*)Wf, s, c, N, K);
    else if (dtype == HWGAT_BF16) ln_fold_k<bf16_t><<<grid, 256, 0, st>>>(W, bias, gamma, beta, (bf16_t*)Wf, s, c, N, K);
    else return HWGAT_EINVAL;
    HWGAT_LAUNCH_CHECK();
}

extern "C" int hwgat_ln_finalize(float* sum_mean, float* sq_rstd, int64_t n, int d, void* stream) {
    if (!sum_mean || !sq_rstd || n <= 0 || d <= 0) return HWGAT_EINVAL;
    const int grid = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
    ln_finalize_k<<<grid, 256, 0, (hipStream_t)stream>>>(sum_mean, sq_rstd, n, 1.0f / d);
    HWGAT_LAUNCH_CHECK();
}

extern "C" int hwgat_ln_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean,
                            float* rstd, int64_t N, int d, int dtype, void* stream) {
    if (!x || !gamma || !beta || !mean || !rstd || N <= 0) return HWGAT_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == HWGAT_F32) return ln_fwd_t<float>(x, gamma, beta, y, mean, rstd, N, d, st);
    if (dtype == HWGAT_BF16) return ln_fwd_t<bf16_t>(x, gamma, beta, y, mean, rstd, N, d, st);
    return HWGAT_EDTYPE;
}

extern "C" int hwgat_ln_bwd(const void* dy, const void* x, const float* mean, const float* rstd,
                            const float* gamma, const void* dres, void* dx, float* dgamma, float* dbeta,
                            int64_t N, int d, int dtype, void* stream) {
    if (!dy || !x || !mean || !rstd || !gamma || !dx || !dgamma || !dbeta || N <= 0) return HWGAT_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == HWGAT_F32)
        return dres ? ln_bwd_t<float, true>(dy, x, mean, rstd, gamma, dres, dx, dgamma, dbeta, N, d, st)
                    : ln_bwd_t<float, false>(dy, x, mean, rstd, gamma, dres, dx, dgamma, dbeta, N, d, st);
    if (dtype == HWGAT_BF16)
        return dres ? ln_bwd_t<bf16_t, true>(dy, x, mean, rstd, gamma, dres, dx, dgamma, dbeta, N, d, st)
                    : ln_bwd_t<bf16_t, false>(dy, x, mean, rstd, gamma, dres, dx, dgamma, dbeta, N, d, st);
    return HWGAT_EDTYPE;
}

extern "C" int hwgat_ln_bwd_masked(const void* dy, const void* x, const float* mean, const float* rstd,
                                   const float* gamma, const void* dres, void* dx, float* dgamma, float* dbeta,
                                   int64_t N, int d, int dtype, void* dx_masked, uint32_t mask_seed, float mask_p,
                                   const uint32_t* seed_base, void* stream) {
    if (!dy || !x || !mean || !rstd || !gamma || !dx || !dgamma || !dbeta || !dres || !dx_masked || N <= 0) return HWGAT_EINVAL;
    if (mask_p <= 0.f || mask_p >= 1.f) return HWGAT_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == HWGAT_F32)
        return ln_bwd_t<float, true, true>(dy, x, mean, rstd, gamma, dres, dx, dgamma, dbeta, N, d, st, dx_masked, mask_seed, mask_p, nullptr, nullptr, seed_base);
    if (dtype == HWGAT_BF16)
        return ln_bwd_t<bf16_t, true, true>(dy, x, mean, rstd, gamma, dres, dx, dgamma, dbeta, N, d, st, dx_masked, mask_seed, mask_p, nullptr, nullptr, seed_base);
    return HWGAT_EDTYPE;
}

// hwgat_ln_bwd (+ optional masked copy) that ALSO writes xn = LN(x) (see ln_bwd_k<.., XN>); dres required
extern "C" int hwgat_ln_bwd_xn(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                               const float* beta, const void* dres, void* dx, float* dgamma, float* dbeta, int64_t N, int d,
                               int dtype, void* dx_masked, uint32_t mask_seed, float mask_p, void* xn,
                               const uint32_t* seed_base, void* stream) {
    if (!dy || !x || !mean || !rstd || !gamma || !beta || !dx || !dgamma || !dbeta || !dres || !xn || N <= 0) return HWGAT_EINVAL;
    if (dx_masked && (mask_p <= 0.f || mask_p >= 1.f)) return HWGAT_EINVAL;
    hipStream_t st = (hipStream_t)stream;
#define GO(T)                                                                                                             \
    return dx_masked ? ln_bwd_t<T, true, true, true>(dy, x, mean, rstd, gamma, dres, dx, dgamma, dbeta, N, d, st, dx_masked, \
                                                     mask_seed, mask_p, beta, xn, seed_base)                              \
                     : ln_bwd_t<T, true, false, true>(dy, x, mean, rstd, gamma, dres, dx, dgamma, dbeta, N, d, st, nullptr, \
                                                      0, 0.f, beta, xn)
    if (dtype == HWGAT_F32) { GO(float); }
    if (dtype == HWGAT_BF16) { GO(bf16_t); }
#undef GO
    return HWGAT_EDTYPE;
}

extern "C" int hwgat_lnpool_partial_rows(int B, int n_tok) { return (B <= 0 || n_tok <= 0) ? HWGAT_EINVAL : pool_chunks(B, n_tok); }

extern "C" int hwgat_lnpool_fwd_det(const void* x, float* xhat_sum, float* mean, float* rstd, int B, int n_tok, int d,
                                    int dtype, float* partial, void* stream) {
    if (!x || !xhat_sum || !mean || !rstd || B <= 0 || n_tok <= 0) return HWGAT_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int ch = pool_chunks(B, n_tok);
#define GO(T, D) lnpool_fwd_k<T, D><<<B * ch, 256, 0, st>>>((const T*)x, xhat_sum, mean, rstd, n_tok, ch, partial)
#define SW(T)                                  \
    switch (d) {                               \
        case 128: GO(T, 128); break;           \
        case 256: GO(T, 256); break;           \
        case 512: GO(T, 512); break;           \
        case 1024: GO(T, 1024); break;         \
        default: return HWGAT_ESHAPE;          \
    }
    if (dtype == HWGAT_F32) { SW(float) }
    else if (dtype == HWGAT_BF16) { SW(bf16_t) }
    else return HWGAT_EDTYPE;
#undef GO
    if (partial) lnpool_reduce_k<<<B, 256, 0, st>>>(partial, xhat_sum, ch, d);
    HWGAT_LAUNCH_CHECK();
}

// Deterministic LayerNorm backward: every option of the three entry points above in one (beta / xn: both or neither; dres
// optional unless dx_masked or xn is given; dx_masked optional), dgamma / dbeta summed in a fixed order through `ws`
// (hwgat_ln_bwd_det_bytes(d) bytes, need not be zeroed).
extern "C" int64_t hwgat_ln_bwd_det_bytes(int d) { return d > 0 ? (int64_t)1024 * 2 * d * 4 : 0; }
extern "C" int hwgat_ln_bwd_det(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                                const float* beta, const void* dres, void* dx, float* dgamma, float* dbeta, int64_t N, int d,
                                int dtype, void* dx_masked, uint32_t mask_seed, float mask_p, void* xn,
                                const uint32_t* seed_base, float* ws, int64_t ws_bytes, void* stream) {
    if (!dy || !x || !mean || !rstd || !gamma || !dx || !dgamma || !dbeta || !ws || N <= 0) return HWGAT_EINVAL;
    if ((xn != nullptr) != (beta != nullptr)) return HWGAT_EINVAL;
    if ((dx_masked || xn) && !dres) return HWGAT_EINVAL;
    if (dx_masked && (mask_p <= 0.f || mask_p >= 1.f)) return HWGAT_EINVAL;
    if (ws_bytes < hwgat_ln_bwd_det_bytes(d)) return HWGAT_ESHAPE;
    hipStream_t st = (hipStream_t)stream;
#define GO(T)                                                                                                              \
    if (xn) return dx_masked ? ln_bwd_t<T, true, true, true, true>(dy, x, mean, rstd, gamma, dres, dx, dgamma, dbeta, N, d, st, \
                                                                   dx_masked, mask_seed, mask_p, beta, xn, seed_base, ws)  \
                             : ln_bwd_t<T, true, false, true, true>(dy, x, mean, rstd, gamma, dres, dx, dgamma, dbeta, N, d, st, \
                                                                    nullptr, 0, 0.f, beta, xn, seed_base, ws);             \
    if (dx_masked) return ln_bwd_t<T, true, true, false, true>(dy, x, mean, rstd, gamma, dres, dx, dgamma, dbeta, N, d, st,    \
                                                               dx_masked, mask_seed, mask_p, nullptr, nullptr, seed_base, ws); \
    return dres ? ln_bwd_t<T, true, false, false, true>(dy, x, mean, rstd, gamma, dres, dx, dgamma, dbeta, N, d, st, nullptr, \
                                                        0, 0.f, nullptr, nullptr, nullptr, ws)                             \
                : ln_bwd_t<T, false, false, false, true>(dy, x, mean, rstd, gamma, dres, dx, dgamma, dbeta, N, d, st, nullptr, \
                                                         0, 0.f, nullptr, nullptr, nullptr, ws)
    if (dtype == HWGAT_F32) { GO(float); }
    if (dtype == HWGAT_BF16) { GO(bf16_t); }
#undef GO
    return HWGAT_EDTYPE;
}

extern "C" int hwgat_lnpool_fwd(const void* x, float* xhat_sum, float* mean, float* rstd, int B, int n_tok, int d,
                                int dtype, void* stream) {
    return hwgat_lnpool_fwd_det(x, xhat_sum, mean, rstd, B, n_tok, d, dtype, nullptr, stream);
}

extern "C" int hwgat_lnpool_bwd_masked(const float* g, const void* x, const float* mean, const float* rstd, void* dx,
                                       int B, int n_tok, int d, int dtype, void* dx_masked, uint32_t mask_seed,
                                       float mask_p, const uint32_t* seed_base, void* stream);

extern "C" int hwgat_lnpool_bwd(const float* g, const void* x, const float* mean, const float* rstd, void* dx,
                                int B, int n_tok, int d, int dtype, void* stream) {
    return hwgat_lnpool_bwd_masked(g, x, mean, rstd, dx, B, n_tok, d, dtype, nullptr, 0, 0.f, nullptr, stream);
}

extern "C" int hwgat_lnpool_bwd_masked(const float* g, const void* x, const float* mean, const float* rstd, void* dx,
                                       int B, int n_tok, int d, int dtype, void* dxm, uint32_t mseed, float mp,
                                       const uint32_t* seed_base, void* stream) {
    if (!g || !x || !mean || !rstd || !dx || B <= 0 || n_tok <= 0) return HWGAT_EINVAL;
    if (dxm && (mp <= 0.f || mp >= 1.f)) return HWGAT_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int ch = pool_chunks(B, n_tok);
#define GO(T, D) lnpool_bwd_k<T, D><<<B * ch, 256, 0, st>>>(g, (const T*)x, mean, rstd, (T*)dx, n_tok, ch, (T*)dxm, mseed, mp, seed_base)
    if (dtype == HWGAT_F32) { SW(float) }
    else if (dtype == HWGAT_BF16) { SW(bf16_t) }
    else return HWGAT_EDTYPE;
#undef GO
#undef SW
    HWGAT_LAUNCH_CHECK();
}
