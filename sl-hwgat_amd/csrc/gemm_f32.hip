// fp32 Linear-layer GEMMs for HWGAT on gfx950, on v_mfma_f32_32x32x2_f32 (exact fp32
// fma chains, 157 TF dense peak), with the surrounding elementwise work fused in.
//
// Replaces, per PartAttentionBlock (reference hwgat/models/HWGATE.py):
//   qkv / proj / fc1 / fc2 nn.Linear forward      (:86, :115, :131, :134)
//   bias add, GELU (:132), Dropout (:116,:133,:135), residual adds (:217,:219)
//   and their autograd backward (dX, dW, db).
//
// Shapes are tall-skinny: M = B*F*K tokens (1e5..1e6), N,K in {128..1536}; weights are
// tiny and L2-resident, activations stream once.  Two kernels:
//
//  gemm_nt_k   C[M,N] = pro(A)[M,K] . W[N,K]^T  (+ epilogue).  Used for every forward
//              Linear and (with the weight transposed on the fly, it is <= 3 MB) every dX.
//              128x128 tile / 256 threads, wave tile 64x64 = 2x2 MFMA tiles, K slabs of
//              32 double-buffered in LDS (rows padded to 36 floats -> conflict-free
//              ds_read_b128 fragments, 4 k-steps per read), register-staged prefetch of the
//              next slab across tile boundaries, one barrier per slab.
//              prologues on A: LayerNorm (x-mean)*rstd*gamma+beta | dropout mask.
//              epilogues: bias | bias+dropout+residual | bias -> (h1, dropout(gelu(h1)))
//                         | gelu'(h1)*dropmask | none.
//  gemm_tn_k   dW[N,K] += A[M,N]^T . B[M,K], db[N] += colsum(A): split over M across the
//              chip (each block owns one 128x128 dW tile for a slice of M), fp32 atomics
//              to combine.  Both operands are consumed in their natural row-major layout
//              (lane = column), so LDS reads are conflict-free ds_read_b32 without padding.
//
// Dropout masks are a counter-based hash of (seed, element index): nothing is stored,
// backward regenerates the mask.
#include "common.h"
#include "fused_ops.h"

namespace {

struct NtArgs {
    const float* A; const float* W; const float* bias; float* C;
    float* C2; const float* res; const float* aux;
    const float* mean; const float* rstd; const float* gamma; const float* beta;
    int64_t M; int N, K;
    uint32_t pro_seed, epi_seed; float pro_p, epi_p;
};

constexpr int BM = 128, BN = 128, BK = 32, LDT = BK + 4;     // LDS tile row stride (floats)

template <int PRO, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_nt_k(NtArgs p) {
    __shared__ __attribute__((aligned(16))) float sm[2 * 2 * BM * LDT];       // [buf][A|W][128][36]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane & 31, hh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = p.N / BN;
    const int n_tiles = (int)(p.M / BM) * tiles_n;
    const int n_slab = p.K / BK;
    const int lrow = tid >> 3, lc4 = (tid & 7) * 4;          // this thread stages rows lrow+32*i, floats lc4..lc4+3
    const uint32_t pro_th = drop_thresh(p.pro_p);
    const float pro_sc = 1.0f / (1.0f - p.pro_p);

    f32x4 ra[4], rw[4];
    float ln_mean[4], ln_rstd[4];

    // XCD-aware tile order: blocks b, b+8, b+16, ... share an XCD (and its L2); give them the
    // n-tiles of ONE 128-row block of A, so A streams from HBM once and is re-read from L2.
    const int row_blocks = (int)(p.M / BM);
    const int swz_tiles = (row_blocks / 8) * 8 * tiles_n;
    auto tile_origin = [&](int t, int64_t& m0, int& n0) {
        int rb, nt;
        if (t < swz_tiles) {
            rb = (t / (8 * tiles_n)) * 8 + (t & 7);
            nt = (t >> 3) % tiles_n;
        } else {
            const int w = t - swz_tiles;
            rb = (row_blocks / 8) * 8 + w / tiles_n;
            nt = w % tiles_n;
        }
        m0 = (int64_t)rb * BM;
        n0 = nt * BN;
    };
    auto issue = [&](int64_t m0, int n0, int slab, bool new_tile) {
        const int k0 = slab * BK + lc4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t row = m0 + lrow + 32 * i;
            ra[i] = *reinterpret_cast<const f32x4*>(p.A + row * p.K + k0);
            rw[i] = *reinterpret_cast<const f32x4*>(p.W + (int64_t)(n0 + lrow + 32 * i) * p.K + k0);
            if constexpr (PRO == PRO_LN) {
                if (new_tile) { ln_mean[i] = p.mean[row]; ln_rstd[i] = p.rstd[row]; }
            }
        }
    };
    auto commit = [&](int buf, int64_t m0, int slab) {
        float* As = sm + buf * (2 * BM * LDT);
        float* Ws = As + BM * LDT;
        const int k0 = slab * BK + lc4;
        f32x4 g, b;
        if constexpr (PRO == PRO_LN) {
            g = *reinterpret_cast<const f32x4*>(p.gamma + k0);
            b = *reinterpret_cast<const f32x4*>(p.beta + k0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x4 a = ra[i];
            if constexpr (PRO == PRO_LN) {
                a.x = (a.x - ln_mean[i]) * ln_rstd[i] * g.x + b.x;
                a.y = (a.y - ln_mean[i]) * ln_rstd[i] * g.y + b.y;
                a.z = (a.z - ln_mean[i]) * ln_rstd[i] * g.z + b.z;
                a.w = (a.w - ln_mean[i]) * ln_rstd[i] * g.w + b.w;
            } else if constexpr (PRO == PRO_DROP) {
                if (pro_th) a *= drop_keep4(p.pro_seed, (uint64_t)(m0 + lrow + 32 * i) * p.K + k0, pro_th, pro_sc);
            }
            *reinterpret_cast<f32x4*>(As + (lrow + 32 * i) * LDT + lc4) = a;
            *reinterpret_cast<f32x4*>(Ws + (lrow + 32 * i) * LDT + lc4) = rw[i];
        }
    };

    int t = blockIdx.x;
    if (t >= n_tiles) return;
    int64_t m0; int n0;
    tile_origin(t, m0, n0);
    issue(m0, n0, 0, true);
    commit(0, m0, 0);
    __syncthreads();
    int buf = 0;

    while (true) {
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

        int tn = t; int64_t mn = m0; int nn = n0;
        for (int s = 0; s < n_slab; ++s) {
            // prefetch the next slab (possibly the next tile's first) into registers
            bool have_next = true, new_tile = false;
            int s_next = s + 1;
            if (s_next == n_slab) {
                tn = t + gridDim.x;
                have_next = tn < n_tiles;
                s_next = 0;
                new_tile = true;
                if (have_next) tile_origin(tn, mn, nn);
            }
            if (have_next) issue(mn, nn, s_next, new_tile);

            const float* As = sm + buf * (2 * BM * LDT);
            const float* Ws = As + BM * LDT;
            const float* ap = As + (wm * 64 + lq) * LDT + 4 * hh;
            const float* wp = Ws + (wn * 64 + lq) * LDT + 4 * hh;
#pragma unroll
            for (int kk = 0; kk < BK / 8; ++kk) {
                const f32x4 a0 = *reinterpret_cast<const f32x4*>(ap + 8 * kk);
                const f32x4 a1 = *reinterpret_cast<const f32x4*>(ap + 32 * LDT + 8 * kk);
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(wp + 8 * kk);
                const f32x4 b1 = *reinterpret_cast<const f32x4*>(wp + 32 * LDT + 8 * kk);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], b0[e], acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], b1[e], acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], b0[e], acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], b1[e], acc[1][1], 0, 0, 0);
                }
            }
            if (have_next) commit(buf ^ 1, mn, s_next);
            __syncthreads();
            buf ^= 1;
        }

        // ---- epilogue.  acc: lane (n = lq, hh), reg r -> C[m = crow(r,hh)][n].  Each wave parks
        // one 32x64 half of its tile in the LDS buffer that is idle now (buf^1), then walks it
        // row-wise so that every global access is a 16 B/lane, 256 B/row-segment vector.
        {
            const uint32_t epi_th = drop_thresh(p.epi_p);
            const float epi_sc = 1.0f / (1.0f - p.epi_p);
            constexpr int SLD = 68;                                   // staging row stride (floats)
            float* stg = sm + (buf ^ 1) * (2 * BM * LDT) + wave * (32 * SLD);
            const int er = lane >> 4, ec = (lane & 15) * 4;
            const int col = n0 + wn * 64 + ec;
            f32x4 bv = {0.f, 0.f, 0.f, 0.f};
            if constexpr (EPI == EPI_BIAS || EPI == EPI_BIAS_DROP_RES || EPI == EPI_BIAS_GELU_DROP)
                if (p.bias) bv = *reinterpret_cast<const f32x4*>(p.bias + col);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) stg[crow(r, hh) * SLD + j * 32 + lq] = acc[i][j][r];
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
#pragma unroll 2
                for (int ps = 0; ps < 8; ++ps) {
                    const int rr = ps * 4 + er;
                    const int64_t off = (m0 + wm * 64 + i * 32 + rr) * p.N + col;
                    f32x4 v = *reinterpret_cast<const f32x4*>(stg + rr * SLD + ec) + bv;
                    f32x4 dk = {1.f, 1.f, 1.f, 1.f};
                    if constexpr (EPI == EPI_BIAS_DROP_RES || EPI == EPI_BIAS_GELU_DROP || EPI == EPI_GELU_BWD) {
                        if (epi_th) dk = drop_keep4(p.epi_seed, (uint64_t)off, epi_th, epi_sc);
                    }
                    if constexpr (EPI == EPI_BIAS_DROP_RES) {
                        v = *reinterpret_cast<const f32x4*>(p.res + off) + v * dk;
                    } else if constexpr (EPI == EPI_BIAS_GELU_DROP) {
                        *reinterpret_cast<f32x4*>(p.C2 + off) = v;
                        v.x = gelu_f(v.x) * dk.x; v.y = gelu_f(v.y) * dk.y;
                        v.z = gelu_f(v.z) * dk.z; v.w = gelu_f(v.w) * dk.w;
                    } else if constexpr (EPI == EPI_GELU_BWD) {
                        const f32x4 h = *reinterpret_cast<const f32x4*>(p.aux + off);
                        v.x *= dk.x * gelu_grad(h.x); v.y *= dk.y * gelu_grad(h.y);
                        v.z *= dk.z * gelu_grad(h.z); v.w *= dk.w * gelu_grad(h.w);
                    }
                    *reinterpret_cast<f32x4*>(p.C + off) = v;
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
        __syncthreads();            // staging lives in buf^1, which the next tile's first commit overwrites
        t += gridDim.x;
        if (t >= n_tiles) break;
        m0 = mn; n0 = nn;
    }
}

// ------------------------------------------------------------------ dW / db
struct TnArgs {
    const float* A; const float* B; float* dW; float* db;
    const float* mean; const float* rstd; const float* gamma; const float* beta;
    int64_t M; int N, K;
    int n_split; int64_t rows_per_split;
    uint32_t pro_seed; float pro_p;
};

constexpr int TM = 32;                                         // rows of M per LDS stage

template <int PRO, bool BLN>
__global__ __launch_bounds__(256, 2) void gemm_tn_k(TnArgs p) {
    __shared__ __attribute__((aligned(16))) float sm[2 * 2 * TM * 128];      // [buf][A|B][32][128]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane & 31, hh = lane >> 5;
    const int wn = wave >> 1, wk = wave & 1;
    const int tiles_n = p.N / 128, tiles_k = p.K / 128, n_tiles = tiles_n * tiles_k;
    // blocks that share one M slice (all dW tiles of a split) sit on one XCD and run back to back,
    // so the slice of A / B they all read is served by that XCD's L2 after the first touch
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int tile = j % n_tiles;
    const int split = (j / n_tiles) * 8 + xcd;
    if (split >= p.n_split) return;
    const int n0 = (tile / tiles_k) * 128, k0 = (tile % tiles_k) * 128;
    const int64_t r_begin = (int64_t)split * p.rows_per_split;
    const int64_t r_end = r_begin + p.rows_per_split < p.M ? r_begin + p.rows_per_split : p.M;
    if (r_begin >= r_end) return;
    const int n_it = (int)((r_end - r_begin) / TM);

    const int lrow = tid >> 5, lc4 = (tid & 31) * 4;           // rows lrow + 8*i, floats lc4..lc4+3
    const uint32_t pro_th = drop_thresh(p.pro_p);
    const float pro_sc = 1.0f / (1.0f - p.pro_p);
    f32x4 ra[4], rb[4];
    float bm[4], bs[4];
    f32x4 colsum = {0.f, 0.f, 0.f, 0.f};
    f32x4 lg = {1.f, 1.f, 1.f, 1.f}, lb = {0.f, 0.f, 0.f, 0.f};
    if constexpr (BLN) {
        lg = *reinterpret_cast<const f32x4*>(p.gamma + k0 + lc4);
        lb = *reinterpret_cast<const f32x4*>(p.beta + k0 + lc4);
    }

    auto issue = [&](int it) {
        const int64_t r0 = r_begin + (int64_t)it * TM + lrow;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ra[i] = *reinterpret_cast<const f32x4*>(p.A + (r0 + 8 * i) * p.N + n0 + lc4);
            rb[i] = *reinterpret_cast<const f32x4*>(p.B + (r0 + 8 * i) * p.K + k0 + lc4);
            if constexpr (BLN) { bm[i] = p.mean[r0 + 8 * i]; bs[i] = p.rstd[r0 + 8 * i]; }
        }
    };
    auto commit = [&](int buf, int it) {
        float* As = sm + buf * (2 * TM * 128);
        float* Bs = As + TM * 128;
        const int64_t r0 = r_begin + (int64_t)it * TM + lrow;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x4 a = ra[i], b = rb[i];
            if constexpr (PRO == PRO_DROP)
                a *= drop_keep4(p.pro_seed, (uint64_t)(r0 + 8 * i) * p.N + n0 + lc4, pro_th, pro_sc);
            if constexpr (BLN) b = (b - bm[i]) * bs[i] * lg + lb;
            colsum += a;
            *reinterpret_cast<f32x4*>(As + (lrow + 8 * i) * 128 + lc4) = a;
            *reinterpret_cast<f32x4*>(Bs + (lrow + 8 * i) * 128 + lc4) = b;
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][jj][e] = 0.f;

    issue(0);
    commit(0, 0);
    __syncthreads();
    int buf = 0;
    for (int it = 0; it < n_it; ++it) {
        const bool have_next = it + 1 < n_it;
        if (have_next) issue(it + 1);
        const float* As = sm + buf * (2 * TM * 128) + wn * 64 + lq;
        const float* Bs = sm + buf * (2 * TM * 128) + TM * 128 + wk * 64 + lq;
        // software-pipelined operand fetch: the LDS reads of chunk c+1 are in flight while the
        // 16 MFMAs of chunk c issue (the compiler then waits with counted lgkmcnt, not 0)
        constexpr int CH = 4, NCH = TM / 2 / CH;
        float fa0[2][CH], fa1[2][CH], fb0[2][CH], fb1[2][CH];
        auto fetch = [&](int c, int slot) {
#pragma unroll
            for (int e = 0; e < CH; ++e) {
                const int ro = (2 * (c * CH + e) + hh) * 128;
                fa0[slot][e] = As[ro]; fa1[slot][e] = As[ro + 32];
                fb0[slot][e] = Bs[ro]; fb1[slot][e] = Bs[ro + 32];
            }
        };
        fetch(0, 0);
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            if (c + 1 < NCH) fetch(c + 1, (c + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);          // keep the prefetch above this chunk's MFMAs
#pragma unroll
            for (int e = 0; e < CH; ++e) {
                const float a0 = fa0[c & 1][e], a1 = fa1[c & 1][e], b0 = fb0[c & 1][e], b1 = fb1[c & 1][e];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (have_next) commit(buf ^ 1, it + 1);
        __syncthreads();
        buf ^= 1;
    }
    // D[i = n][j = k]: lane (k = lq, hh), reg r -> dW[n = crow(r,hh)][k]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wn * 64 + i * 32 + crow(r, hh);
                const int k = k0 + wk * 64 + jj * 32 + lq;
                atomicAdd(p.dW + (int64_t)n * p.K + k, acc[i][jj][r]);
            }
    if (p.db != nullptr && k0 == 0) {                           // one k-tile column owns the bias gradient
        float* red = sm;                                        // [8][128] partial column sums
        __syncthreads();
        *reinterpret_cast<f32x4*>(red + lrow * 128 + lc4) = colsum;
        __syncthreads();
        if (tid < 128) {
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) s += red[q * 128 + tid];
            atomicAdd(p.db + n0 + tid, s);
        }
    }
}

// out[c][r] = in[r][c]  (weights only: <= 3 MB)
__global__ void transpose_k(const float* __restrict__ in, float* __restrict__ out, int R, int C) {
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int i = threadIdx.y; i < 32; i += 8)
        if (r0 + i < R && c0 + threadIdx.x < C) tile[i][threadIdx.x] = in[(int64_t)(r0 + i) * C + c0 + threadIdx.x];
    __syncthreads();
    for (int i = threadIdx.y; i < 32; i += 8)
        if (c0 + i < C && r0 + threadIdx.x < R) out[(int64_t)(c0 + i) * R + r0 + threadIdx.x] = tile[threadIdx.x][i];
}

template <int PRO>
int launch_nt(const NtArgs& a, int epi, int grid, hipStream_t st) {
    switch (epi) {
        case EPI_BIAS: gemm_nt_k<PRO, EPI_BIAS><<<grid, 256, 0, st>>>(a); break;
        case EPI_BIAS_DROP_RES: gemm_nt_k<PRO, EPI_BIAS_DROP_RES><<<grid, 256, 0, st>>>(a); break;
        case EPI_BIAS_GELU_DROP: gemm_nt_k<PRO, EPI_BIAS_GELU_DROP><<<grid, 256, 0, st>>>(a); break;
        case EPI_GELU_BWD: gemm_nt_k<PRO, EPI_GELU_BWD><<<grid, 256, 0, st>>>(a); break;
        case EPI_NONE: gemm_nt_k<PRO, EPI_NONE><<<grid, 256, 0, st>>>(a); break;
        default: return HWGAT_EINVAL;
    }
    HWGAT_LAUNCH_CHECK();
}

}  // namespace

extern "C" int hwgat_linear_nt_f32(const float* A, const float* W, const float* bias, float* C, int64_t M,
                                   int N, int K, int pro, const float* mean, const float* rstd,
                                   const float* gamma, const float* beta, uint32_t pro_seed, float pro_p,
                                   int epi, const float* res, float* C2, const float* aux, uint32_t epi_seed,
                                   float epi_p, void* stream) {
    if (!A || !W || !C || M <= 0 || N <= 0 || K <= 0) return HWGAT_EINVAL;
    if (M % BM || N % BN || K % BK || (M / BM) * (int64_t)(N / BN) > 0x7fffffff) return HWGAT_ESHAPE;
    if (pro == PRO_LN && (!mean || !rstd || !gamma || !beta)) return HWGAT_EINVAL;
    if (epi == EPI_BIAS_DROP_RES && !res) return HWGAT_EINVAL;
    if (epi == EPI_BIAS_GELU_DROP && !C2) return HWGAT_EINVAL;
    if (epi == EPI_GELU_BWD && !aux) return HWGAT_EINVAL;
    if (pro_p < 0.f || pro_p >= 1.f || epi_p < 0.f || epi_p >= 1.f) return HWGAT_EINVAL;
    NtArgs a{A, W, bias, C, C2, res, aux, mean, rstd, gamma, beta, M, N, K, pro_seed, epi_seed, pro_p, epi_p};
    const int64_t tiles = (M / BM) * (N / BN);
    const int grid = (int)(tiles < 512 ? tiles : 512);          // 2 resident blocks per CU, persistent over tiles
    hipStream_t st = (hipStream_t)stream;
    switch (pro) {
        case PRO_NONE: return launch_nt<PRO_NONE>(a, epi, grid, st);
        case PRO_LN: return launch_nt<PRO_LN>(a, epi, grid, st);
        case PRO_DROP: return launch_nt<PRO_DROP>(a, epi, grid, st);
        default: return HWGAT_EINVAL;
    }
}

extern "C" int hwgat_linear_tn_f32(const float* A, const float* B, float* dW, float* db, int64_t M, int N,
                                   int K, uint32_t pro_seed, float pro_p, const float* mean,
                                   const float* rstd, const float* gamma, const float* beta, void* stream) {
    if (!A || !B || !dW || M <= 0 || N <= 0 || K <= 0) return HWGAT_EINVAL;
    if (mean && (!rstd || !gamma || !beta)) return HWGAT_EINVAL;
    if (M % TM || N % 128 || K % 128) return HWGAT_ESHAPE;
    if (pro_p < 0.f || pro_p >= 1.f) return HWGAT_EINVAL;
    const int n_tiles = (N / 128) * (K / 128);
    // Blocks are equal-sized and 2 are resident per CU (512 slots), so they execute in rounds:
    // pick the number of M slices so that blocks = n_split * n_tiles is an exact multiple of 512
    // (no nearly-empty last round), about 2-3 rounds, and every slice is >= 16 LDS stages deep.
    auto gcd = [](int a, int b) { while (b) { int t = a % b; a = b; b = t; } return a; };
    const int r_min = n_tiles / gcd(n_tiles, 512);
    int r = r_min;
    while (r < 2) r += r_min;
    int64_t want = (int64_t)512 * r / n_tiles;
    const int64_t max_split = M / (TM * 16) > 0 ? M / (TM * 16) : 1;
    if (want > max_split) want = max_split;
    if (want < 1) want = 1;
    int64_t rows = (M + want - 1) / want;
    rows = (rows + TM - 1) / TM * TM;
    const int n_split = (int)((M + rows - 1) / rows);
    TnArgs a{A, B, dW, db, mean, rstd, gamma, beta, M, N, K, n_split, rows, pro_seed, pro_p};
    const int grid = ((n_split + 7) / 8) * 8 * n_tiles;
    hipStream_t st = (hipStream_t)stream;
    if (pro_p > 0.f) {
        if (mean) gemm_tn_k<PRO_DROP, true><<<grid, 256, 0, st>>>(a);
        else gemm_tn_k<PRO_DROP, false><<<grid, 256, 0, st>>>(a);
    } else {
        if (mean) gemm_tn_k<PRO_NONE, true><<<grid, 256, 0, st>>>(a);
        else gemm_tn_k<PRO_NONE, false><<<grid, 256, 0, st>>>(a);
    }
    HWGAT_LAUNCH_CHECK();
}

extern "C" int hwgat_transpose_f32(const float* in, float* out, int R, int C, void* stream) {
    if (!in || !out || R <= 0 || C <= 0) return HWGAT_EINVAL;
    dim3 grid((C + 31) / 32, (R + 31) / 32), block(32, 8);
    transpose_k<<<grid, block, 0, (hipStream_t)stream>>>(in, out, R, C);
    HWGAT_LAUNCH_CHECK();
}

// expose the dropout hash so host tests can reproduce masks bit for bit
__global__ void drop_mask_k(float* out, int64_t n, uint32_t seed, float p) {
    const uint32_t th = drop_thresh(p);
    const float sc = 1.0f / (1.0f - p);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        out[i] = drop_keep(seed, (uint64_t)i, th, sc);
}
extern "C" int hwgat_dropout_mask_f32(float* out, int64_t n, uint32_t seed, float p, void* stream) {
    if (!out || n <= 0 || p < 0.f || p >= 1.f) return HWGAT_EINVAL;
    const int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    drop_mask_k<<<grid, 256, 0, (hipStream_t)stream>>>(out, n, seed, p);
    HWGAT_LAUNCH_CHECK();
}
