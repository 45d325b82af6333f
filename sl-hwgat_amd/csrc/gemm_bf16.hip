// bf16 Linear-layer GEMMs for HWGAT on gfx950 (BASELINE config 3: bf16 activations with
// MFMA projections), on v_mfma_f32_32x32x16_bf16 (fp32 accumulate) with the same fused
// prologues / epilogues as the fp32 family (gemm_f32.hip).
//
// With bf16 MFMA (16x the fp32 rate) these kernels are HBM/L2-streaming kernels, not
// compute kernels: the tile loop is organised for bytes in flight, the MFMAs are noise.
//
//  gemm_nt_bf16_k  C[M,N] = pro(A)[M,K] . W[N,K]^T (+epilogue); A, W, C bf16, bias/LN fp32.
//                  128x128 tile, K slabs of 64 (one 128-byte line per row) double-buffered in LDS
//                  with 144-byte rows (conflict-free ds_read_b128 operand fragments: a lane's
//                  fragment is 8 consecutive k of its row, exactly one 16-byte read).
//  gemm_tn_bf16_k  dW[N,K](fp32) += A[M,N]^T . B[M,K], db += colsum(A); A, B bf16.  The MFMA
//                  operand needs 8 consecutive m per lane for a fixed column, i.e. a transposed
//                  read of the row-major tile: ds_read_b64_tr_b16 (gfx950 hardware transpose)
//                  on LDS rows padded to 320 bytes (4 rows of a 4x16 block land in disjoint
//                  bank quarters -> conflict-free).
#include <stdlib.h>
#include "common.h"
#include "fused_ops.h"
#include "gemm_bf16.h"

namespace {

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

// 128x128 tile, 4 waves, K slabs of 64 (144-byte LDS rows), 72 KB -> 2 blocks / CU.
// Measured alternatives (all slower or neutral on MI355X, config 3; NT time per step):
//   K32 x 3 blocks/CU 19.8 ms (vs 20.2) | two-slab register prefetch ring: neutral | 8-wave 256x256: 26.4 ms
//   | 128x256 (64x128 wave tiles, 2 blocks/CU): 22.0 ms (vs 18.9).
using NtB64 = TileCfg<2, 2, 2, 2, 64, 2, 8, 8>;

// A (tile, K-slab) cursor over the work of one persistent block.
struct SlabIt {
    int t, s, n0;
    int64_t m0;
    bool valid;
};

// RAGGED: launch over the last M % 128 rows of a token count that is not a multiple of the tile (see the fp32
// family, gemm_f32.hip): clamped loads, guarded stores, dropout hashed with the global row index.
template <int PRO, int EPI, typename C, bool RAGGED = false, int STAT = 0>      // STAT: see gemm_nt256_bf16_k
__global__ __launch_bounds__(C::THREADS, C::OCC * C::THREADS / 256) void gemm_nt_bf16_k(NtArgsB p) {
    HWGAT_RESOLVE_SEEDS2(p);
    constexpr int BM = C::BM, BN = C::BN, BK = C::BK, LDT = C::LDT, PA = C::PA, PW = C::PW, RPP = C::RPP;
    constexpr int TMW = C::TMW, TNW = C::TNW;
    __shared__ __attribute__((aligned(16))) bf16_t sm[2 * (BM + BN) * LDT];   // [buf][A rows | W rows][LDT]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane & 31, hh = lane >> 5;
    const int wm = wave / C::WN, wn = wave % C::WN;
    const int tiles_n = p.N / BN;
    const int row_blocks = RAGGED ? (int)((p.M + BM - 1) / BM) : (int)(p.M / BM);
    const int n_tiles = row_blocks * tiles_n;
    const int64_t m_last = p.M - 1;
    const int n_slab = p.K / BK;
    const int lrow = tid / C::TPR, lc8 = (tid % C::TPR) * 8;   // rows lrow + RPP*i, bf16 columns lc8..lc8+7
    const uint32_t pro_th = drop_thresh(p.pro_p);
    const float pro_sc = 1.0f / (1.0f - p.pro_p);

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
    auto advance = [&](SlabIt& it) {
        if (++it.s == n_slab) {
            it.s = 0;
            it.t += gridDim.x;
            it.valid = it.t < n_tiles;
            if (it.valid) tile_origin(it.t, it.m0, it.n0);
        }
    };
    // These kernels are latency-bound (16 MFMAs per slab): the loads of TWO slabs are kept in flight
    // in two register sets, each committed to LDS one iteration after the other was issued.
    auto issue = [&](u32x4 (&ra)[PA], u32x4 (&rw)[PW], float (&lm)[PA], float (&lr)[PA], const SlabIt& it) {
        const int k0 = it.s * BK + lc8;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            int64_t row = it.m0 + lrow + RPP * i;
            if constexpr (RAGGED) row = row < m_last ? row : m_last;
            ra[i] = *reinterpret_cast<const u32x4*>(p.A + row * p.K + k0);
            if constexpr (PRO == PRO_LN) { lm[i] = p.mean[row]; lr[i] = p.rstd[row]; }
        }
#pragma unroll
        for (int i = 0; i < PW; ++i)
            rw[i] = *reinterpret_cast<const u32x4*>(p.W + (int64_t)(it.n0 + lrow + RPP * i) * p.K + k0);
    };

    auto commit = [&](u32x4 (&ra)[PA], u32x4 (&rw)[PW], float (&lm)[PA], float (&lr)[PA], const SlabIt& it, int buf) {
        bf16_t* As = sm + buf * ((BM + BN) * LDT);
        bf16_t* Ws = As + BM * LDT;
        const int k0 = it.s * BK + lc8;
        float g[8], b[8];
        if constexpr (PRO == PRO_LN) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { g[e] = p.gamma[k0 + e]; b[e] = p.beta[k0 + e]; }
        }
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            u32x4 a = ra[i];
            if constexpr (PRO == PRO_LN) {
                float v[8];
                unpack8(a, v);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (v[e] - lm[i]) * lr[i] * g[e] + b[e];
                a = pack8(v);
            } else if constexpr (PRO == PRO_DROP) {
                if (pro_th) {
                    const uint64_t e0 = (uint64_t)((RAGGED ? p.row0 : 0) + it.m0 + lrow + RPP * i) * p.K + k0;
                    const f32x4 k0v = drop_keep4(p.pro_seed, e0, pro_th, pro_sc);
                    const f32x4 k1v = drop_keep4(p.pro_seed, e0 + 4, pro_th, pro_sc);
                    float v[8];
                    unpack8(a, v);
                    v[0] *= k0v.x; v[1] *= k0v.y; v[2] *= k0v.z; v[3] *= k0v.w;
                    v[4] *= k1v.x; v[5] *= k1v.y; v[6] *= k1v.z; v[7] *= k1v.w;
                    a = pack8(v);
                }
            }
            *reinterpret_cast<u32x4*>(As + (lrow + RPP * i) * LDT + lc8) = a;
        }
#pragma unroll
        for (int i = 0; i < PW; ++i) *reinterpret_cast<u32x4*>(Ws + (lrow + RPP * i) * LDT + lc8) = rw[i];
    };

    f32x16 acc[TMW][TNW];
    auto compute = [&](int buf, bool first) {
        if (first) {
#pragma unroll
            for (int i = 0; i < TMW; ++i)
#pragma unroll
                for (int j = 0; j < TNW; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        }
        const bf16_t* As = sm + buf * ((BM + BN) * LDT);
        const bf16_t* Ws = As + BM * LDT;
        const bf16_t* ap = As + (wm * (TMW * 32) + lq) * LDT + 8 * hh;
        const bf16_t* wp = Ws + (wn * (TNW * 32) + lq) * LDT + 8 * hh;
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
            bf16x8 af[TMW], bfr[TNW];
#pragma unroll
            for (int i = 0; i < TMW; ++i) af[i] = *reinterpret_cast<const bf16x8*>(ap + i * 32 * LDT + 16 * kk);
#pragma unroll
            for (int j = 0; j < TNW; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(wp + j * 32 * LDT + 16 * kk);
#pragma unroll
            for (int i = 0; i < TMW; ++i)
#pragma unroll
                for (int j = 0; j < TNW; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
    };
    // epilogue through an idle LDS buffer (fp32 staging, 32 x SW per wave), then row-wise 16 B/lane bf16 stores
    // (8 columns per lane).  The residual / pre-activation operand of a whole piece is requested BEFORE the piece is
    // parked: all its passes are in flight together instead of one HBM round trip per pass (PMC on the GELU-backward
    // launch: waves parked in s_waitcnt 52 % of the time at 2.1 TB/s; the plain product on the same shape streams 5).
    auto epilogue = [&](int64_t m0, int n0, int idle) {
        const uint32_t epi_th = drop_thresh(p.epi_p);
        const float epi_sc = 1.0f / (1.0f - p.epi_p);
        constexpr int SW = ((BM + BN) * LDT * 2 >= (C::THREADS / 64) * 32 * 68 * 4) ? 64 : 32;
        constexpr int SLD = SW + 4, LPR = SW / 8, RPS = 64 / LPR, NPS = 32 / RPS;
        float* stg = reinterpret_cast<float*>(sm + idle * ((BM + BN) * LDT)) + wave * (32 * SLD);
        const int er = lane / LPR, ec = (lane % LPR) * 8;
        float* rowstat = reinterpret_cast<float*>(sm + idle * ((BM + BN) * LDT)) + (C::THREADS / 64) * (32 * SLD);   // [BM][2]
        if constexpr (STAT == X_STAT || STAT == X_STAT_MERGE) {
            static_assert(((C::THREADS / 64) * 32 * SLD + 2 * BM) * 4 <= (BM + BN) * LDT * 2, "no room for the row statistics");
            for (int q = tid; q < 2 * BM; q += C::THREADS) rowstat[q] = 0.f;
            __syncthreads();
        }
#pragma unroll
        for (int jc = 0; jc < TNW * 32 / SW; ++jc) {
            const int col = n0 + wn * (TNW * 32) + jc * SW + ec;
            f32x4 bv0 = {0.f, 0.f, 0.f, 0.f}, bv1 = bv0;
            if constexpr (epi_has_bias(EPI))
                if (p.bias) { bv0 = *reinterpret_cast<const f32x4*>(p.bias + col); bv1 = *reinterpret_cast<const f32x4*>(p.bias + col + 4); }
            f32x4 sv0 = bv0, sv1 = bv0, cv0 = bv0, cv1 = bv0;  // X_LNFOLD: s_n and c_n of the lane's 8 columns
            if constexpr (STAT == X_LNFOLD) {
                sv0 = *reinterpret_cast<const f32x4*>(p.gamma + col); sv1 = *reinterpret_cast<const f32x4*>(p.gamma + col + 4);
                cv0 = *reinterpret_cast<const f32x4*>(p.beta + col); cv1 = *reinterpret_cast<const f32x4*>(p.beta + col + 4);
            }
#pragma unroll
            for (int i = 0; i < TMW; ++i) {
                float rr_[NPS], rm[NPS];                        // X_LNFOLD: rstd and mean of the piece's rows
                if constexpr (STAT == X_LNFOLD) {
#pragma unroll
                    for (int ps = 0; ps < NPS; ++ps) {
                        int64_t grow = m0 + wm * (TMW * 32) + i * 32 + ps * RPS + er;
                        if constexpr (RAGGED) grow = grow < m_last ? grow : m_last;
                        rr_[ps] = p.rstd[grow]; rm[ps] = p.mean[grow];
                    }
                }
                MergeWalk mw;
                if constexpr (STAT == X_STAT_MERGE) mw.start(m0 + wm * (TMW * 32) + i * 32 + er, p.mg_F, p.mg_K, RPS);
                u32x4 ex[NPS];
                if constexpr (epi_reads_extra(EPI)) {
                    const bf16_t* src = EPI == EPI_BIAS_DROP_RES ? p.res : p.aux;
#pragma unroll
                    for (int ps = 0; ps < NPS; ++ps) {
                        int64_t grow = m0 + wm * (TMW * 32) + i * 32 + ps * RPS + er;
                        if constexpr (RAGGED) grow = grow < m_last ? grow : m_last;
                        ex[ps] = *reinterpret_cast<const u32x4*>(src + grow * p.N + col);
                    }
                }
#pragma unroll
                for (int j = 0; j < SW / 32; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        stg[crow(r, hh) * SLD + j * 32 + lq] = acc[i][jc * (SW / 32) + j][r];
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int ps = 0; ps < NPS; ++ps) {
                    const int rr = ps * RPS + er;
                    const int64_t grow = m0 + wm * (TMW * 32) + i * 32 + rr;
                    if constexpr (RAGGED) { if (grow > m_last) continue; }
                    const int64_t off = grow * p.N + col;
                    f32x4 v0 = *reinterpret_cast<const f32x4*>(stg + rr * SLD + ec);
                    f32x4 v1 = *reinterpret_cast<const f32x4*>(stg + rr * SLD + ec + 4);
                    if constexpr (STAT == X_LNFOLD) {
                        const float t = rm[ps] * rr_[ps];
                        v0 = v0 * rr_[ps] + (cv0 - sv0 * t); v1 = v1 * rr_[ps] + (cv1 - sv1 * t);
                    } else {
                        v0 += bv0; v1 += bv1;
                    }
                    float o8[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                    float dk[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
                    if constexpr (epi_drops(EPI)) {
                        if (epi_th) {
                            const uint64_t e0 = (uint64_t)(off + (RAGGED ? p.row0 * p.N : 0));
                            const f32x4 k0 = drop_keep4(p.epi_seed, e0, epi_th, epi_sc), k1 = drop_keep4(p.epi_seed, e0 + 4, epi_th, epi_sc);
                            dk[0] = k0.x; dk[1] = k0.y; dk[2] = k0.z; dk[3] = k0.w; dk[4] = k1.x; dk[5] = k1.y; dk[6] = k1.z; dk[7] = k1.w;
                        }
                    }
                    if constexpr (EPI == EPI_BIAS_DROP_RES) {
                        float rs[8];
                        unpack8(ex[ps], rs);
#pragma unroll
                        for (int e = 0; e < 8; ++e) o8[e] = rs[e] + o8[e] * dk[e];
                    } else if constexpr (EPI == EPI_BIAS_GELU_DROP) {
                        const u32x4 pre = pack8(o8);
                        *reinterpret_cast<u32x4*>(p.C2 + off) = pre;
                        float h[8];
                        unpack8(pre, h);                       // gelu on the bf16-rounded pre-activation that backward will see
#pragma unroll
                        for (int e = 0; e < 8; ++e) o8[e] = gelu_f(h[e]) * dk[e];
                    } else if constexpr (EPI == EPI_BIAS_GELU_DROP_G) {
                        float g8[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) gelu_fwd_grad(o8[e], dk[e], o8[e], g8[e]);
                        *reinterpret_cast<u32x4*>(p.C2 + off) = pack8(g8);
                    } else if constexpr (EPI == EPI_MUL_AUX) {
                        float h[8];
                        unpack8(ex[ps], h);
#pragma unroll
                        for (int e = 0; e < 8; ++e) o8[e] *= h[e];
                    } else if constexpr (EPI == EPI_GELU_BWD) {
                        float h[8];
                        unpack8(ex[ps], h);
#pragma unroll
                        for (int e = 0; e < 8; ++e) o8[e] = o8[e] * dk[e] * gelu_grad(h[e]);
                    }
                    const u32x4 outv = pack8(o8);
                    if constexpr (STAT == X_STAT || STAT == X_STAT_MERGE) {                  // statistics of the bf16 values the next LayerNorm reads
                        float q[8];
                        unpack8(outv, q);
                        float s1 = 0.f, s2 = 0.f;
#pragma unroll
                        for (int e = 0; e < 8; ++e) { s1 += q[e]; s2 += q[e] * q[e]; }
                        s1 = group_sum<LPR>(s1); s2 = group_sum<LPR>(s2);
                        const int lr = wm * (TMW * 32) + i * 32 + rr;
                        if ((lane % LPR) == 0) { atomicAdd(rowstat + 2 * lr, s1); atomicAdd(rowstat + 2 * lr + 1, s2); }
                        if constexpr (STAT == X_STAT_MERGE) {
                            *reinterpret_cast<u32x4*>(p.C + mw.off(p.N) + col) = outv;
                            mw.next();
                        } else {
                            *reinterpret_cast<u32x4*>(p.C + off) = outv;
                        }
                    } else {
                        *reinterpret_cast<u32x4*>(p.C + off) = outv;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
        if constexpr (STAT == X_STAT || STAT == X_STAT_MERGE) {
            __syncthreads();
            for (int q = tid; q < BM; q += C::THREADS) {
                int64_t mr = m0 + q;
                if constexpr (STAT == X_STAT_MERGE) { MergeWalk w; w.start(m0 + q, p.mg_F, p.mg_K, 1); mr = w.mrow(); }
                atomicAdd(p.stat_sum + mr, rowstat[2 * q]);
                atomicAdd(p.stat_sq + mr, rowstat[2 * q + 1]);
            }
        }
    };

    SlabIt ci{(int)blockIdx.x, 0, 0, 0, true};
    if (ci.t >= n_tiles) return;
    tile_origin(ci.t, ci.m0, ci.n0);
    SlabIt ii = ci, meta0 = ci, meta1 = ci;

    u32x4 ra0[PA], rw0[PW], ra1[PA], rw1[PW];
    float lm0[PA], lr0[PA], lm1[PA], lr1[PA];
    bool v0 = false, v1 = false;

    issue(ra0, rw0, lm0, lr0, ii); meta0 = ii; advance(ii);
    if (ii.valid) { issue(ra1, rw1, lm1, lr1, ii); meta1 = ii; v1 = true; advance(ii); }
    commit(ra0, rw0, lm0, lr0, meta0, 0);
    __syncthreads();

    while (true) {
        // ---- phase A: slab `ci` is in LDS buffer 0, slab ci+1 is in flight in register set 1
        if (ii.valid) { issue(ra0, rw0, lm0, lr0, ii); meta0 = ii; v0 = true; advance(ii); } else v0 = false;
        compute(0, ci.s == 0);
        if (v1) commit(ra1, rw1, lm1, lr1, meta1, 1);
        __syncthreads();
        if (ci.s == n_slab - 1) { epilogue(ci.m0, ci.n0, 0); __syncthreads(); }
        advance(ci);
        if (!ci.valid) break;
        // ---- phase B: slab `ci` is in LDS buffer 1, slab ci+1 is in flight in register set 0
        if (ii.valid) { issue(ra1, rw1, lm1, lr1, ii); meta1 = ii; v1 = true; advance(ii); } else v1 = false;
        compute(1, ci.s == 0);
        if (v0) commit(ra0, rw0, lm0, lr0, meta0, 0);
        __syncthreads();
        if (ci.s == n_slab - 1) { epilogue(ci.m0, ci.n0, 1); __syncthreads(); }
        advance(ci);
        if (!ci.valid) break;
    }
}

// ------------------------------------------------------------------ dW / db (bf16 operands)

constexpr int TMB = 32;              // rows of M per LDS stage (two k16 steps)
constexpr int LDW = 160;             // LDS row stride in bf16 (320 B): tr-read rows hit disjoint bank quarters

// 8 consecutive m (k index of the MFMA) for column (c0 + lane_in_group) -> one operand fragment
__device__ __forceinline__ bf16x8 tr_frag(const bf16_t* tile, int m_base, int c0, int lane) {
    const int il = lane & 15, q = il >> 2, pcol = (il & 3) * 4;
    const bf16_t* a0 = tile + (m_base + q) * LDW + c0 + pcol;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)a0);
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(a0 + 4 * LDW));
    bf16x8 f = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    return f;
}

template <int PRO, bool BLN, bool RAGGED = false>
__global__ __launch_bounds__(256, 3) void gemm_tn_bf16_k(TnArgsB p) {
    HWGAT_RESOLVE_SEED1(p);
    __shared__ __attribute__((aligned(16))) bf16_t sm[2 * 2 * TMB * LDW];    // [buf][A|B][32][160]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane & 31, hh = lane >> 5;
    const int wn = wave >> 1, wk = wave & 1;
    const int tiles_n = p.N / 128, tiles_k = p.K / 128, n_tiles = tiles_n * tiles_k;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int tile = j % n_tiles;
    const int split = (j / n_tiles) * 8 + xcd;
    if (split >= p.n_split) return;
    const int n0 = (tile / tiles_k) * 128, k0 = (tile % tiles_k) * 128;
    const int64_t r_begin = (int64_t)split * p.rows_per_split;
    const int64_t r_end = r_begin + p.rows_per_split < p.M ? r_begin + p.rows_per_split : p.M;
    if (r_begin >= r_end) return;
    const int n_it = RAGGED ? (int)((r_end - r_begin + TMB - 1) / TMB) : (int)((r_end - r_begin) / TMB);
    const int64_t m_last = p.M - 1;

    const int lrow = tid >> 4, lc8 = (tid & 15) * 8;           // rows lrow + 16*i, bf16 columns lc8..lc8+7
    const uint32_t pro_th = drop_thresh(p.pro_p);
    const float pro_sc = 1.0f / (1.0f - p.pro_p);
    u32x4 ra[2], rb[2];
    float bm[2], bs[2];
    float colsum[8], lg[8], lb[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { colsum[e] = 0.f; lg[e] = 1.f; lb[e] = 0.f; }
    if constexpr (BLN) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { lg[e] = p.gamma[k0 + lc8 + e]; lb[e] = p.beta[k0 + lc8 + e]; }
    }

    auto issue = [&](int it) {
        const int64_t r0 = r_begin + (int64_t)it * TMB + lrow;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int64_t row = r0 + 16 * i;
            if constexpr (RAGGED) row = row < m_last ? row : m_last;
            ra[i] = *reinterpret_cast<const u32x4*>(p.A + row * p.N + n0 + lc8);
            rb[i] = *reinterpret_cast<const u32x4*>(p.B + row * p.K + k0 + lc8);
            if constexpr (BLN) { bm[i] = p.mean[row]; bs[i] = p.rstd[row]; }
        }
    };
    auto commit = [&](int buf, int it) {
        bf16_t* As = sm + buf * (2 * TMB * LDW);
        bf16_t* Bs = As + TMB * LDW;
        const int64_t r0 = r_begin + (int64_t)it * TMB + lrow;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            u32x4 a = ra[i], b = rb[i];
            float av[8];
            unpack8(a, av);
            if constexpr (PRO == PRO_DROP) {
                const uint64_t e0 = (uint64_t)((RAGGED ? p.row0 : 0) + r0 + 16 * i) * p.N + n0 + lc8;
                const f32x4 k0v = drop_keep4(p.pro_seed, e0, pro_th, pro_sc);
                const f32x4 k1v = drop_keep4(p.pro_seed, e0 + 4, pro_th, pro_sc);
                av[0] *= k0v.x; av[1] *= k0v.y; av[2] *= k0v.z; av[3] *= k0v.w;
                av[4] *= k1v.x; av[5] *= k1v.y; av[6] *= k1v.z; av[7] *= k1v.w;
                a = pack8(av);
            }
            if constexpr (RAGGED) {
                if (r0 + 16 * i > m_last) {                      // rows past the end count as zero
                    a = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
                    for (int e = 0; e < 8; ++e) av[e] = 0.f;
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) colsum[e] += av[e];
            if constexpr (BLN) {
                float bv[8];
                unpack8(b, bv);
#pragma unroll
                for (int e = 0; e < 8; ++e) bv[e] = (bv[e] - bm[i]) * bs[i] * lg[e] + lb[e];
                b = pack8(bv);
            }
            *reinterpret_cast<u32x4*>(As + (lrow + 16 * i) * LDW + lc8) = a;
            *reinterpret_cast<u32x4*>(Bs + (lrow + 16 * i) * LDW + lc8) = b;
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
    const int csub = 16 * ((lane >> 4) & 1);                   // 16-lane group -> column half of a 32-wide fragment
    for (int it = 0; it < n_it; ++it) {
        const bool have_next = it + 1 < n_it;
        if (have_next) issue(it + 1);
        const bf16_t* As = sm + buf * (2 * TMB * LDW);
        const bf16_t* Bs = As + TMB * LDW;
#pragma unroll
        for (int s = 0; s < TMB / 16; ++s) {
            const int mb = 16 * s + 8 * hh;
            const bf16x8 a0 = tr_frag(As, mb, wn * 64 + csub, lane);
            const bf16x8 a1 = tr_frag(As, mb, wn * 64 + 32 + csub, lane);
            const bf16x8 b0 = tr_frag(Bs, mb, wk * 64 + csub, lane);
            const bf16x8 b1 = tr_frag(Bs, mb, wk * 64 + 32 + csub, lane);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (have_next) commit(buf ^ 1, it + 1);
        __syncthreads();
        buf ^= 1;
    }
    const bool det = p.det_dw != nullptr;                       // deterministic mode: see TnArgsB
    float* dwo = det ? p.det_dw + (int64_t)split * p.N * p.K : p.dW;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wn * 64 + i * 32 + crow(r, hh);
                const int k = k0 + wk * 64 + jj * 32 + lq;
                HWGAT_TN_ACC(det, dwo, (int64_t)n * p.K + k, acc[i][jj][r]);
            }
    if (p.db != nullptr && k0 == 0) {
        float* red = reinterpret_cast<float*>(sm);             // [16][128] partial column sums
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) red[lrow * 128 + lc8 + e] = colsum[e];
        __syncthreads();
        if (tid < 128) {
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) s += red[q * 128 + tid];
            HWGAT_TN_ACC(det, det ? p.det_db + (int64_t)split * p.N : p.db, n0 + tid, s);
        }
    }
}

template <int PRO, typename C, bool RAGGED = false>
int launch_nt_b(const NtArgsB& a, int epi, hipStream_t st, bool fold = false) {
    const int64_t tiles = ((a.M + C::BM - 1) / C::BM) * (a.N / C::BN);
    const int grid = (int)(tiles < C::SLOTS ? tiles : C::SLOTS);
    if (fold) {                                                 // PRO_LN_FOLD: plain loaders, row-affine epilogue
        if constexpr (PRO == PRO_NONE) {
            if (epi == EPI_BIAS) gemm_nt_bf16_k<PRO_NONE, EPI_BIAS, C, RAGGED, X_LNFOLD><<<grid, C::THREADS, 0, st>>>(a);
            else if (epi == EPI_BIAS_GELU_DROP) gemm_nt_bf16_k<PRO_NONE, EPI_BIAS_GELU_DROP, C, RAGGED, X_LNFOLD><<<grid, C::THREADS, 0, st>>>(a);
            else if (epi == EPI_BIAS_GELU_DROP_G) gemm_nt_bf16_k<PRO_NONE, EPI_BIAS_GELU_DROP_G, C, RAGGED, X_LNFOLD><<<grid, C::THREADS, 0, st>>>(a);
            else return HWGAT_EINVAL;
            HWGAT_LAUNCH_CHECK();
        }
        return HWGAT_EINVAL;
    }
    if (a.stat_sum != nullptr) {                                // validated by the caller: PRO_NONE, EPI_BIAS_DROP_RES, whole tiles
        if constexpr (PRO == PRO_NONE && !RAGGED) {
            if (a.mg_K > 0) gemm_nt_bf16_k<PRO_NONE, EPI_BIAS_DROP_RES, C, false, 2><<<grid, C::THREADS, 0, st>>>(a);
            else gemm_nt_bf16_k<PRO_NONE, EPI_BIAS_DROP_RES, C, false, 1><<<grid, C::THREADS, 0, st>>>(a);
            HWGAT_LAUNCH_CHECK();
        }
        return HWGAT_EINVAL;
    }
    switch (epi) {
        case EPI_BIAS: gemm_nt_bf16_k<PRO, EPI_BIAS, C, RAGGED><<<grid, C::THREADS, 0, st>>>(a); break;
        case EPI_BIAS_DROP_RES: gemm_nt_bf16_k<PRO, EPI_BIAS_DROP_RES, C, RAGGED><<<grid, C::THREADS, 0, st>>>(a); break;
        case EPI_BIAS_GELU_DROP: gemm_nt_bf16_k<PRO, EPI_BIAS_GELU_DROP, C, RAGGED><<<grid, C::THREADS, 0, st>>>(a); break;
        case EPI_GELU_BWD: gemm_nt_bf16_k<PRO, EPI_GELU_BWD, C, RAGGED><<<grid, C::THREADS, 0, st>>>(a); break;
        case EPI_BIAS_GELU_DROP_G: gemm_nt_bf16_k<PRO, EPI_BIAS_GELU_DROP_G, C, RAGGED><<<grid, C::THREADS, 0, st>>>(a); break;
        case EPI_MUL_AUX: gemm_nt_bf16_k<PRO, EPI_MUL_AUX, C, RAGGED><<<grid, C::THREADS, 0, st>>>(a); break;
        case EPI_NONE: gemm_nt_bf16_k<PRO, EPI_NONE, C, RAGGED><<<grid, C::THREADS, 0, st>>>(a); break;
        default: return HWGAT_EINVAL;
    }
    HWGAT_LAUNCH_CHECK();
}

// the same launch restricted to rows [r0, r0 + rows) of every M-indexed operand
NtArgsB nt_rows_b(NtArgsB a, int64_t r0, int64_t rows) {
    a.A += r0 * a.K;
    a.C += r0 * a.N;
    if (a.C2) a.C2 += r0 * a.N;
    if (a.res) a.res += r0 * a.N;
    if (a.aux) a.aux += r0 * a.N;
    if (a.mean) { a.mean += r0; a.rstd += r0; }
    a.M = rows;
    a.row0 = r0;
    return a;
}

}  // namespace

extern "C" int hwgat_linear_nt_bf16_ex(const void* A, const void* W, const float* bias, void* C, int64_t M, int N,
                                       int K, int pro, const float* mean, const float* rstd, const float* gamma,
                                       const float* beta, uint32_t pro_seed, float pro_p, int epi, const void* res,
                                       void* C2, const void* aux, uint32_t epi_seed, float epi_p, float* stat_sum,
                                       float* stat_sq, int merge_F, int merge_K, const uint32_t* seed_base, void* stream) {
    if (!A || !W || !C || M <= 0 || N <= 0 || K <= 0) return HWGAT_EINVAL;
    if (N % 128 || K % 64 || ((M + 127) / 128) * (int64_t)(N / 128) > 0x7fffffff) return HWGAT_ESHAPE;   // any M
    if ((pro == PRO_LN || pro == PRO_LN_FOLD) && (!mean || !rstd || !gamma || !beta)) return HWGAT_EINVAL;
    if (pro == PRO_LN_FOLD) {                                      // gamma = s[N], beta = c[N] of hwgat_ln_fold; whole tiles
        if (epi != EPI_BIAS && epi != EPI_BIAS_GELU_DROP && epi != EPI_BIAS_GELU_DROP_G) return HWGAT_EINVAL;
        if (M % 128) return HWGAT_ESHAPE;
    }
    if (epi == EPI_BIAS_DROP_RES && !res) return HWGAT_EINVAL;
    if ((epi == EPI_BIAS_GELU_DROP || epi == EPI_BIAS_GELU_DROP_G) && !C2) return HWGAT_EINVAL;
    if ((epi == EPI_GELU_BWD || epi == EPI_MUL_AUX) && !aux) return HWGAT_EINVAL;
    if (pro_p < 0.f || pro_p >= 1.f || epi_p < 0.f || epi_p >= 1.f) return HWGAT_EINVAL;
    if (pro == PRO_DROP && pro_p == 0.f) pro = PRO_NONE;          // eval mode: no mask to hash
    const bool stat = stat_sum != nullptr || stat_sq != nullptr || merge_K > 0;
    if (stat) {
        if (!stat_sum || !stat_sq || pro != PRO_NONE || epi != EPI_BIAS_DROP_RES) return HWGAT_EINVAL;
        if (M % 256) return HWGAT_ESHAPE;
        if (merge_K > 0 && (merge_F <= 0 || (merge_F & 1) || M % ((int64_t)merge_F * merge_K))) return HWGAT_EINVAL;
    }
    NtArgsB a{(const bf16_t*)A, (const bf16_t*)W, bias, (bf16_t*)C, (bf16_t*)C2, (const bf16_t*)res,
              (const bf16_t*)aux, mean, rstd, gamma, beta, M, N, K, pro_seed, epi_seed, pro_p, epi_p, 0, stat_sum, stat_sq, merge_K > 0 ? merge_F : 0, merge_K > 0 ? merge_K : 0};
    a.seed_base = seed_base;
    hipStream_t st = (hipStream_t)stream;
    const int64_t m_bulk = M / 128 * 128;                       // ragged token count: bulk launch + RAGGED tail launch
    if (m_bulk != M) {
        if (m_bulk) {
            const int rc = hwgat_linear_nt_bf16(A, W, bias, C, m_bulk, N, K, pro, mean, rstd, gamma, beta, pro_seed, pro_p,
                                                epi, res, C2, aux, epi_seed, epi_p, seed_base, stream);
            if (rc) return rc;
        }
        const NtArgsB t = nt_rows_b(a, m_bulk, M - m_bulk);
        switch (pro) {
            case PRO_NONE: return launch_nt_b<PRO_NONE, NtB64, true>(t, epi, st);
            case PRO_LN: return launch_nt_b<PRO_LN, NtB64, true>(t, epi, st);
            case PRO_DROP: return launch_nt_b<PRO_DROP, NtB64, true>(t, epi, st);
            default: return HWGAT_EINVAL;
        }
    }
    // outputs whose width is a multiple of 256: the 256x256 one-wave-per-SIMD kernel (gemm_bf16_nt256.hip: half the
    // L2 -> LDS stream of the 128x128 tile) over the 256-aligned rows; HWGAT_NT_KERNEL=old keeps the 128x128 kernel
    static const bool nt_old = [] { const char* e = lab_env("HWGAT_NT_KERNEL"); return e && e[0] == 'o'; }();
    static const int nt256_min_k = [] { const char* e = lab_env("HWGAT_NT256_MINK"); return e ? atoi(e) : 128; }();
    // (fewer than 128 such tiles: the 128 x 128 kernel spreads a serving-size launch over four times as many CUs)
    if (!nt_old && N % 256 == 0 && K >= nt256_min_k && M >= 256 &&
        ((M / 256) * (N / 256) >= 128 || a.stat_sum != nullptr)) {     // (the row statistics of the 256-wide kernels are the order-fixed ones: eval determinism)
        const int64_t m256 = M / 256 * 256;
        NtArgsB b = a;
        b.M = m256;
        // ... on the eight-wave LDS-DMA kernel (gemm_bf16_nt8w.hip) where it applies; HWGAT_NT8W=0 (lab builds) keeps
        // the one-wave-per-SIMD kernel for A/B runs
        static const bool no8w = [] { const char* e = lab_env("HWGAT_NT8W"); return e && e[0] == '0'; }();
        const int rc = (!no8w && hwgat_nt8w_bf16_takes(b, pro, epi)) ? hwgat_launch_nt8w_bf16(b, pro, epi, st)
                                                                      : hwgat_launch_nt256_bf16(b, pro, epi, st);
        if (rc || m256 == M) return rc;
        const NtArgsB t = nt_rows_b(a, m256, M - m256);   // 128 rows left: RAGGED instantiation (global row index in the dropout hash)
        switch (pro) {
            case PRO_NONE: return launch_nt_b<PRO_NONE, NtB64, true>(t, epi, st);
            case PRO_LN_FOLD: return launch_nt_b<PRO_NONE, NtB64, true>(t, epi, st, true);
            case PRO_LN: return launch_nt_b<PRO_LN, NtB64, true>(t, epi, st);
            case PRO_DROP: return launch_nt_b<PRO_DROP, NtB64, true>(t, epi, st);
            default: return HWGAT_EINVAL;
        }
    }
#define NTB_GO(P) return launch_nt_b<P, NtB64>(a, epi, st)
    switch (pro) {
        case PRO_NONE: NTB_GO(PRO_NONE);
        case PRO_LN_FOLD: return launch_nt_b<PRO_NONE, NtB64>(a, epi, st, true);
        case PRO_LN: NTB_GO(PRO_LN);
        case PRO_DROP: NTB_GO(PRO_DROP);
        default: return HWGAT_EINVAL;
    }
#undef NTB_GO
}

extern "C" int hwgat_linear_nt_bf16(const void* A, const void* W, const float* bias, void* C, int64_t M, int N,
                                    int K, int pro, const float* mean, const float* rstd, const float* gamma,
                                    const float* beta, uint32_t pro_seed, float pro_p, int epi, const void* res,
                                    void* C2, const void* aux, uint32_t epi_seed, float epi_p,
                                    const uint32_t* seed_base, void* stream) {
    return hwgat_linear_nt_bf16_ex(A, W, bias, C, M, N, K, pro, mean, rstd, gamma, beta, pro_seed, pro_p, epi, res, C2, aux,
                                   epi_seed, epi_p, nullptr, nullptr, 0, 0, seed_base, stream);
}

extern "C" int64_t hwgat_linear_tn_bf16_ws_bytes(int64_t M, int N, int K) { return 4 * hwgat_tn8w_bf16_ws_floats(M, N, K); }

// plain operands with a caller-owned workspace: the M-split partial tiles are combined by a fixed-order reduction instead of
// global atomics (faster, and dW is bit-reproducible); shapes the slab kernel does not take, or a workspace that is too
// small / NULL, fall back to hwgat_linear_tn_bf16 (atomics).  db always accumulates with atomics (N floats per block).
extern "C" int hwgat_linear_tn_bf16_ws(const void* A, const void* B, float* dW, float* db, int64_t M, int N, int K,
                                       float* ws, int64_t ws_bytes, void* stream) {
    if (!A || !B || !dW || M <= 0 || N <= 0 || K <= 0) return HWGAT_EINVAL;
    const int64_t need = hwgat_linear_tn_bf16_ws_bytes(M, N, K);
    if (need == 0 || !ws || ws_bytes < need)
        return hwgat_linear_tn_bf16(A, B, dW, db, M, N, K, 0, 0.f, nullptr, nullptr, nullptr, nullptr, nullptr, stream);
    TnArgsB a{(const bf16_t*)A, (const bf16_t*)B, dW, db, nullptr, nullptr, nullptr, nullptr, M, N, K, 1, M, 0, 0.f, 0};
    return hwgat_launch_tn8w_bf16(a, (hipStream_t)stream, ws);
}

static int tn_bf16_impl(const void* A, const void* B, float* dW, float* db, int64_t M, int N, int K,
                        uint32_t pro_seed, float pro_p, const float* mean, const float* rstd,
                        const float* gamma, const float* beta, const uint32_t* seed_base, DetWs det, void* stream) {
    if (!A || !B || !dW || M <= 0 || N <= 0 || K <= 0) return HWGAT_EINVAL;
    if (mean && (!rstd || !gamma || !beta)) return HWGAT_EINVAL;
    if (N % 128 || K % 128) return HWGAT_ESHAPE;                 // any M
    if (pro_p < 0.f || pro_p >= 1.f) return HWGAT_EINVAL;
    const int64_t m_bulk = M / TMB * TMB;
    if (m_bulk != M) {                                          // bulk launch + one RAGGED stage for the last M % 32 rows
        if (det.dw) return HWGAT_ESHAPE;                        // deterministic mode: whole 32-row stages only (one launch, one reduction)
        if (m_bulk) {
            const int rc = hwgat_linear_tn_bf16(A, B, dW, db, m_bulk, N, K, pro_seed, pro_p, mean, rstd, gamma, beta, seed_base, stream);
            if (rc) return rc;
        }
        const int64_t rows_t = M - m_bulk;
        TnArgsB t{(const bf16_t*)A + m_bulk * N, (const bf16_t*)B + m_bulk * K, dW, db, mean ? mean + m_bulk : nullptr,
                  mean ? rstd + m_bulk : nullptr, gamma, beta, rows_t, N, K, 1, TMB, pro_seed, pro_p, m_bulk};
        t.seed_base = seed_base;
        const int grid_t = 8 * (N / 128) * (K / 128);
        hipStream_t stt = (hipStream_t)stream;
        if (pro_p > 0.f) {
            if (mean) gemm_tn_bf16_k<PRO_DROP, true, true><<<grid_t, 256, 0, stt>>>(t);
            else gemm_tn_bf16_k<PRO_DROP, false, true><<<grid_t, 256, 0, stt>>>(t);
        } else {
            if (mean) gemm_tn_bf16_k<PRO_NONE, true, true><<<grid_t, 256, 0, stt>>>(t);
            else gemm_tn_bf16_k<PRO_NONE, false, true><<<grid_t, 256, 0, stt>>>(t);
        }
        HWGAT_LAUNCH_CHECK();
    }
    // dW at least 256 x 256: the 128x128-wave-tile kernel (gemm_bf16_tn256.hip); HWGAT_TN_KERNEL=old keeps this file's
    static const bool tn_old = [] { const char* e = lab_env("HWGAT_TN_KERNEL"); return e && e[0] == 'o'; }();
    // ... where its tiles fill the 256 CUs in whole rounds of equal blocks (tile count a divisor of 256: 1, 2, 4, 8 ...);
    // 3 or 12 tiles (the qkv weight) need three rounds of short M slices and lose to the 128x128 kernel:
    // stage 2 dWqkv 492 vs 454 us, stage 1 349 vs 272 (same box, tools/tn_lab.py)
    // plain operands, whole 256x256 tiles: the eight-wave LDS-DMA kernel (gemm_bf16_tn8w.hip); HWGAT_TN8W=0 (lab builds)
    // keeps the kernels below for A/B runs
    static const bool no8w = [] { const char* e = lab_env("HWGAT_TN8W"); return e && e[0] == '0'; }();
    if (!no8w && hwgat_tn8w_bf16_takes(M, N, K, pro_p, mean)) {
        TnArgsB a{(const bf16_t*)A, (const bf16_t*)B, dW, db, nullptr, nullptr, nullptr, nullptr, M, N, K, 1, M, pro_seed, pro_p, 0};
        a.det_dw = det.dw; a.det_db = det.db; a.det_cap = det.cap;
        return hwgat_launch_tn8w_bf16(a, (hipStream_t)stream);
    }
    const int t256 = (N / 256) * (K / 256);
    if (!tn_old && N % 256 == 0 && K % 256 == 0 && M % 32 == 0 && 256 % t256 == 0 && !(pro_p > 0.f && mean)) {
        TnArgsB a{(const bf16_t*)A, (const bf16_t*)B, dW, db, mean, rstd, gamma, beta, M, N, K, 1, M, pro_seed, pro_p, 0};
        a.seed_base = seed_base;
        a.det_dw = det.dw; a.det_db = det.db; a.det_cap = det.cap;
        return hwgat_launch_tn256_bf16(a, (hipStream_t)stream);
    }
    const int n_tiles = (N / 128) * (K / 128);
    auto gcd = [](int a, int b) { while (b) { int t = a % b; a = b; b = t; } return a; };
    // three 128x128 blocks per CU (40 KB of LDS, <= 168 registers), ONE round: every block ends in 64 KB of float atomics, so
    // fewer, longer M slices are cheaper (tools/tnb_sweep.sh, stage 0 of config 3 per step: 512 blocks x 2 rounds 1.20 ms,
    // 512 x 1 1.13, 768 x 1 1.09, 768 x 2 1.24, 1024 x 1 1.26)
    static const int slots = [] { const char* e = lab_env("HWGAT_TNB_SLOTS"); return e ? atoi(e) : 768; }();      // blocks of one round
    static const int min_rounds = [] { const char* e = lab_env("HWGAT_TNB_ROUNDS"); return e ? atoi(e) : 1; }();
    const int r_min = n_tiles / gcd(n_tiles, slots);
    int r = r_min;
    while (r < min_rounds) r += r_min;
    int64_t want = (int64_t)slots * r / n_tiles;
    const int64_t max_split = M / (TMB * 16) > 0 ? M / (TMB * 16) : 1;
    if (want > max_split) want = max_split;
    if (want < 1) want = 1;
    int64_t rows = (M + want - 1) / want;
    rows = (rows + TMB - 1) / TMB * TMB;
    const int n_split = (int)((M + rows - 1) / rows);
    TnArgsB a{(const bf16_t*)A, (const bf16_t*)B, dW, db, mean, rstd, gamma, beta, M, N, K, n_split, rows,
              pro_seed, pro_p, 0};
    a.seed_base = seed_base;
    if (det.dw) {
        if (n_split > det.cap) return HWGAT_ESHAPE;
        a.det_dw = det.dw; a.det_db = det.db; a.det_cap = det.cap;
    }
    const int grid = ((n_split + 7) / 8) * 8 * n_tiles;
    hipStream_t st = (hipStream_t)stream;
    if (pro_p > 0.f) {
        if (mean) gemm_tn_bf16_k<PRO_DROP, true><<<grid, 256, 0, st>>>(a);
        else gemm_tn_bf16_k<PRO_DROP, false><<<grid, 256, 0, st>>>(a);
    } else {
        if (mean) gemm_tn_bf16_k<PRO_NONE, true><<<grid, 256, 0, st>>>(a);
        else gemm_tn_bf16_k<PRO_NONE, false><<<grid, 256, 0, st>>>(a);
    }
    HWGAT_LAUNCH_CHECK();
}

extern "C" int hwgat_linear_tn_bf16(const void* A, const void* B, float* dW, float* db, int64_t M, int N, int K,
                                    uint32_t pro_seed, float pro_p, const float* mean, const float* rstd,
                                    const float* gamma, const float* beta, const uint32_t* seed_base, void* stream) {
    return tn_bf16_impl(A, B, dW, db, M, N, K, pro_seed, pro_p, mean, rstd, gamma, beta, seed_base, DetWs{nullptr, nullptr, 0}, stream);
}

// Deterministic form: the same kernels, but every block stores its partial dW tile / bias gradient into its M split's
// image of the caller's ZERO-FILLED workspace and one fixed-order pass adds the images: run-to-run identical bits.
// ws_bytes >= hwgat_linear_tn_det_bytes(M, N, K); M % 32 == 0.
extern "C" int hwgat_linear_tn_bf16_det(const void* A, const void* B, float* dW, float* db, int64_t M, int N, int K,
                                        uint32_t pro_seed, float pro_p, const float* mean, const float* rstd,
                                        const float* gamma, const float* beta, const uint32_t* seed_base, float* ws,
                                        int64_t ws_bytes, void* stream) {
    if (!ws || N <= 0 || K <= 0) return HWGAT_EINVAL;
    const int64_t per = (int64_t)N * K + N;
    const int64_t cap = ws_bytes / 4 / per;
    if (cap < 1) return HWGAT_ESHAPE;
    const DetWs det{ws, ws + cap * (int64_t)N * K, (int)(cap > 0x7fffffff ? 0x7fffffff : cap)};
    int rc = tn_bf16_impl(A, B, dW, db, M, N, K, pro_seed, pro_p, mean, rstd, gamma, beta, seed_base, det, stream);
    if (rc) return rc;
    rc = hwgat_tn_det_reduce(det.dw, dW, det.cap, (int64_t)N * K, (int64_t)N * K, (hipStream_t)stream);
    if (rc || !db) return rc;
    return hwgat_tn_det_reduce(det.db, db, det.cap, N, N, (hipStream_t)stream);
}
