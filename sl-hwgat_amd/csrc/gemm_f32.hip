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
#include <stdlib.h>
#include <type_traits>
#include "common.h"
#include "fused_ops.h"
#include "gemm_f32.h"

namespace {

constexpr int BK_MIN = 16;                                      // K must be a multiple of the slab depth

// Tiling configuration: WM x WN waves, each owning TM x TN MFMA tiles of 32x32.
//   NtSmall  2x2 waves x (2x2) tiles = 128x128 block, 256 threads, 2 blocks / CU
//   NtBig    4x2 waves x (2x4) tiles = 256x256 block, 512 threads, 1 block / CU (still 2 waves / SIMD):
//            twice the MFMAs per barrier, half the global->LDS bytes per flop, 25 % fewer LDS
//            fragment reads per MFMA.  Needs M % 256 == N % 256 == 0.
using NtSmall = TileCfg<2, 2, 2, 2>;
using NtBig = TileCfg<4, 2, 2, 4, 32, 1>;
using NtK16 = TileCfg<2, 2, 2, 2, 16, 3>;                      // 41 KB LDS -> 3 blocks / CU

// RAGGED: the launch covers the last M % 128 rows of a token count that is not a multiple of the tile
// (HGATE: M = B*F*29).  It is a separate instantiation so that the bulk launch stays exactly the code
// measured in DESIGN.md: loads clamp to the last row, stores are guarded, and dropout masks are hashed
// with the global row index (p.row0 = first row of this launch in the full matrix).
// STAT (EPI_BIAS_DROP_RES only): the epilogue also produces the LayerNorm statistics of its OUTPUT rows (sum and sum
// of squares, reduced per tile in LDS, then one coalesced run of global atomics per tile) and can store in the
// TemporalMerging layout -- see NtArgs.  It is a separate instantiation: the plain launches stay the measured code.
// (STAT: 0 = plain, 1 = + row statistics, 2 = + row statistics and the merged store)
template <int PRO, int EPI, typename C, bool RAGGED = false, int STAT = 0>
__global__ __launch_bounds__(C::THREADS, C::OCC * C::THREADS / 256) void gemm_nt_k(NtArgs p) {
    HWGAT_RESOLVE_SEEDS2(p);
    constexpr int BM = C::BM, BN = C::BN, TMW = C::TMW, TNW = C::TNW, PA = C::PA, PW = C::PW, RPP = C::RPP;
    constexpr int BK = C::BK, LDT = C::LDT;
    __shared__ __attribute__((aligned(16))) float sm[2 * (BM + BN) * LDT];    // [buf][A rows | W rows][36]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane & 31, hh = lane >> 5;
    const int wm = wave / C::WN, wn = wave % C::WN;
    const int tiles_n = p.N / BN;
    const int row_blocks = RAGGED ? (int)((p.M + BM - 1) / BM) : (int)(p.M / BM);
    const int n_tiles = row_blocks * tiles_n;
    const int64_t m_last = p.M - 1;
    const int n_slab = p.K / BK;
    const int lrow = tid / C::TPR, lc4 = (tid % C::TPR) * 4;   // this thread stages rows lrow + RPP*i, floats lc4..lc4+3
    const uint32_t pro_th = drop_thresh(p.pro_p);
    const float pro_sc = 1.0f / (1.0f - p.pro_p);

    f32x4 ra[PA], rw[PW];
    float ln_mean[PA], ln_rstd[PA];
    f32x4 ln_g, ln_b;                                          // gamma/beta of the staged slab, prefetched with it

    // XCD-aware tile order: blocks b, b+8, b+16, ... share an XCD (and its L2); give them the
    // n-tiles of ONE row block of A, so A streams from HBM once and is re-read from L2.
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
        for (int i = 0; i < PA; ++i) {
            int64_t row = m0 + lrow + RPP * i;
            if constexpr (RAGGED) row = row < m_last ? row : m_last;
            ra[i] = *reinterpret_cast<const f32x4*>(p.A + row * p.K + k0);
            if constexpr (PRO == PRO_LN) {
                if (new_tile) { ln_mean[i] = p.mean[row]; ln_rstd[i] = p.rstd[row]; }
            }
        }
#pragma unroll
        for (int i = 0; i < PW; ++i)
            rw[i] = *reinterpret_cast<const f32x4*>(p.W + (int64_t)(n0 + lrow + RPP * i) * p.K + k0);
        if constexpr (PRO == PRO_LN) {
            ln_g = *reinterpret_cast<const f32x4*>(p.gamma + k0);
            ln_b = *reinterpret_cast<const f32x4*>(p.beta + k0);
        }
    };
    auto commit = [&](int buf, int64_t m0, int slab) {
        float* As = sm + buf * ((BM + BN) * LDT);
        float* Ws = As + BM * LDT;
        const int k0 = slab * BK + lc4;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            f32x4 a = ra[i];
            if constexpr (PRO == PRO_LN) {
                a = (a - ln_mean[i]) * (ln_rstd[i] * ln_g) + ln_b;
            } else if constexpr (PRO == PRO_DROP) {
                if (pro_th) a *= drop_keep4(p.pro_seed, (uint64_t)((RAGGED ? p.row0 : 0) + m0 + lrow + RPP * i) * p.K + k0, pro_th, pro_sc);
            }
            *reinterpret_cast<f32x4*>(As + (lrow + RPP * i) * LDT + lc4) = a;
        }
#pragma unroll
        for (int i = 0; i < PW; ++i) *reinterpret_cast<f32x4*>(Ws + (lrow + RPP * i) * LDT + lc4) = rw[i];
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
        f32x16 acc[TMW][TNW];
#pragma unroll
        for (int i = 0; i < TMW; ++i)
#pragma unroll
            for (int j = 0; j < TNW; ++j)
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

            const float* As = sm + buf * ((BM + BN) * LDT);
            const float* Ws = As + BM * LDT;
            const float* ap = As + (wm * (TMW * 32) + lq) * LDT + 4 * hh;
            const float* wp = Ws + (wn * (TNW * 32) + lq) * LDT + 4 * hh;
#pragma unroll
            for (int kk = 0; kk < BK / 8; ++kk) {
                f32x4 af[TMW], bf[TNW];
#pragma unroll
                for (int i = 0; i < TMW; ++i) af[i] = *reinterpret_cast<const f32x4*>(ap + i * 32 * LDT + 8 * kk);
#pragma unroll
                for (int j = 0; j < TNW; ++j) bf[j] = *reinterpret_cast<const f32x4*>(wp + j * 32 * LDT + 8 * kk);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < TMW; ++i)
#pragma unroll
                        for (int j = 0; j < TNW; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[j][e], acc[i][j], 0, 0, 0);
            }
            if (have_next) commit(buf ^ 1, mn, s_next);
            __syncthreads();
            buf ^= 1;
        }

        // ---- epilogue.  acc: lane (n = lq, hh), reg r -> C[m = crow(r,hh)][n].  Each wave parks
        // one 32x64 piece of its tile in the LDS buffer that is idle now (buf^1), then walks it
        // row-wise so that every global access is a 16 B/lane, 256 B/row-segment vector.
        {
            const uint32_t epi_th = drop_thresh(p.epi_p);
            const float epi_sc = 1.0f / (1.0f - p.epi_p);
            // staging piece: 32 rows x SW columns per wave (64 wide when the idle buffer has room)
            constexpr int SW = ((BM + BN) * LDT >= (C::THREADS / 64) * 32 * 68) ? 64 : 32;
            constexpr int SLD = SW + 4, LPR = SW / 4, RPS = 64 / LPR, NPS = 32 / RPS;
            float* stg = sm + (buf ^ 1) * ((BM + BN) * LDT) + wave * (32 * SLD);
            const int er = lane / LPR, ec = (lane % LPR) * 4;
            float* rowstat = sm + (buf ^ 1) * ((BM + BN) * LDT) + (C::THREADS / 64) * (32 * SLD);   // [BM][2] behind the staging
            if constexpr (STAT == X_STAT || STAT == X_STAT_MERGE) {
                static_assert((C::THREADS / 64) * 32 * SLD + 2 * BM <= (BM + BN) * LDT, "no room for the row statistics");
                for (int q = tid; q < 2 * BM; q += C::THREADS) rowstat[q] = 0.f;
                __syncthreads();
            }
#pragma unroll
            for (int jc = 0; jc < TNW * 32 / SW; ++jc) {
                const int col = n0 + wn * (TNW * 32) + jc * SW + ec;
                f32x4 bv = {0.f, 0.f, 0.f, 0.f};
                if constexpr (epi_has_bias(EPI))
                    if (p.bias) bv = *reinterpret_cast<const f32x4*>(p.bias + col);
                f32x4 sv = bv, cv = bv;                         // X_LNFOLD: s_n and c_n of the lane's columns
                if constexpr (STAT == X_LNFOLD) {
                    sv = *reinterpret_cast<const f32x4*>(p.gamma + col);
                    cv = *reinterpret_cast<const f32x4*>(p.beta + col);
                }
#pragma unroll
                for (int i = 0; i < TMW; ++i) {
                    float rr_[NPS], rm[NPS];                    // X_LNFOLD: rstd and mean of the piece's rows
                    if constexpr (STAT == X_LNFOLD) {
#pragma unroll
                        for (int ps = 0; ps < NPS; ++ps) {
                            int64_t grow = m0 + wm * (TMW * 32) + i * 32 + ps * RPS + er;
                            if constexpr (RAGGED) grow = grow < m_last ? grow : m_last;
                            rr_[ps] = p.rstd[grow]; rm[ps] = p.mean[grow];
                        }
                    }
                    // the piece's residual / pre-activation operand: ALL its passes are requested before the piece is
                    // parked, so NPS loads per lane are in flight together (they used to be issued pass by pass, two
                    // at a time, each paying an HBM round trip: the GELU-backward launch, whose largest stream this
                    // is, ran at 99 TFLOP/s against 130 for the plain product)
                    MergeWalk mw;
                    if constexpr (STAT == X_STAT_MERGE) mw.start(m0 + wm * (TMW * 32) + i * 32 + er, p.mg_F, p.mg_K, RPS);
                    f32x4 ex[NPS];
                    if constexpr (epi_reads_extra(EPI)) {
                        const float* src = EPI == EPI_BIAS_DROP_RES ? p.res : p.aux;
#pragma unroll
                        for (int ps = 0; ps < NPS; ++ps) {
                            int64_t grow = m0 + wm * (TMW * 32) + i * 32 + ps * RPS + er;
                            if constexpr (RAGGED) grow = grow < m_last ? grow : m_last;
                            ex[ps] = *reinterpret_cast<const f32x4*>(src + grow * p.N + col);
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
                        const int64_t off = grow * p.N + col;       // natural layout: residual, dropout hash
                        f32x4 v = *reinterpret_cast<const f32x4*>(stg + rr * SLD + ec);
                        if constexpr (STAT == X_LNFOLD) v = v * rr_[ps] + (cv - sv * (rm[ps] * rr_[ps]));
                        else v += bv;
                        f32x4 dk = {1.f, 1.f, 1.f, 1.f};
                        if constexpr (epi_drops(EPI)) {
                            if (epi_th) dk = drop_keep4(p.epi_seed, (uint64_t)(off + (RAGGED ? p.row0 * p.N : 0)), epi_th, epi_sc);
                        }
                        if constexpr (EPI == EPI_BIAS_DROP_RES) {
                            v = ex[ps] + v * dk;
                        } else if constexpr (EPI == EPI_BIAS_GELU_DROP) {
                            *reinterpret_cast<f32x4*>(p.C2 + off) = v;
                            v.x = gelu_f(v.x) * dk.x; v.y = gelu_f(v.y) * dk.y;
                            v.z = gelu_f(v.z) * dk.z; v.w = gelu_f(v.w) * dk.w;
                        } else if constexpr (EPI == EPI_BIAS_GELU_DROP_G) {
                            float u4[4] = {v.x, v.y, v.z, v.w}, g4[4];
                            const float k4[4] = {dk.x, dk.y, dk.z, dk.w};
#pragma unroll
                            for (int e = 0; e < 4; ++e) gelu_fwd_grad(u4[e], k4[e], u4[e], g4[e]);
                            v = f32x4{u4[0], u4[1], u4[2], u4[3]};
                            const f32x4 gq = {g4[0], g4[1], g4[2], g4[3]};
                            *reinterpret_cast<f32x4*>(p.C2 + off) = gq;
                        } else if constexpr (EPI == EPI_MUL_AUX) {
                            v *= ex[ps];
                        } else if constexpr (EPI == EPI_GELU_BWD) {
                            const f32x4 h = ex[ps];
                            v.x *= dk.x * gelu_grad(h.x); v.y *= dk.y * gelu_grad(h.y);
                            v.z *= dk.z * gelu_grad(h.z); v.w *= dk.w * gelu_grad(h.w);
                        }
                        if constexpr (STAT == X_STAT || STAT == X_STAT_MERGE) {
                            float s1 = (v.x + v.y) + (v.z + v.w), s2 = (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
                            s1 = group_sum<LPR>(s1); s2 = group_sum<LPR>(s2);
                            if ((lane % LPR) == 0) {
                                const int lr = wm * (TMW * 32) + i * 32 + rr;
                                atomicAdd(rowstat + 2 * lr, s1);
                                atomicAdd(rowstat + 2 * lr + 1, s2);
                            }
                            if constexpr (STAT == X_STAT_MERGE) {
                                *reinterpret_cast<f32x4*>(p.C + mw.off(p.N) + col) = v;
                                mw.next();
                            } else {
                                *reinterpret_cast<f32x4*>(p.C + off) = v;
                            }
                        } else {
                            *reinterpret_cast<f32x4*>(p.C + off) = v;
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
            }
            if constexpr (STAT == X_STAT || STAT == X_STAT_MERGE) {
                __syncthreads();
                for (int q = tid; q < BM; q += C::THREADS) {            // consecutive rows -> coalesced global atomics
                    int64_t mr = m0 + q;
                    if constexpr (STAT == X_STAT_MERGE) { MergeWalk w; w.start(m0 + q, p.mg_F, p.mg_K, 1); mr = w.mrow(); }
                    atomicAdd(p.stat_sum + mr, rowstat[2 * q]);
                    atomicAdd(p.stat_sq + mr, rowstat[2 * q + 1]);
                }
            }
        }
        __syncthreads();            // staging lives in buf^1, which the next tile's first commit overwrites
        t += gridDim.x;
        if (t >= n_tiles) break;
        m0 = mn; n0 = nn;
    }
}

// ------------------------------------------------------------------ dW / db
// rows of M per LDS stage = C::BK (32, or 16 for the 3-blocks-per-CU configuration)

// dW tile configurations (output tile BT x BT of dW, both operands [32][BT] per stage):
//   TnSmall  2x2 waves x (2x2) tiles = 128x128, 256 threads, 2 blocks / CU
//   TnBig    4x2 waves x (2x4) tiles = 256x256, 512 threads, 1 block / CU (N % 256 == K % 256 == 0)
using TnSmall = TileCfg<2, 2, 2, 2>;
using TnBig = TileCfg<4, 2, 2, 4, 32, 1>;
using TnK16 = TileCfg<2, 2, 2, 2, 16, 3>;                      // 32 KB LDS -> 3 blocks / CU

// MF16 selects v_mfma_f32_16x16x4_f32 (the form the vendor library uses: same flop rate, half the
// accumulator-register traffic per flop) instead of v_mfma_f32_32x32x2_f32.  Measured on MI355X:
// +2..4 % at K <= 256, -2 % at K = 512 -- not the source of the library's 0.91 vs our 0.82 MFMA
// utilisation; kept behind HWGAT_GEMM_TILE=m for A/B runs.
// RAGGED: launch over the last M % 32 rows (see gemm_nt_k): one partial stage, rows >= M count as zero.
template <int PRO, bool BLN, typename C, bool MF16 = false, bool RAGGED = false>
__global__ __launch_bounds__(C::THREADS, C::OCC * C::THREADS / 256) void gemm_tn_k(TnArgs p) {
    HWGAT_RESOLVE_SEED1(p);
    constexpr int TM = C::BK;
    constexpr int BT = C::BM;                                  // == C::BN
    constexpr int LDR = MF16 ? BT + 16 : BT;                   // LDS row stride: +16 keeps the 4 rows of a 16x16x4 operand on distinct banks
    constexpr int TNW = C::TMW, TKW = C::TNW;
    constexpr int TPR = BT / 4;                                // threads per staged row (16 B each)
    constexpr int RPP = C::THREADS / TPR;                      // rows per pass (8 in every config)
    constexpr int NP = TM / RPP;                               // passes per operand per stage
    __shared__ __attribute__((aligned(16))) float sm[2 * 2 * TM * LDR];      // [buf][A|B][TM][LDR]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane & 31, hh = lane >> 5;
    const int wn = wave / C::WN, wk = wave % C::WN;
    const int tiles_n = p.N / BT, tiles_k = p.K / BT, n_tiles = tiles_n * tiles_k;
    // blocks that share one M slice (all dW tiles of a split) sit on one XCD and run back to back,
    // so the slice of A / B they all read is served by that XCD's L2 after the first touch
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int tile = j % n_tiles;
    const int split = (j / n_tiles) * 8 + xcd;
    if (split >= p.n_split) return;
    const int n0 = (tile / tiles_k) * BT, k0 = (tile % tiles_k) * BT;
    const int64_t r_begin = (int64_t)split * p.rows_per_split;
    const int64_t r_end = r_begin + p.rows_per_split < p.M ? r_begin + p.rows_per_split : p.M;
    if (r_begin >= r_end) return;
    const int n_it = RAGGED ? (int)((r_end - r_begin + TM - 1) / TM) : (int)((r_end - r_begin) / TM);
    const int64_t m_last = p.M - 1;

    const int lrow = tid / TPR, lc4 = (tid % TPR) * 4;         // rows lrow + RPP*i, floats lc4..lc4+3
    const uint32_t pro_th = drop_thresh(p.pro_p);
    const float pro_sc = 1.0f / (1.0f - p.pro_p);
    f32x4 ra[NP], rb[NP];
    float bm[NP], bs[NP];
    f32x4 colsum = {0.f, 0.f, 0.f, 0.f};
    f32x4 lg = {1.f, 1.f, 1.f, 1.f}, lb = {0.f, 0.f, 0.f, 0.f};
    if constexpr (BLN) {
        lg = *reinterpret_cast<const f32x4*>(p.gamma + k0 + lc4);
        lb = *reinterpret_cast<const f32x4*>(p.beta + k0 + lc4);
    }

    auto issue = [&](int it) {
        const int64_t r0 = r_begin + (int64_t)it * TM + lrow;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            int64_t row = r0 + RPP * i;
            if constexpr (RAGGED) row = row < m_last ? row : m_last;
            ra[i] = *reinterpret_cast<const f32x4*>(p.A + row * p.N + n0 + lc4);
            rb[i] = *reinterpret_cast<const f32x4*>(p.B + row * p.K + k0 + lc4);
            if constexpr (BLN) { bm[i] = p.mean[row]; bs[i] = p.rstd[row]; }
        }
    };
    auto commit = [&](int buf, int it) {
        float* As = sm + buf * (2 * TM * LDR);
        float* Bs = As + TM * LDR;
        const int64_t r0 = r_begin + (int64_t)it * TM + lrow;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            f32x4 a = ra[i], b = rb[i];
            if constexpr (PRO == PRO_DROP)
                a *= drop_keep4(p.pro_seed, (uint64_t)((RAGGED ? p.row0 : 0) + r0 + RPP * i) * p.N + n0 + lc4, pro_th, pro_sc);
            if constexpr (BLN) b = (b - bm[i]) * bs[i] * lg + lb;
            if constexpr (RAGGED) { if (r0 + RPP * i > m_last) a = f32x4{0.f, 0.f, 0.f, 0.f}; }
            colsum += a;
            *reinterpret_cast<f32x4*>(As + (lrow + RPP * i) * LDR + lc4) = a;
            *reinterpret_cast<f32x4*>(Bs + (lrow + RPP * i) * LDR + lc4) = b;
        }
    };

    // accumulators: 32x32 tiles (f32x16) or 16x16 tiles (f32x4), 64 registers per lane either way
    constexpr int TI = MF16 ? TNW * 2 : TNW, TJ = MF16 ? TKW * 2 : TKW, TS = MF16 ? 16 : 32;
    typedef typename std::conditional<MF16, f32x4, f32x16>::type acc_t;
    acc_t acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int jj = 0; jj < TJ; ++jj)
#pragma unroll
            for (int e = 0; e < (MF16 ? 4 : 16); ++e) acc[i][jj][e] = 0.f;

    issue(0);
    commit(0, 0);
    __syncthreads();
    int buf = 0;
    // operand lane maps: 32x32x2: lane (col = lane&31, k = lane>>5), 2 rows of m per step;
    //                    16x16x4: lane (col = lane&15, k = lane>>4), 4 rows of m per step
    const int lc = MF16 ? (lane & 15) : lq, lk = MF16 ? (lane >> 4) : hh;
    constexpr int KPS = MF16 ? 4 : 2;                          // rows of m consumed per MFMA step
    for (int it = 0; it < n_it; ++it) {
        const bool have_next = it + 1 < n_it;
        if (have_next) issue(it + 1);
        const float* As = sm + buf * (2 * TM * LDR) + wn * (TNW * 32) + lc;
        const float* Bs = sm + buf * (2 * TM * LDR) + TM * LDR + wk * (TKW * 32) + lc;
        // software-pipelined operand fetch: the LDS reads of chunk c+1 are in flight while the
        // MFMAs of chunk c issue (the compiler then waits with counted lgkmcnt, not 0)
        constexpr int NSTEP = TM / KPS, CH = MF16 ? 2 : 4, NCH = NSTEP / CH;
        float fa[2][CH][TI], fb[2][CH][TJ];
        auto fetch = [&](int c, int slot) {
#pragma unroll
            for (int e = 0; e < CH; ++e) {
                const int ro = (KPS * (c * CH + e) + lk) * LDR;
#pragma unroll
                for (int i = 0; i < TI; ++i) fa[slot][e][i] = As[ro + TS * i];
#pragma unroll
                for (int jj = 0; jj < TJ; ++jj) fb[slot][e][jj] = Bs[ro + TS * jj];
            }
        };
        fetch(0, 0);
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            if (c + 1 < NCH) fetch(c + 1, (c + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);          // keep the prefetch above this chunk's MFMAs
#pragma unroll
            for (int e = 0; e < CH; ++e)
#pragma unroll
                for (int i = 0; i < TI; ++i)
#pragma unroll
                    for (int jj = 0; jj < TJ; ++jj) {
                        if constexpr (MF16)
                            acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[c & 1][e][i], fb[c & 1][e][jj], acc[i][jj], 0, 0, 0);
                        else
                            acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c & 1][e][i], fb[c & 1][e][jj], acc[i][jj], 0, 0, 0);
                    }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (have_next) commit(buf ^ 1, it + 1);
        __syncthreads();
        buf ^= 1;
    }
    const bool det = p.det_dw != nullptr;                       // deterministic mode: see TnArgs
    float* dwo = det ? p.det_dw + (int64_t)split * p.N * p.K : p.dW;
    // D[i = n][j = k]: 32x32: lane (k = lq, hh), reg r -> dW[n = crow(r,hh)][k]
    //                  16x16: lane (k = lane&15, q = lane>>4), reg r -> dW[n = 4q + r][k]
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int jj = 0; jj < TJ; ++jj)
#pragma unroll
            for (int r = 0; r < (MF16 ? 4 : 16); ++r) {
                const int n = n0 + wn * (TNW * 32) + i * TS + (MF16 ? 4 * lk + r : crow(r, hh));
                const int k = k0 + wk * (TKW * 32) + jj * TS + lc;
                HWGAT_TN_ACC(det, dwo, (int64_t)n * p.K + k, acc[i][jj][r]);
            }
    if (p.db != nullptr && k0 == 0) {                           // one k-tile column owns the bias gradient
        float* red = sm;                                        // [RPP][BT] partial column sums
        __syncthreads();
        *reinterpret_cast<f32x4*>(red + lrow * BT + lc4) = colsum;
        __syncthreads();
        if (tid < BT) {
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < RPP; ++q) s += red[q * BT + tid];
            HWGAT_TN_ACC(det, det ? p.det_db + (int64_t)split * p.N : p.db, n0 + tid, s);
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

template <int PRO, typename C, bool RAGGED = false>
int launch_nt(const NtArgs& a, int epi, hipStream_t st, bool fold = false) {
    const int64_t tiles = ((a.M + C::BM - 1) / C::BM) * (a.N / C::BN);
    const int grid = (int)(tiles < C::SLOTS ? tiles : C::SLOTS);      // persistent over tiles
    if (fold) {                                                         // PRO_LN_FOLD: plain loaders, row-affine epilogue
        if constexpr (PRO == PRO_NONE) {
            if (epi == EPI_BIAS) gemm_nt_k<PRO_NONE, EPI_BIAS, C, RAGGED, X_LNFOLD><<<grid, C::THREADS, 0, st>>>(a);
            else if (epi == EPI_BIAS_GELU_DROP) gemm_nt_k<PRO_NONE, EPI_BIAS_GELU_DROP, C, RAGGED, X_LNFOLD><<<grid, C::THREADS, 0, st>>>(a);
            else if (epi == EPI_BIAS_GELU_DROP_G) gemm_nt_k<PRO_NONE, EPI_BIAS_GELU_DROP_G, C, RAGGED, X_LNFOLD><<<grid, C::THREADS, 0, st>>>(a);
            else return HWGAT_EINVAL;
            HWGAT_LAUNCH_CHECK();
        }
        return HWGAT_EINVAL;
    }
    if (a.stat_sum != nullptr) {                                        // validated by the caller: PRO_NONE, EPI_BIAS_DROP_RES, M % 128 == 0
        if constexpr (PRO == PRO_NONE && !RAGGED) {
            if (a.mg_K > 0) gemm_nt_k<PRO_NONE, EPI_BIAS_DROP_RES, C, false, 2><<<grid, C::THREADS, 0, st>>>(a);
            else gemm_nt_k<PRO_NONE, EPI_BIAS_DROP_RES, C, false, 1><<<grid, C::THREADS, 0, st>>>(a);
            HWGAT_LAUNCH_CHECK();
        }
        return HWGAT_EINVAL;
    }
    switch (epi) {
        case EPI_BIAS: gemm_nt_k<PRO, EPI_BIAS, C, RAGGED><<<grid, C::THREADS, 0, st>>>(a); break;
        case EPI_BIAS_DROP_RES: gemm_nt_k<PRO, EPI_BIAS_DROP_RES, C, RAGGED><<<grid, C::THREADS, 0, st>>>(a); break;
        case EPI_BIAS_GELU_DROP: gemm_nt_k<PRO, EPI_BIAS_GELU_DROP, C, RAGGED><<<grid, C::THREADS, 0, st>>>(a); break;
        case EPI_GELU_BWD: gemm_nt_k<PRO, EPI_GELU_BWD, C, RAGGED><<<grid, C::THREADS, 0, st>>>(a); break;
        case EPI_BIAS_GELU_DROP_G: gemm_nt_k<PRO, EPI_BIAS_GELU_DROP_G, C, RAGGED><<<grid, C::THREADS, 0, st>>>(a); break;
        case EPI_MUL_AUX: gemm_nt_k<PRO, EPI_MUL_AUX, C, RAGGED><<<grid, C::THREADS, 0, st>>>(a); break;
        case EPI_NONE: gemm_nt_k<PRO, EPI_NONE, C, RAGGED><<<grid, C::THREADS, 0, st>>>(a); break;
        default: return HWGAT_EINVAL;
    }
    HWGAT_LAUNCH_CHECK();
}

// the same launch restricted to rows [r0, r0 + rows) of every M-indexed operand
NtArgs nt_rows(NtArgs a, int64_t r0, int64_t rows) {
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
TnArgs tn_rows(TnArgs a, int64_t r0, int64_t rows) {
    a.A += r0 * a.N;
    a.B += r0 * a.K;
    if (a.mean) { a.mean += r0; a.rstd += r0; }
    a.M = rows;
    a.row0 = r0;
    return a;
}

// tile choice; HWGAT_GEMM_TILE=small|big overrides (A/B measurements only)
int tile_override() {
    static const int v = [] {
        const char* e = lab_env("HWGAT_GEMM_TILE");
        return !e ? 0 : (e[0] == 's' ? 1 : (e[0] == 'b' ? 2 : (e[0] == 'k' ? 3 : (e[0] == 'm' ? 6 : (e[0] == 't' ? 7 : 0)))));
    }();
    return v;
}

template <int PRO, bool BLN, typename C, bool MF16 = false, bool RAGGED = false>
int launch_tn(TnArgs a, hipStream_t st) {
    constexpr int TM = C::BK;
    const int n_tiles = (a.N / C::BM) * (a.K / C::BM);
    // Blocks are equal-sized and fill C::SLOTS resident slots, so they execute in rounds: pick the
    // number of M slices so that blocks = n_split * n_tiles is an exact multiple of the slots (no
    // nearly-empty last round), >= 2 rounds, and every slice is >= 16 LDS stages deep.
    auto gcd = [](int x, int y) { while (y) { int t = x % y; x = y; y = t; } return x; };
    const int r_min = n_tiles / gcd(n_tiles, C::SLOTS);
    // one round where the tiles fill the slots (nearly) evenly -- a second round only doubles the atomic traffic at the
    // end of the M slices (see hwgat_launch_tn256); HWGAT_TN_ROUNDS=2 restores the round-1 rule for A/B runs
    static const int min_rounds = [] { const char* e = lab_env("HWGAT_TN_ROUNDS"); return e ? atoi(e) : 1; }();
    int r = r_min;
    while (r < min_rounds) r += r_min;
    int64_t want = (int64_t)C::SLOTS * r / n_tiles;
    const int64_t max_split = a.M / (TM * 16) > 0 ? a.M / (TM * 16) : 1;
    if (want > max_split) want = max_split;
    if (want < 1) want = 1;
    int64_t rows = (a.M + want - 1) / want;
    rows = (rows + TM - 1) / TM * TM;
    a.n_split = (int)((a.M + rows - 1) / rows);
    a.rows_per_split = rows;
    if (a.det_dw && a.n_split > a.det_cap) return HWGAT_ESHAPE;
    const int grid = ((a.n_split + 7) / 8) * 8 * n_tiles;
    gemm_tn_k<PRO, BLN, C, MF16, RAGGED><<<grid, C::THREADS, 0, st>>>(a);
    HWGAT_LAUNCH_CHECK();
}

}  // namespace

extern "C" int hwgat_linear_nt_f32_ex(const float* A, const float* W, const float* bias, float* C, int64_t M,
                                      int N, int K, int pro, const float* mean, const float* rstd,
                                      const float* gamma, const float* beta, uint32_t pro_seed, float pro_p,
                                      int epi, const float* res, float* C2, const float* aux, uint32_t epi_seed,
                                      float epi_p, float* stat_sum, float* stat_sq, int merge_F, int merge_K,
                                      const uint32_t* seed_base, void* stream) {
    if (!A || !W || !C || M <= 0 || N <= 0 || K <= 0) return HWGAT_EINVAL;
    if (N % 128 || K % 32 || ((M + 127) / 128) * (int64_t)(N / 128) > 0x7fffffff) return HWGAT_ESHAPE;   // any M
    if ((pro == PRO_LN || pro == PRO_LN_FOLD) && (!mean || !rstd || !gamma || !beta)) return HWGAT_EINVAL;
    if (pro == PRO_LN_FOLD) {                                      // gamma = s[N], beta = c[N] of hwgat_ln_fold; whole tiles
        if (epi != EPI_BIAS && epi != EPI_BIAS_GELU_DROP && epi != EPI_BIAS_GELU_DROP_G) return HWGAT_EINVAL;
        if (M % 128) return HWGAT_ESHAPE;
    }
    if (epi == EPI_BIAS_DROP_RES && !res) return HWGAT_EINVAL;
    if ((epi == EPI_BIAS_GELU_DROP || epi == EPI_BIAS_GELU_DROP_G) && !C2) return HWGAT_EINVAL;
    if ((epi == EPI_GELU_BWD || epi == EPI_MUL_AUX) && !aux) return HWGAT_EINVAL;
    if (pro_p < 0.f || pro_p >= 1.f || epi_p < 0.f || epi_p >= 1.f) return HWGAT_EINVAL;
    const bool stat = stat_sum != nullptr || stat_sq != nullptr || merge_K > 0;
    if (stat) {
        if (!stat_sum || !stat_sq || pro != PRO_NONE || epi != EPI_BIAS_DROP_RES) return HWGAT_EINVAL;
        if (M % 256) return HWGAT_ESHAPE;                          // whole tiles of either kernel only
        if (merge_K > 0 && (merge_F <= 0 || (merge_F & 1) || M % ((int64_t)merge_F * merge_K))) return HWGAT_EINVAL;
    }
    if (pro == PRO_DROP && pro_p == 0.f) pro = PRO_NONE;          // eval mode: no mask to hash
    NtArgs a{A, W, bias, C, C2, res, aux, mean, rstd, gamma, beta, M, N, K, pro_seed, epi_seed, pro_p, epi_p, 0, stat_sum, stat_sq, merge_K > 0 ? merge_F : 0, merge_K > 0 ? merge_K : 0};
    a.seed_base = seed_base;
    hipStream_t st = (hipStream_t)stream;
    // a token count that is not a multiple of the 128-row tile: bulk launch over the aligned rows with the
    // unmodified kernels, then one small RAGGED launch for the last M % 128 rows
    const int64_t m_bulk = M / 128 * 128;
    if (m_bulk != M) {
        if (m_bulk) {
            const int rc = hwgat_linear_nt_f32(A, W, bias, C, m_bulk, N, K, pro, mean, rstd, gamma, beta, pro_seed, pro_p,
                                               epi, res, C2, aux, epi_seed, epi_p, seed_base, stream);
            if (rc) return rc;
        }
        const NtArgs t = nt_rows(a, m_bulk, M - m_bulk);
        switch (pro) {
            case PRO_NONE: return launch_nt<PRO_NONE, NtSmall, true>(t, epi, st);
            case PRO_LN: return launch_nt<PRO_LN, NtSmall, true>(t, epi, st);
            case PRO_DROP: return launch_nt<PRO_DROP, NtSmall, true>(t, epi, st);
            default: return HWGAT_EINVAL;
        }
    }
    // Tile choice, measured on MI355X (profiles/r01f_gemm_tile_ab.txt):
    //  - the 8-wave 256x256 tile loses to two independent 128x128 blocks per CU (110 vs 129 TF at
    //    K=512): kept only behind HWGAT_GEMM_TILE=big for A/B runs;
    //  - four resident blocks (K16, 128 VGPRs) are 2-3 % slower than two; K slabs of 16 with THREE
    //    resident blocks per CU are ~1 % slower for plain epilogues but
    //    7-15 % faster when the epilogue is heavy (dropout+residual, GELU, GELU backward): the third
    //    block's MFMAs cover the epilogue's loads/stores.
    // (prologue-carrying launches gain nothing from K16: 570.5 vs 571.1 clips/s)
    // Outputs whose width is a multiple of 256: the 256x256 one-wave-per-SIMD kernel (gemm_f32_nt256.hip) over the
    // 256-aligned rows.  Same box, TFLOP/s, 128x128 kernels -> this one (tools/nt_lab.py, profiles/r02b_nt_lab_*.txt):
    // stage 2 plain dX 131.7 -> 143.2 and 130.4 -> 141.1, LN-prologue qkv 120.6 -> 132.1, fc1 107.8 -> 121.0, fc2 125.9 ->
    // 136.0, GELU-backward 102.7 -> 114.7, dropout-prologue dX 111.1 -> 124.3, projection 119.1 -> 126.4; stage 1 (K = 256
    // ... 768) +1 ... +10 %; stage 0 (N = 256, K = 128) +3 ... +5 %.  HWGAT_NT_KERNEL=old keeps everything on the 128x128
    // kernels, HWGAT_NT256_MINK moves the K threshold (A/B runs).
    static const bool nt_old = [] { const char* e = lab_env("HWGAT_NT_KERNEL"); return e && e[0] == 'o'; }();
    static const int nt256_min_k = [] { const char* e = lab_env("HWGAT_NT256_MINK"); return e ? atoi(e) : 128; }();
    // (serving batches: fewer than 128 tiles of 256 x 256 leave most of the 256 CUs without a tile -- the 128 x 128 kernel
    //  has four times as many; B = 1 eval forward 3.96 -> see profiles/r03_serve_lab.txt)
    if (!nt_old && tile_override() == 0 && N % 256 == 0 && K >= nt256_min_k && M >= 256 &&
        ((M / 256) * (N / 256) >= 128 || a.stat_sum != nullptr)) {     // (the row statistics of the 256-wide kernels are the order-fixed ones: eval determinism)
        const int64_t m256 = M / 256 * 256;
        NtArgs b = a;
        b.M = m256;
        // (an eight-wave ping-pong twin of the bf16 kernel gemm_bf16_nt8w.hip was built and measured in round 3: the same
        // time within +-2 % on every launch, 55.14 vs 55.39 ms per step -- fp32 MFMA launches are bound by the matrix
        // pipe at the clock the chip sustains, not by staging or epilogue issue; it lives in tools/f32_nt8w/, not here)
        const int rc = hwgat_launch_nt256(b, pro, epi, st);
        if (rc || m256 == M) return rc;
        const NtArgs t = nt_rows(a, m256, M - m256);      // 128 rows left: the RAGGED instantiation hashes dropout
        switch (pro) {                                     // masks with the global row index (row0)
            case PRO_NONE: return launch_nt<PRO_NONE, NtSmall, true>(t, epi, st);
            case PRO_LN_FOLD: return launch_nt<PRO_NONE, NtSmall, true>(t, epi, st, true);
            case PRO_LN: return launch_nt<PRO_LN, NtSmall, true>(t, epi, st);
            case PRO_DROP: return launch_nt<PRO_DROP, NtSmall, true>(t, epi, st);
            default: return HWGAT_EINVAL;
        }
    }
    const bool heavy = epi == EPI_BIAS_DROP_RES || epi == EPI_BIAS_GELU_DROP || epi == EPI_GELU_BWD || epi == EPI_BIAS_GELU_DROP_G || epi == EPI_MUL_AUX;
    const bool big = tile_override() == 2 && (M % 256 == 0) && (N % 256 == 0);
    const bool k16 = tile_override() == 3 || (tile_override() == 0 && heavy);
#define NT_GO(P) return big ? launch_nt<P, NtBig>(a, epi, st) : (k16 ? launch_nt<P, NtK16>(a, epi, st) : launch_nt<P, NtSmall>(a, epi, st))
    switch (pro) {
        case PRO_NONE: NT_GO(PRO_NONE);
        case PRO_LN_FOLD: return k16 ? launch_nt<PRO_NONE, NtK16>(a, epi, st, true) : launch_nt<PRO_NONE, NtSmall>(a, epi, st, true);
        case PRO_LN: NT_GO(PRO_LN);
        case PRO_DROP: NT_GO(PRO_DROP);
        default: return HWGAT_EINVAL;
    }
#undef NT_GO
}

extern "C" int hwgat_linear_nt_f32(const float* A, const float* W, const float* bias, float* C, int64_t M,
                                   int N, int K, int pro, const float* mean, const float* rstd,
                                   const float* gamma, const float* beta, uint32_t pro_seed, float pro_p,
                                   int epi, const float* res, float* C2, const float* aux, uint32_t epi_seed,
                                   float epi_p, const uint32_t* seed_base, void* stream) {
    return hwgat_linear_nt_f32_ex(A, W, bias, C, M, N, K, pro, mean, rstd, gamma, beta, pro_seed, pro_p, epi, res, C2, aux,
                                  epi_seed, epi_p, nullptr, nullptr, 0, 0, seed_base, stream);
}

static bool tn256_takes(int64_t M, int N, int K, float pro_p, const float* mean) {
    const int ov = tile_override();
    return M % 32 == 0 && (ov == 0 || ov == 7) && N % 256 == 0 && K % 256 == 0 && (int64_t)N * K > 256 * 256 && !(pro_p > 0.f && mean);
}

extern "C" int64_t hwgat_linear_tn_f32_ws_bytes(int64_t M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0 || !tn256_takes(M, N, K, 0.f, nullptr)) return 0;
    return hwgat_tn256_ws_floats(M, N, K) * 4;
}

extern "C" int hwgat_linear_tn_f32_ws(const float* A, const float* B, float* dW, float* db, int64_t M, int N,
                                      int K, uint32_t pro_seed, float pro_p, const float* mean,
                                      const float* rstd, const float* gamma, const float* beta, float* ws,
                                      int64_t ws_bytes, const uint32_t* seed_base, void* stream) {
    if (!A || !B || !dW || M <= 0 || N <= 0 || K <= 0) return HWGAT_EINVAL;
    if (mean && (!rstd || !gamma || !beta)) return HWGAT_EINVAL;
    if (pro_p < 0.f || pro_p >= 1.f) return HWGAT_EINVAL;
    if (ws && ws_bytes > 0 && N % 128 == 0 && K % 128 == 0 && tn256_takes(M, N, K, pro_p, mean)) {
        TnArgs a{A, B, dW, db, mean, rstd, gamma, beta, M, N, K, 0, 0, pro_seed, pro_p, 0};
        a.seed_base = seed_base;
        return hwgat_launch_tn256(a, (hipStream_t)stream, ws, ws_bytes / 4);
    }
    return hwgat_linear_tn_f32(A, B, dW, db, M, N, K, pro_seed, pro_p, mean, rstd, gamma, beta, seed_base, stream);
}

static int tn_f32_impl(const float* A, const float* B, float* dW, float* db, int64_t M, int N,
                       int K, uint32_t pro_seed, float pro_p, const float* mean,
                       const float* rstd, const float* gamma, const float* beta,
                       const uint32_t* seed_base, DetWs det, void* stream) {
    if (!A || !B || !dW || M <= 0 || N <= 0 || K <= 0) return HWGAT_EINVAL;
    if (mean && (!rstd || !gamma || !beta)) return HWGAT_EINVAL;
    if (N % 128 || K % 128) return HWGAT_ESHAPE;                 // any M
    if (pro_p < 0.f || pro_p >= 1.f) return HWGAT_EINVAL;
    TnArgs a{A, B, dW, db, mean, rstd, gamma, beta, M, N, K, 0, 0, pro_seed, pro_p, 0};
    a.seed_base = seed_base;
    a.det_dw = det.dw; a.det_db = det.db; a.det_cap = det.cap;
    hipStream_t st = (hipStream_t)stream;
    const int64_t m_bulk = M / 32 * 32;                         // rows per LDS stage; the tail gets a RAGGED launch
    if (m_bulk != M) {
        if (det.dw) return HWGAT_ESHAPE;                        // deterministic mode: whole stages only (one launch, one reduction)
        if (m_bulk) {
            const int rc = hwgat_linear_tn_f32(A, B, dW, db, m_bulk, N, K, pro_seed, pro_p, mean, rstd, gamma, beta, seed_base, stream);
            if (rc) return rc;
        }
        const TnArgs t = tn_rows(a, m_bulk, M - m_bulk);
        if (pro_p > 0.f) return mean ? launch_tn<PRO_DROP, true, TnSmall, false, true>(t, st) : launch_tn<PRO_DROP, false, TnSmall, false, true>(t, st);
        return mean ? launch_tn<PRO_NONE, true, TnSmall, false, true>(t, st) : launch_tn<PRO_NONE, false, TnSmall, false, true>(t, st);
    }
    // 256-aligned multi-tile outputs: the one-wave-per-SIMD 256x256 kernel with pinned MFMA/memory
    // interleave (gemm_f32_tn256.hip; +2..8 % over the variants below, e.g. 131 vs 121-126 TF on
    // 1536x512).  Single 256x256 outputs and everything else: 128x128 blocks, two per CU.
    // HWGAT_GEMM_TILE = small | big | k16 | m(fma16) select the alternatives for A/B runs.
    const int ov = tile_override();
    if ((ov == 0 || ov == 7) && N % 256 == 0 && K % 256 == 0 && (int64_t)N * K > 256 * 256 && !(pro_p > 0.f && mean))
        return hwgat_launch_tn256(a, st);
    const bool big = ov == 2 && (N % 256 == 0) && (K % 256 == 0);
    const bool k16 = ov == 3;
    const bool mf16 = ov == 6;
#define TN_GO(P, L) return mf16 ? launch_tn<P, L, TnSmall, true>(a, st) : (big ? launch_tn<P, L, TnBig>(a, st) : (k16 ? launch_tn<P, L, TnK16>(a, st) : launch_tn<P, L, TnSmall>(a, st)))
    if (pro_p > 0.f) { if (mean) TN_GO(PRO_DROP, true); else TN_GO(PRO_DROP, false); }
    if (mean) TN_GO(PRO_NONE, true);
    TN_GO(PRO_NONE, false);
#undef TN_GO
}

extern "C" int hwgat_linear_tn_f32(const float* A, const float* B, float* dW, float* db, int64_t M, int N,
                                   int K, uint32_t pro_seed, float pro_p, const float* mean,
                                   const float* rstd, const float* gamma, const float* beta,
                                   const uint32_t* seed_base, void* stream) {
    return tn_f32_impl(A, B, dW, db, M, N, K, pro_seed, pro_p, mean, rstd, gamma, beta, seed_base, DetWs{nullptr, nullptr, 0}, stream);
}

// ---- deterministic weight gradients (see TnArgs::det_dw)
namespace {
__global__ __launch_bounds__(256) void tn_det_reduce_k(const float* __restrict__ ws, float* __restrict__ out, int cap, int64_t stride,
                                                       int64_t count) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    float s = 0.f;
    int sp = 0;
    for (; sp + 8 <= cap; sp += 8) {                           // eight loads in flight, added in split order
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = ws[(sp + j) * stride + i];
#pragma unroll
        for (int j = 0; j < 8; ++j) s += v[j];
    }
    for (; sp < cap; ++sp) s += ws[sp * stride + i];
    out[i] += s;
}
}  // namespace
int hwgat_tn_det_reduce(const float* ws, float* out, int cap, int64_t stride, int64_t count, hipStream_t st) {
    tn_det_reduce_k<<<(unsigned)((count + 255) / 256), 256, 0, st>>>(ws, out, cap, stride, count);
    HWGAT_LAUNCH_CHECK();
}

// M splits any dW kernel may choose for this shape (an upper bound over the kernels of both dtypes), + margin
static int64_t tn_det_cap(int64_t M, int N, int K) {
    auto gcd = [](int64_t x, int64_t y) { while (y) { int64_t t = x % y; x = y; y = t; } return x; };
    const int64_t t128 = (int64_t)(N / 128) * (K / 128);
    int64_t cap = 0;
    for (int64_t slots : {512, 768, 1024}) {                    // fp32 / bf16 128x128-tile kernels (2, 3, 4 resident blocks per CU)
        const int64_t r = t128 / gcd(t128, slots);
        cap = cap > slots * r / t128 ? cap : slots * r / t128;
    }
    if (N % 256 == 0 && K % 256 == 0) {                         // 256x256-tile kernels: multiples of 8, at least 8
        const int64_t t256 = (int64_t)(N / 256) * (K / 256), r = t256 / gcd(t256, 256);
        const int64_t w = 256 * r / t256 > 8 ? 256 * r / t256 : 8;
        cap = cap > w ? cap : w;
    }
    const int64_t by_rows = M / 32 + 1;                          // never more splits than 32-row stages
    cap = cap < by_rows ? cap : by_rows;
    return cap + 8;
}
extern "C" int64_t hwgat_linear_tn_det_bytes(int64_t M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0 || N % 128 || K % 128) return 0;
    return tn_det_cap(M, N, K) * ((int64_t)N * K + N) * 4;
}

// Deterministic form of hwgat_linear_tn_f32 (see hwgat_linear_tn_bf16_det): zero-filled workspace of
// hwgat_linear_tn_det_bytes(M, N, K) bytes, M % 32 == 0.
extern "C" int hwgat_linear_tn_f32_det(const float* A, const float* B, float* dW, float* db, int64_t M, int N,
                                       int K, uint32_t pro_seed, float pro_p, const float* mean,
                                       const float* rstd, const float* gamma, const float* beta,
                                       const uint32_t* seed_base, float* ws, int64_t ws_bytes, void* stream) {
    if (!ws || N <= 0 || K <= 0) return HWGAT_EINVAL;
    const int64_t per = (int64_t)N * K + N;
    const int64_t cap = ws_bytes / 4 / per;
    if (cap < 1) return HWGAT_ESHAPE;
    const DetWs det{ws, ws + cap * (int64_t)N * K, (int)(cap > 0x7fffffff ? 0x7fffffff : cap)};
    int rc = tn_f32_impl(A, B, dW, db, M, N, K, pro_seed, pro_p, mean, rstd, gamma, beta, seed_base, det, stream);
    if (rc) return rc;
    rc = hwgat_tn_det_reduce(det.dw, dW, det.cap, (int64_t)N * K, (int64_t)N * K, (hipStream_t)stream);
    if (rc || !db) return rc;
    return hwgat_tn_det_reduce(det.db, db, det.cap, N, N, (hipStream_t)stream);
}

extern "C" int hwgat_transpose_f32(const float* in, float* out, int R, int C, void* stream) {
    if (!in || !out || R <= 0 || C <= 0) return HWGAT_EINVAL;
    dim3 grid((C + 31) / 32, (R + 31) / 32), block(32, 8);
    transpose_k<<<grid, block, 0, (hipStream_t)stream>>>(in, out, R, C);
    HWGAT_LAUNCH_CHECK();
}

// expose the dropout hash so host tests can reproduce masks bit for bit
__global__ void drop_mask_k(float* out, int64_t n, uint32_t seed, float p, const uint32_t* seed_base) {
    seed += seed_base_of(seed_base);
    const uint32_t th = drop_thresh(p);
    const float sc = 1.0f / (1.0f - p);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        out[i] = drop_keep(seed, (uint64_t)i, th, sc);
}
extern "C" int hwgat_dropout_mask_f32(float* out, int64_t n, uint32_t seed, float p, const uint32_t* seed_base, void* stream) {
    if (!out || n <= 0 || p < 0.f || p >= 1.f) return HWGAT_EINVAL;
    const int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    drop_mask_k<<<grid, 256, 0, (hipStream_t)stream>>>(out, n, seed, p, seed_base);
    HWGAT_LAUNCH_CHECK();
}
