// bf16 NT linear GEMM with 128x128 WAVE tiles for HWGAT on gfx950 (BASELINE config 3): the large-N companion of
// gemm_nt_bf16_k (gemm_bf16.hip) and the bf16 twin of gemm_nt256_k (gemm_f32_nt256.hip), same semantics
//   C[M,N] = pro(A)[M,K] . W[N,K]^T (+ fused epilogue), A / W / C / C2 / res / aux bf16, bias / LayerNorm fp32,
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation (reference hwgat/models/HWGATE.py:86,115-116,131-135,203,217,219).
//
// With bf16 MFMA (16x the fp32 rate) the 128x128 kernel is bound by the L2 -> LDS stream, not by the matrix pipe:
// M N K 2 B (1/BM + 1/BN) = 6x the HBM bytes at the stage-2 shapes, ~10-12 TB/s, 750 TFLOP/s.  A 256x256 tile halves
// that stream.  Structure = gemm_nt256_k: one wave per SIMD, 4 waves x (128x128) = 4x4 MFMA tiles in 256 AGPRs,
// K slabs of 64 elements (128-byte rows, 144-byte LDS stride: a lane's operand for one k-step of 16 is one
// conflict-free ds_read_b128) double-buffered in LDS, ONE pinned slab body for every slab (the loaders are retargeted
// before a tile's last slab), persistent over tiles in the XCD-aware order, epilogue through the idle LDS buffer
// with the residual / pre-activation operand of a piece requested before the piece is parked, 16-byte bf16 stores.
// The byte geometry of the LDS image is identical to the fp32 kernel's (32 floats = 64 bf16 per row and slab).
// Needs M % 256 == N % 256 == K % 64 == 0; hwgat_linear_nt_bf16 sends everything else to gemm_nt_bf16_k.
#include <type_traits>
#include "common.h"
#include "fused_ops.h"
#include "gemm_bf16.h"

namespace {

constexpr int BT = 256, BK = 64, LDB = 144;                     // tile edge, slab depth (elements), LDS row stride (bytes)
constexpr int BUFB = 2 * BT * LDB;                              // bytes per LDS buffer: A rows | W rows (73 728)
constexpr int SLD = 132;                                        // epilogue staging row stride (floats)

constexpr int SG_MFMA = 0x008, SG_VALU = 0x002, SG_VMEM_RD = 0x020, SG_DS_RD = 0x100, SG_DS_WR = 0x200;

// pin one 16-MFMA chunk: G groups of { 16/G MFMAs, NV vector instructions, N1 x M1, N2 x M2 }
template <int G, int M1, int N1, int M2 = 0, int N2 = 0, int NV = 0>
__device__ __forceinline__ void pin() {
#pragma unroll
    for (int q = 0; q < G; ++q) {
        __builtin_amdgcn_sched_group_barrier(SG_MFMA, 16 / G, 0);
        if constexpr (NV > 0) __builtin_amdgcn_sched_group_barrier(SG_VALU, NV, 0);
        if constexpr (N1 > 0) __builtin_amdgcn_sched_group_barrier(M1, N1, 0);
        if constexpr (N2 > 0) __builtin_amdgcn_sched_group_barrier(M2, N2, 0);
    }
}

// STAT: 0 = plain epilogue, 1 = + row statistics, 2 = + row statistics and the merged store (separate instantiations)
template <int PRO, int EPI, int STAT = 0>
__global__ __launch_bounds__(256, 1) void gemm_nt256_bf16_k(NtArgsB p) {
    HWGAT_RESOLVE_SEEDS2(p);
    __shared__ __attribute__((aligned(16))) unsigned char smb[2 * BUFB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane & 31, hh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = p.N / BT;
    const int row_blocks = (int)(p.M / BT);
    const int n_tiles = row_blocks * tiles_n;
    const int n_slab = p.K / BK;
    const int lrow = tid >> 3, lc8 = (tid & 7) * 8;             // this thread stages rows lrow + 32 i (i < 8), elements lc8..lc8+7
    const uint32_t pro_th = drop_thresh(p.pro_p);
    const float pro_sc = 1.0f / (1.0f - p.pro_p);

    u32x4 ra[8], rw[8];
    float ln_mean[8], ln_rstd[8];
    f32x4 ln_g0, ln_g1, ln_b0, ln_b1;

    const int swz_tiles = (row_blocks / 8) * 8 * tiles_n;       // XCD-aware tile order, see gemm_nt_k
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
        m0 = (int64_t)rb * BT;
        n0 = nt * BT;
    };
    const bf16_t* pa; const bf16_t* pw;
    auto set_tile = [&](int64_t m0, int n0) {                   // which tile `issue` reads / whose statistics `commit` applies
        pa = p.A + (m0 + lrow) * p.K + lc8;
        pw = p.W + (int64_t)(n0 + lrow) * p.K + lc8;
        if constexpr (PRO == PRO_LN) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { ln_mean[i] = p.mean[m0 + lrow + 32 * i]; ln_rstd[i] = p.rstd[m0 + lrow + 32 * i]; }
        }
    };
    const int64_t rstep = (int64_t)32 * p.K;
    auto issue = [&](int slab) {
        const bf16_t* a = pa + slab * BK;
        const bf16_t* w = pw + slab * BK;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            ra[i] = *reinterpret_cast<const u32x4*>(a + i * rstep);
            rw[i] = *reinterpret_cast<const u32x4*>(w + i * rstep);
        }
        if constexpr (PRO == PRO_LN) {
            const float* gp = p.gamma + slab * BK + lc8;
            const float* bp = p.beta + slab * BK + lc8;
            ln_g0 = *reinterpret_cast<const f32x4*>(gp); ln_g1 = *reinterpret_cast<const f32x4*>(gp + 4);
            ln_b0 = *reinterpret_cast<const f32x4*>(bp); ln_b1 = *reinterpret_cast<const f32x4*>(bp + 4);
        }
    };
    auto commit = [&](int buf, int64_t m0, int slab) {
        unsigned char* As = smb + buf * BUFB + lrow * LDB + lc8 * 2;
        unsigned char* Ws = As + BT * LDB;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            u32x4 a = ra[i];
            if constexpr (PRO == PRO_LN) {
                float v[8];
                unpack8(a, v);
                const float mu = ln_mean[i], rs = ln_rstd[i];
                v[0] = (v[0] - mu) * (rs * ln_g0.x) + ln_b0.x; v[1] = (v[1] - mu) * (rs * ln_g0.y) + ln_b0.y;
                v[2] = (v[2] - mu) * (rs * ln_g0.z) + ln_b0.z; v[3] = (v[3] - mu) * (rs * ln_g0.w) + ln_b0.w;
                v[4] = (v[4] - mu) * (rs * ln_g1.x) + ln_b1.x; v[5] = (v[5] - mu) * (rs * ln_g1.y) + ln_b1.y;
                v[6] = (v[6] - mu) * (rs * ln_g1.z) + ln_b1.z; v[7] = (v[7] - mu) * (rs * ln_g1.w) + ln_b1.w;
                a = pack8(v);
            } else if constexpr (PRO == PRO_DROP) {             // branch-free (p = 0 keeps everything, scale 1)
                const uint64_t e0 = (uint64_t)(m0 + lrow + 32 * i) * p.K + slab * BK + lc8;
                const f32x4 k0 = drop_keep4(p.pro_seed, e0, pro_th, pro_sc), k1 = drop_keep4(p.pro_seed, e0 + 4, pro_th, pro_sc);
                float v[8];
                unpack8(a, v);
                v[0] *= k0.x; v[1] *= k0.y; v[2] *= k0.z; v[3] *= k0.w; v[4] *= k1.x; v[5] *= k1.y; v[6] *= k1.z; v[7] *= k1.w;
                a = pack8(v);
            }
            *reinterpret_cast<u32x4*>(As + i * 32 * LDB) = a;
            *reinterpret_cast<u32x4*>(Ws + i * 32 * LDB) = rw[i];
        }
    };
    constexpr int PRO_VALU = PRO == PRO_LN ? 18 : (PRO == PRO_DROP ? 40 : 0);   // vector instructions per staged (A, W) row pair

    f32x16 acc[4][4];
    struct Frag { bf16x8 a[4], b[4]; };                         // operands of ONE k-step of 16: 4 A row tiles, 4 W row tiles
    Frag f0, f1;
    auto fetch = [&](Frag& f, int buf, int kk) {                // lane (row lq, k group hh): 8 consecutive k = 16 bytes
        const unsigned char* ap = smb + buf * BUFB + (wm * 128 + lq) * LDB + 16 * hh + 32 * kk;
        const unsigned char* wp = smb + buf * BUFB + BT * LDB + (wn * 128 + lq) * LDB + 16 * hh + 32 * kk;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f.a[i] = *reinterpret_cast<const bf16x8*>(ap + i * 32 * LDB);
            f.b[i] = *reinterpret_cast<const bf16x8*>(wp + i * 32 * LDB);
        }
    };
    auto mfma16 = [&](const Frag& f) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i], f.b[j], acc[i][j], 0, 0, 0);
    };
    // one slab = 4 k-steps of 16 MFMAs.  The 16 global loads of the NEXT slab ride in k-step 0, its LDS commit (with the
    // prologue arithmetic) in k-step 2; one body for every slab (see gemm_nt256_k).
    auto slab_body = [&](int buf, int64_t m_next, int s_next) {
        fetch(f1, buf, 1);
        issue(s_next);
        mfma16(f0);
        pin<16, SG_DS_RD, 1, SG_VMEM_RD, 1>();
        __builtin_amdgcn_sched_barrier(0);
        fetch(f0, buf, 2);
        mfma16(f1);
        pin<8, SG_DS_RD, 1>();
        __builtin_amdgcn_sched_barrier(0);
        fetch(f1, buf, 3);
        commit(buf ^ 1, m_next, s_next);
        mfma16(f0);
        pin<8, SG_DS_WR, 2, SG_DS_RD, 1, PRO_VALU>();
        __builtin_amdgcn_sched_barrier(0);
        mfma16(f1);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    };

    int t = blockIdx.x;
    if (t >= n_tiles) return;
    int64_t m0; int n0;
    tile_origin(t, m0, n0);
    set_tile(m0, n0);
    issue(0);
    commit(0, m0, 0);
    __syncthreads();
    int buf = 0;

    while (true) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

        const int tn = t + gridDim.x;
        int64_t mn = m0; int nn = n0;
        for (int s = 0; s < n_slab; ++s) {
            const bool last = s + 1 == n_slab;
            if (last) {                                         // the slab streamed in next belongs to the next tile
                if (tn < n_tiles) tile_origin(tn, mn, nn);
                set_tile(mn, nn);
            }
            fetch(f0, buf, 0);
            slab_body(buf, last ? mn : m0, last ? 0 : s + 1);
            buf ^= 1;
        }
        // buf = the next tile's slab 0 (if any); buf^1 is idle now

        // ---- epilogue.  acc[i][j]: lane (n = lq, hh), reg r -> C[m = 32 i + crow(r,hh)][n = 32 j + lq] of the wave tile.
        // A wave parks 32 x 128 fp32 pieces in the idle buffer and walks them 4 rows x 128 columns per pass, 8 columns
        // (16 bytes of bf16) per lane.
        {
            const uint32_t epi_th = drop_thresh(p.epi_p);
            const float epi_sc = 1.0f / (1.0f - p.epi_p);
            float* stg = reinterpret_cast<float*>(smb + (buf ^ 1) * BUFB) + wave * (32 * SLD);
            const int er = lane >> 4, ec = (lane & 15) * 8;     // pass ps covers rows 4 ps + er, columns ec..ec+7
            // per-row partial (sum, sum of squares) of each column half: [wn][256][2] behind the staging strips; every
            // slot has exactly one writer (plain LDS stores, no atomics, nothing to zero)
            float* rowstat = reinterpret_cast<float*>(smb + (buf ^ 1) * BUFB) + 4 * (32 * SLD);
            static_assert(STAT == X_NONE || STAT == X_LNFOLD || (4 * 32 * SLD + 4 * BT) * 4 <= BUFB, "no room for the row statistics");
            const int col = n0 + wn * 128 + ec;
            f32x4 bv0 = {0.f, 0.f, 0.f, 0.f}, bv1 = bv0;
            if constexpr (epi_has_bias(EPI))
                if (p.bias) { bv0 = *reinterpret_cast<const f32x4*>(p.bias + col); bv1 = *reinterpret_cast<const f32x4*>(p.bias + col + 4); }
            f32x4 sv0 = bv0, sv1 = bv0, cv0 = bv0, cv1 = bv0;  // X_LNFOLD: s_n and c_n of the lane's 8 columns
            if constexpr (STAT == X_LNFOLD) {
                sv0 = *reinterpret_cast<const f32x4*>(p.gamma + col); sv1 = *reinterpret_cast<const f32x4*>(p.gamma + col + 4);
                cv0 = *reinterpret_cast<const f32x4*>(p.beta + col); cv1 = *reinterpret_cast<const f32x4*>(p.beta + col + 4);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int64_t off0 = (m0 + wm * 128 + i * 32 + er) * p.N + col;       // row 4 ps + er: + 4 ps N
                const int64_t rs4 = 4 * (int64_t)p.N;
                float rr[8], rm[8];                             // X_LNFOLD: rstd and mean of the piece's rows
                if constexpr (STAT == X_LNFOLD) {
                    const int64_t r0 = m0 + wm * 128 + i * 32 + er;
#pragma unroll
                    for (int ps = 0; ps < 8; ++ps) { rr[ps] = p.rstd[r0 + 4 * ps]; rm[ps] = p.mean[r0 + 4 * ps]; }
                }
                MergeWalk mw;
                if constexpr (STAT == X_STAT_MERGE) mw.start(m0 + wm * 128 + i * 32 + er, p.mg_F, p.mg_K, 4);
                u32x4 ex[8];                                    // residual / pre-activation of the piece: in flight while it is parked
                if constexpr (epi_reads_extra(EPI)) {
                    const bf16_t* src = (EPI == EPI_BIAS_DROP_RES ? p.res : p.aux) + off0;
#pragma unroll
                    for (int ps = 0; ps < 8; ++ps) ex[ps] = *reinterpret_cast<const u32x4*>(src + ps * rs4);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) stg[crow(r, hh) * SLD + j * 32 + lq] = acc[i][j][r];
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int ps = 0; ps < 8; ++ps) {
                    const int64_t off = off0 + ps * rs4;
                    f32x4 v0 = *reinterpret_cast<const f32x4*>(stg + (4 * ps + er) * SLD + ec);
                    f32x4 v1 = *reinterpret_cast<const f32x4*>(stg + (4 * ps + er) * SLD + ec + 4);
                    if constexpr (STAT == X_LNFOLD) {
                        const float t = rm[ps] * rr[ps];
                        v0 = v0 * rr[ps] + (cv0 - sv0 * t); v1 = v1 * rr[ps] + (cv1 - sv1 * t);
                    } else {
                        v0 += bv0; v1 += bv1;
                    }
                    float o8[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                    float dk[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
                    if constexpr (epi_drops(EPI)) {
                        if (epi_th) {
                            const f32x4 k0 = drop_keep4(p.epi_seed, (uint64_t)off, epi_th, epi_sc), k1 = drop_keep4(p.epi_seed, (uint64_t)off + 4, epi_th, epi_sc);
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
                    if constexpr (STAT == X_STAT || STAT == X_STAT_MERGE) {                  // statistics of what the next LayerNorm will read: the bf16 values
                        float q[8];
                        unpack8(outv, q);
                        float s1 = 0.f, s2 = 0.f;
#pragma unroll
                        for (int e = 0; e < 8; ++e) { s1 += q[e]; s2 += q[e] * q[e]; }
                        s1 = group_sum<16>(s1); s2 = group_sum<16>(s2);
                        const int lr = wm * 128 + i * 32 + 4 * ps + er;
                        if ((lane & 15) == 0) { f32x2 st = {s1, s2}; *reinterpret_cast<f32x2*>(rowstat + (wn * BT + lr) * 2) = st; }
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
            if constexpr (STAT == X_STAT || STAT == X_STAT_MERGE) {
                __syncthreads();
                int64_t mr = m0 + tid;
                if constexpr (STAT == X_STAT_MERGE) { MergeWalk w; w.start(m0 + tid, p.mg_F, p.mg_K, 1); mr = w.mrow(); }
                atomicAdd(p.stat_sum + mr, rowstat[2 * tid] + rowstat[2 * (BT + tid)]);
                atomicAdd(p.stat_sq + mr, rowstat[2 * tid + 1] + rowstat[2 * (BT + tid) + 1]);
            }
        }
        __syncthreads();            // staging lives in buf^1, which the next tile's second slab overwrites
        t = tn;
        if (t >= n_tiles) break;
        m0 = mn; n0 = nn;
    }
}

template <int PRO>
int launch(const NtArgsB& a, int epi, int grid, hipStream_t st, bool fold = false) {
    if (a.stat_sum != nullptr) {                                // validated by the caller: PRO_NONE, EPI_BIAS_DROP_RES
        if constexpr (PRO == PRO_NONE) {
            if (a.mg_K > 0) gemm_nt256_bf16_k<PRO_NONE, EPI_BIAS_DROP_RES, 2><<<grid, 256, 0, st>>>(a);
            else gemm_nt256_bf16_k<PRO_NONE, EPI_BIAS_DROP_RES, 1><<<grid, 256, 0, st>>>(a);
            HWGAT_LAUNCH_CHECK();
        }
        return HWGAT_EINVAL;
    }
    if (fold) {                                                 // PRO_LN_FOLD: plain loaders, row-affine epilogue
        if constexpr (PRO == PRO_NONE) {
            if (epi == EPI_BIAS) gemm_nt256_bf16_k<PRO_NONE, EPI_BIAS, X_LNFOLD><<<grid, 256, 0, st>>>(a);
            else if (epi == EPI_BIAS_GELU_DROP) gemm_nt256_bf16_k<PRO_NONE, EPI_BIAS_GELU_DROP, X_LNFOLD><<<grid, 256, 0, st>>>(a);
            else if (epi == EPI_BIAS_GELU_DROP_G) gemm_nt256_bf16_k<PRO_NONE, EPI_BIAS_GELU_DROP_G, X_LNFOLD><<<grid, 256, 0, st>>>(a);
            else return HWGAT_EINVAL;
            HWGAT_LAUNCH_CHECK();
        }
        return HWGAT_EINVAL;
    }
    switch (epi) {
        case EPI_BIAS: gemm_nt256_bf16_k<PRO, EPI_BIAS><<<grid, 256, 0, st>>>(a); break;
        case EPI_BIAS_DROP_RES: gemm_nt256_bf16_k<PRO, EPI_BIAS_DROP_RES><<<grid, 256, 0, st>>>(a); break;
        case EPI_BIAS_GELU_DROP: gemm_nt256_bf16_k<PRO, EPI_BIAS_GELU_DROP><<<grid, 256, 0, st>>>(a); break;
        case EPI_GELU_BWD: gemm_nt256_bf16_k<PRO, EPI_GELU_BWD><<<grid, 256, 0, st>>>(a); break;
        case EPI_BIAS_GELU_DROP_G: gemm_nt256_bf16_k<PRO, EPI_BIAS_GELU_DROP_G><<<grid, 256, 0, st>>>(a); break;
        case EPI_MUL_AUX: gemm_nt256_bf16_k<PRO, EPI_MUL_AUX><<<grid, 256, 0, st>>>(a); break;
        case EPI_NONE: gemm_nt256_bf16_k<PRO, EPI_NONE><<<grid, 256, 0, st>>>(a); break;
        default: return HWGAT_EINVAL;
    }
    HWGAT_LAUNCH_CHECK();
}

}  // namespace

int hwgat_launch_nt256_bf16(const NtArgsB& a, int pro, int epi, hipStream_t st) {
    if (a.M % BT || a.N % BT || a.K % BK) return HWGAT_ESHAPE;
    const int64_t tiles = (a.M / BT) * (a.N / BT);
    if (tiles > 0x7fffffff) return HWGAT_ESHAPE;
    const int grid = (int)(tiles < 256 ? tiles : 256);          // persistent: one block per CU
    switch (pro) {
        case PRO_NONE: return launch<PRO_NONE>(a, epi, grid, st);
        case PRO_LN_FOLD: return launch<PRO_NONE>(a, epi, grid, st, true);
        case PRO_LN: return launch<PRO_LN>(a, epi, grid, st);
        case PRO_DROP: return launch<PRO_DROP>(a, epi, grid, st);
        default: return HWGAT_EINVAL;
    }
}
