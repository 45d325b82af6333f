// fp32 NT linear GEMM with 128x128 WAVE tiles for HWGAT on gfx950: the large-N companion of gemm_nt_k
// (gemm_f32.hip), same semantics  C[M,N] = pro(A)[M,K] . W[N,K]^T (+ fused epilogue)  and the same
// prologues / epilogues (LayerNorm / dropout mask on A; bias | bias+dropout+residual | bias+GELU+dropout
// with the pre-activation kept | GELU-backward | none) -- reference hwgat/models/HWGATE.py:86,115-116,
// 131-135,203,217,219 and their autograd backward.
//
// Why a second kernel: with two independent 128x128 blocks per CU (gemm_nt_k) every SIMD hosts two
// waves whose non-MFMA phases (address arithmetic, fragment reads after the barrier, LDS commit, the
// barrier itself, the epilogue) collide at random and cannot be scheduled against each other; the
// matrix pipe sat idle ~20 % of the time.  Here ONE wave per SIMD owns the pipe:
//   * 256x256 C tile per block, 4 waves x (128x128) = 4x4 v_mfma_f32_32x32x2_f32 tiles, the 256
//     accumulator registers in AGPRs, 512-register budget (launch_bounds(256, 1));
//   * K slabs of 32 double-buffered in LDS (rows padded to 36 floats: a lane's fragment for FOUR
//     k-steps is one conflict-free ds_read_b128), 256 MFMAs (16 384 matrix-pipe cycles) per barrier;
//   * the slab body is straight-line code whose order is pinned with sched_group_barrier: fragment
//     reads for the next 64 MFMAs, the 16 global loads of the next slab and its 16 LDS stores are
//     spread one by one between the MFMAs, so the pipe never waits on a cluster of memory instructions;
//   * the prologue arithmetic (LayerNorm apply, dropout hash) rides in the MFMA shadows of the chunk
//     that commits; mean / rstd are reloaded only when the row block changes;
//   * persistent over tiles in an XCD-aware order (the n-tiles of one 256-row block of A run on one
//     XCD, so A streams from HBM once); the next tile's first slab is already in LDS when the
//     epilogue starts;
//   * epilogue: each wave parks 32x128 pieces of its tile in the idle LDS buffer and walks them
//     row-wise (2 rows x 512 B per wave-instruction, 16 B per lane); residual / GELU-backward operands
//     of a piece are requested BEFORE the piece is parked, so their latency hides behind the parking.
// Needs M % 256 == N % 256 == K % 32 == 0; the caller (hwgat_linear_nt_f32) sends everything else to
// gemm_nt_k.
#include <type_traits>
#include "common.h"
#include "fused_ops.h"
#include "gemm_f32.h"

namespace {

constexpr int BT = 256, BK = 32, LDT = 36;                      // tile edge, slab depth, LDS row stride (floats)
constexpr int BUF = 2 * BT * LDT;                               // floats per LDS buffer: A rows | W rows  (73 728 B)
constexpr int SLD = 132;                                        // epilogue staging row stride (128 + 4)

constexpr int SG_MFMA = 0x008, SG_VALU = 0x002, SG_VMEM_RD = 0x020, SG_DS_RD = 0x100, SG_DS_WR = 0x200;

// pin one 64-MFMA chunk: G groups of { 64/G MFMAs, NV vector instructions, N1 x M1, N2 x M2 }
template <int G, int M1, int N1, int M2 = 0, int N2 = 0, int NV = 0>
__device__ __forceinline__ void pin() {
#pragma unroll
    for (int q = 0; q < G; ++q) {
        __builtin_amdgcn_sched_group_barrier(SG_MFMA, 64 / G, 0);
        if constexpr (NV > 0) __builtin_amdgcn_sched_group_barrier(SG_VALU, NV, 0);
        if constexpr (N1 > 0) __builtin_amdgcn_sched_group_barrier(M1, N1, 0);
        if constexpr (N2 > 0) __builtin_amdgcn_sched_group_barrier(M2, N2, 0);
    }
}

// STAT (EPI_BIAS_DROP_RES only): row sums / sums of squares of the output for the next LayerNorm and the optional
// TemporalMerging store, exactly as in gemm_nt_k (see NtArgs in gemm_f32.h)
// (STAT: 0 = plain epilogue, 1 = + row statistics, 2 = + row statistics and the merged store: separate instantiations,
//  because the merged store's index arithmetic would otherwise sit, as a branch per pass, in every epilogue;
//  3 = X_LNFOLD, the row-affine epilogue of a folded LayerNorm -- fused_ops.h)
#ifdef HWGAT_LAB
// lab build only: s_memtime at the start and the end of every workgroup, [workgroup][2], through a pointer set with
// hwgat_lab_nt256_stamps (tools/nt256_clock.py: the shader clock the chip holds under this kernel)
__device__ unsigned long long* g_nt256_stamps = nullptr;
#define NT256_STAMP(i) do { if (g_nt256_stamps && threadIdx.x == 0) g_nt256_stamps[blockIdx.x * 2 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define NT256_STAMP(i) do {} while (0)
#endif

template <int PRO, int EPI, int STAT = 0>
__global__ __launch_bounds__(256, 1) void gemm_nt256_k(NtArgs p) {
    HWGAT_RESOLVE_SEEDS2(p);
    __shared__ __attribute__((aligned(16))) float sm[2 * BUF];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane & 31, hh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = p.N / BT;
    const int row_blocks = (int)(p.M / BT);
    const int n_tiles = row_blocks * tiles_n;
    const int n_slab = p.K / BK;
    const int lrow = tid >> 3, lc4 = (tid & 7) * 4;             // this thread stages rows lrow + 32 i (i < 8), floats lc4..lc4+3
    const uint32_t pro_th = drop_thresh(p.pro_p);
    const float pro_sc = 1.0f / (1.0f - p.pro_p);

    f32x4 ra[8], rw[8];
    float ln_mean[8], ln_rstd[8];
    f32x4 ln_g, ln_b;

    // XCD-aware tile order (see gemm_nt_k): blocks b, b+8, ... share an XCD and take the n-tiles of one row block
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
        m0 = (int64_t)rb * BT;
        n0 = nt * BT;
    };
    // staged-slab source pointers: advanced by BK floats per slab, recomputed on a new tile
    const float* pa; const float* pw;
    auto set_tile = [&](int64_t m0, int n0) {                   // which tile `issue` reads / whose statistics `commit` applies
        pa = p.A + (m0 + lrow) * p.K + lc4;
        pw = p.W + (int64_t)(n0 + lrow) * p.K + lc4;
        if constexpr (PRO == PRO_LN) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { ln_mean[i] = p.mean[m0 + lrow + 32 * i]; ln_rstd[i] = p.rstd[m0 + lrow + 32 * i]; }
        }
    };
    const int64_t rstep = (int64_t)32 * p.K;                    // 32 rows further down
    auto issue = [&](int slab) {
        const float* a = pa + slab * BK;
        const float* w = pw + slab * BK;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            ra[i] = *reinterpret_cast<const f32x4*>(a + i * rstep);
            rw[i] = *reinterpret_cast<const f32x4*>(w + i * rstep);
        }
        if constexpr (PRO == PRO_LN) {
            ln_g = *reinterpret_cast<const f32x4*>(p.gamma + slab * BK + lc4);
            ln_b = *reinterpret_cast<const f32x4*>(p.beta + slab * BK + lc4);
        }
    };
    auto commit = [&](int buf, int64_t m0, int slab) {
        float* As = sm + buf * BUF + lrow * LDT + lc4;
        float* Ws = As + BT * LDT;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            f32x4 a = ra[i];
            if constexpr (PRO == PRO_LN) {
                a = (a - ln_mean[i]) * (ln_rstd[i] * ln_g) + ln_b;
            } else if constexpr (PRO == PRO_DROP) {
                a *= drop_keep4(p.pro_seed, (uint64_t)(m0 + lrow + 32 * i) * p.K + slab * BK + lc4, pro_th, pro_sc);
            }
            *reinterpret_cast<f32x4*>(As + i * 32 * LDT) = a;
            *reinterpret_cast<f32x4*>(Ws + i * 32 * LDT) = rw[i];
        }
    };
    // vector instructions of the prologue per PAIR of staged rows (one A row with arithmetic + one W row without):
    // pinned right in front of that pair's LDS stores, so a row's values are live only across a few MFMAs
    constexpr int PRO_VALU = PRO == PRO_LN ? 14 : (PRO == PRO_DROP ? 44 : 0);

    f32x16 acc[4][4];
    struct Frag { f32x4 a[4], b[4]; };                          // fragments of 4 k-steps: 4 A row tiles, 4 W row tiles
    Frag f0, f1;
    auto fetch = [&](Frag& f, int buf, int kk) {
        const float* ap = sm + buf * BUF + (wm * 128 + lq) * LDT + 4 * hh + 8 * kk;
        const float* wp = sm + buf * BUF + BT * LDT + (wn * 128 + lq) * LDT + 4 * hh + 8 * kk;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f.a[i] = *reinterpret_cast<const f32x4*>(ap + i * 32 * LDT);
            f.b[i] = *reinterpret_cast<const f32x4*>(wp + i * 32 * LDT);
        }
    };
    auto mfma64 = [&](const Frag& f) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[i][e], f.b[j][e], acc[i][j], 0, 0, 0);
    };
    // one slab: 4 chunks of 64 MFMAs.  The 16 global loads of the NEXT slab ride in chunk 0, its LDS commit (with the
    // prologue arithmetic) in chunk 2.  ONE body for every slab: the next slab of a tile's last slab is the next
    // tile's first (the loaders are retargeted before that body runs; when there is no next tile they harmlessly
    // re-read this one), so there is no specialised tile-boundary body for the register allocator to spill in
    // (the round's first version had one, with 50 scratch accesses and an s_waitcnt vmcnt(0) behind every load).
    auto slab_body = [&](int buf, int64_t m_next, int s_next) {
        fetch(f1, buf, 1);
        issue(s_next);
        mfma64(f0);
        pin<16, SG_DS_RD, 1, SG_VMEM_RD, 1>();
        __builtin_amdgcn_sched_barrier(0);
        fetch(f0, buf, 2);
        mfma64(f1);
        pin<8, SG_DS_RD, 1>();
        __builtin_amdgcn_sched_barrier(0);
        fetch(f1, buf, 3);
        commit(buf ^ 1, m_next, s_next);
        mfma64(f0);
        pin<8, SG_DS_WR, 2, SG_DS_RD, 1, PRO_VALU>();
        __builtin_amdgcn_sched_barrier(0);
        mfma64(f1);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    };

    int t = blockIdx.x;
    if (t >= n_tiles) return;
    int64_t m0; int n0;
    tile_origin(t, m0, n0);
    set_tile(m0, n0);
    NT256_STAMP(0);
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

        // ---- epilogue.  acc[i][j]: lane (n = lq, hh), reg r -> C[m = 32 i + crow(r,hh)][n = 32 j + lq] of the wave tile
        {
            const uint32_t epi_th = drop_thresh(p.epi_p);
            const float epi_sc = 1.0f / (1.0f - p.epi_p);
            float* stg = sm + (buf ^ 1) * BUF + wave * (32 * SLD);
            const int er = lane >> 5, ec = (lane & 31) * 4;     // pass ps covers rows 2 ps + er, floats ec..ec+3
            // per-row partial (sum, sum of squares) of each column half: [wn][256][2] behind the staging strips; every
            // slot has exactly one writer (plain LDS stores, no atomics, nothing to zero)
            float* rowstat = sm + (buf ^ 1) * BUF + 4 * (32 * SLD);
            static_assert(STAT == X_NONE || STAT == X_LNFOLD || 4 * 32 * SLD + 4 * BT <= BUF, "no room for the row statistics");
            const int col = n0 + wn * 128 + ec;
            f32x4 bv = {0.f, 0.f, 0.f, 0.f};
            if constexpr (epi_has_bias(EPI))
                if (p.bias) bv = *reinterpret_cast<const f32x4*>(p.bias + col);
            f32x4 sv = bv, cv = bv;                             // X_LNFOLD: s_n and c_n of the lane's columns
            if constexpr (STAT == X_LNFOLD) {
                sv = *reinterpret_cast<const f32x4*>(p.gamma + col);
                cv = *reinterpret_cast<const f32x4*>(p.beta + col);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int64_t off0 = (m0 + wm * 128 + i * 32 + er) * p.N + col;       // row 2 ps + er: + 2 ps N
                float rr[16], rm[16];                           // X_LNFOLD: rstd and mean * rstd of the piece's rows
                if constexpr (STAT == X_LNFOLD) {
                    const int64_t r0 = m0 + wm * 128 + i * 32 + er;
#pragma unroll
                    for (int ps = 0; ps < 16; ++ps) { rr[ps] = p.rstd[r0 + 2 * ps]; rm[ps] = p.mean[r0 + 2 * ps]; }
                }
                const int64_t rs2 = 2 * (int64_t)p.N;
                MergeWalk mw;
                if constexpr (STAT == X_STAT_MERGE) mw.start(m0 + wm * 128 + i * 32 + er, p.mg_F, p.mg_K, 2);
                // operands of this piece (residual / pre-activation): in flight while the piece is parked
                f32x4 ex[16];
                if constexpr (epi_reads_extra(EPI)) {
                    const float* src = (EPI == EPI_BIAS_DROP_RES ? p.res : p.aux) + off0;
#pragma unroll
                    for (int ps = 0; ps < 16; ++ps) ex[ps] = *reinterpret_cast<const f32x4*>(src + ps * rs2);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) stg[crow(r, hh) * SLD + j * 32 + lq] = acc[i][j][r];
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int ps = 0; ps < 16; ++ps) {
                    const int64_t off = off0 + ps * rs2;
                    f32x4 v = *reinterpret_cast<const f32x4*>(stg + (2 * ps + er) * SLD + ec);
                    if constexpr (STAT == X_LNFOLD) v = v * rr[ps] + (cv - sv * (rm[ps] * rr[ps]));
                    else v += bv;
                    f32x4 dk = {1.f, 1.f, 1.f, 1.f};
                    if constexpr (epi_drops(EPI)) {
                        dk = drop_keep4(p.epi_seed, (uint64_t)off, epi_th, epi_sc);
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
                        s1 = group_sum<32>(s1); s2 = group_sum<32>(s2);
                        const int lr = wm * 128 + i * 32 + 2 * ps + er;
                        if ((lane & 31) == 0) { f32x2 st = {s1, s2}; *reinterpret_cast<f32x2*>(rowstat + (wn * BT + lr) * 2) = st; }
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
            if constexpr (STAT == X_STAT || STAT == X_STAT_MERGE) {
                __syncthreads();
                int64_t mr = m0 + tid;                          // 256 consecutive rows -> coalesced global atomics
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
    NT256_STAMP(1);
}

template <int PRO>
int launch(const NtArgs& a, int epi, int grid, hipStream_t st, bool fold = false) {
    if (a.stat_sum != nullptr) {                                // validated by the caller: PRO_NONE, EPI_BIAS_DROP_RES
        if constexpr (PRO == PRO_NONE) {
            if (a.mg_K > 0) gemm_nt256_k<PRO_NONE, EPI_BIAS_DROP_RES, 2><<<grid, 256, 0, st>>>(a);
            else gemm_nt256_k<PRO_NONE, EPI_BIAS_DROP_RES, 1><<<grid, 256, 0, st>>>(a);
            HWGAT_LAUNCH_CHECK();
        }
        return HWGAT_EINVAL;
    }
    if (fold) {                                                 // PRO_LN_FOLD: plain loaders, row-affine epilogue
        if constexpr (PRO == PRO_NONE) {
            if (epi == EPI_BIAS) gemm_nt256_k<PRO_NONE, EPI_BIAS, X_LNFOLD><<<grid, 256, 0, st>>>(a);
            else if (epi == EPI_BIAS_GELU_DROP) gemm_nt256_k<PRO_NONE, EPI_BIAS_GELU_DROP, X_LNFOLD><<<grid, 256, 0, st>>>(a);
            else if (epi == EPI_BIAS_GELU_DROP_G) gemm_nt256_k<PRO_NONE, EPI_BIAS_GELU_DROP_G, X_LNFOLD><<<grid, 256, 0, st>>>(a);
            else return HWGAT_EINVAL;
            HWGAT_LAUNCH_CHECK();
        }
        return HWGAT_EINVAL;
    }
    switch (epi) {
        case EPI_BIAS: gemm_nt256_k<PRO, EPI_BIAS><<<grid, 256, 0, st>>>(a); break;
        case EPI_BIAS_DROP_RES: gemm_nt256_k<PRO, EPI_BIAS_DROP_RES><<<grid, 256, 0, st>>>(a); break;
        case EPI_BIAS_GELU_DROP: gemm_nt256_k<PRO, EPI_BIAS_GELU_DROP><<<grid, 256, 0, st>>>(a); break;
        case EPI_GELU_BWD: gemm_nt256_k<PRO, EPI_GELU_BWD><<<grid, 256, 0, st>>>(a); break;
        case EPI_BIAS_GELU_DROP_G: gemm_nt256_k<PRO, EPI_BIAS_GELU_DROP_G><<<grid, 256, 0, st>>>(a); break;
        case EPI_MUL_AUX: gemm_nt256_k<PRO, EPI_MUL_AUX><<<grid, 256, 0, st>>>(a); break;
        case EPI_NONE: gemm_nt256_k<PRO, EPI_NONE><<<grid, 256, 0, st>>>(a); break;
        default: return HWGAT_EINVAL;
    }
    HWGAT_LAUNCH_CHECK();
}

}  // namespace

int hwgat_launch_nt256(const NtArgs& a, int pro, int epi, hipStream_t st) {
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

#ifdef HWGAT_LAB
extern "C" int hwgat_lab_nt256_stamps(void* buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_nt256_stamps), &buf, sizeof(buf)) == hipSuccess ? 0 : HWGAT_EINVAL;
}
#endif
