// bf16 weight-gradient GEMM with 128x128 WAVE tiles for HWGAT on gfx950: the large-output companion of
// gemm_tn_bf16_k (gemm_bf16.hip), same semantics (dW(fp32) += A^T B over an M slice, db += colsum(A), dropout mask
// on A, LayerNorm on B; A, B bf16) -- the bf16 twin of gemm_tn256_k (gemm_f32_tn256.hip).
//
// Why: with 128x128 tiles every 32-row stage costs a thread the same unpack / normalise / mask / pack work as
// here but feeds a quarter of the MFMAs; the small kernel is VALU- and L2-stream-bound at 440-560 TFLOP/s.
//   * 256x256 dW tile per block, 4 waves x (128x128) = 4x4 v_mfma_f32_32x32x16_bf16 tiles, 256 accumulator
//     registers in AGPRs, ONE wave per SIMD;
//   * 32-row stages (two k16 steps) of both operands in a 3-deep LDS ring, rows padded to 576 bytes so the four
//     rows a ds_read_b64_tr_b16 group touches land in disjoint bank quarters;
//   * the operand needs 8 consecutive m per lane for a fixed column: ds_read_b64_tr_b16 (hardware transpose);
//   * a stage is 1 024 matrix-pipe cycles (0.43 us) and HBM answers in ~2 us, so the global loads run NSET = 3 stages
//     ahead in three register sets: a set is committed (transformed) to the ring at the top of a stage and re-issued for
//     the stage NSET further on right behind it -- 96 KB per CU permanently in flight; one barrier per 32 MFMAs;
//   * one branch-free body per phase, instruction order pinned per MFMA: an LDS read and a share of the commit's
//     VALU work behind every MFMA, LDS writes and global loads spread between them;
//   * db: v_dot2c_f32_bf16 against (1, 0) / (0, 1) sums the two columns of a packed word without unpacking.
#include <stdlib.h>
#include <type_traits>
#include "common.h"
#include "fused_ops.h"
#include "gemm_bf16.h"

namespace {

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;

constexpr int BT = 256, TMB = 32, NST = 3;
constexpr int LDW = 288;                                        // LDS row stride in bf16 (576 B = 2 x 256 + 64)
constexpr int STG = 2 * TMB * LDW;                              // bf16 per stage: A[32][288] | B[32][288] = 36 KB

constexpr int SG_MFMA = 0x008, SG_VMEM_RD = 0x020, SG_DS_RD = 0x100, SG_DS_WR = 0x200;   // LLVM SchedGroupMask bits

typedef __attribute__((address_space(3))) bf16_t lds_bf16;

// 8 consecutive m (k index of the MFMA) for one column -> one operand fragment.  `a0` already carries the lane's
// part of the address (row 8 hh + q of the k-step, column pcol + csub of the fragment): every read of a stage is that
// one register plus an immediate.
__device__ __forceinline__ bf16x8 tr_frag(const lds_bf16* a0) {
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)a0);
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(a0 + 4 * LDW));
    bf16x8 f = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    return f;
}

template <int PRO, bool BLN, int NSET>     // NSET: 2 or 3
__global__ __launch_bounds__(256, 1) void gemm_tn256_bf16_k(TnArgsB p) {
    HWGAT_RESOLVE_SEED1(p);
    __shared__ __attribute__((aligned(16))) bf16_t sm[NST * STG];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane & 31, hh = lane >> 5;
    const int wn = wave >> 1, wk = wave & 1;
    const int tiles_k = p.K / BT, n_tiles = (p.N / BT) * tiles_k;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int tile = j % n_tiles;
    const int split = (j / n_tiles) * 8 + xcd;
    if (split >= p.n_split) return;
    const int n0 = (tile / tiles_k) * BT, k0 = (tile % tiles_k) * BT;
    const int64_t r_begin = (int64_t)split * p.rows_per_split;
    const int64_t r_end = r_begin + p.rows_per_split < p.M ? r_begin + p.rows_per_split : p.M;
    if (r_begin >= r_end) return;
    const int n_it = (int)((r_end - r_begin) / TMB);
    constexpr int VPM = PRO == PRO_DROP ? 14 : BLN ? 6 : 3;      // VALU instructions pinned behind each MFMA (the commit's arithmetic)

    const int lrow = tid >> 5, lc8 = (tid & 31) * 8;            // rows lrow + 8*i (i < 4), bf16 columns lc8..lc8+7
    const uint32_t pro_th = drop_thresh(p.pro_p);
    const float pro_sc = 1.0f / (1.0f - p.pro_p);
    u32x4 ra[NSET][4], rb[NSET][4];                             // staging register sets (stage k uses set k % NSET)
    float bm[NSET][4], bs[NSET][4];
    float colsum[8], lg[8], lb[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { colsum[e] = 0.f; lg[e] = 1.f; lb[e] = 0.f; }
    if constexpr (BLN) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { lg[e] = p.gamma[k0 + lc8 + e]; lb[e] = p.beta[k0 + lc8 + e]; }
    }
    const bool want_db = p.db != nullptr && k0 == 0;
    // (1, 0) and (0, 1) as packed bf16 pairs, zero in the tiles that do not own db
    const uint32_t one_lo = want_db ? 0x00003F80u : 0u, one_hi = want_db ? 0x3F800000u : 0u;
    const float cs_on = want_db ? 1.0f : 0.0f;

    const bf16_t* gA = p.A + (r_begin + lrow) * p.N + n0 + lc8;
    const bf16_t* gB = p.B + (r_begin + lrow) * p.K + k0 + lc8;
    // rows lrow + 8 i of a stage, i = 2 H and 2 H + 1 (H = 0, 1: the halves are requested and committed separately so
    // that the commit's arithmetic spreads over both k16 steps of a stage)
    auto issue = [&](auto QC, int it_want, auto HC) {
        constexpr int Q = decltype(QC)::value, H = decltype(HC)::value;
        const int it = it_want < n_it ? it_want : n_it - 1;
        const int64_t ro = (int64_t)it * TMB;
#pragma unroll
        for (int i = 2 * H; i < 2 * H + 2; ++i) {
            ra[Q][i] = *reinterpret_cast<const u32x4*>(gA + (ro + 8 * i) * p.N);
            rb[Q][i] = *reinterpret_cast<const u32x4*>(gB + (ro + 8 * i) * p.K);
            if constexpr (BLN) {
                const int64_t row = r_begin + lrow + ro + 8 * i;
                bs[Q][i] = p.rstd[row];                         // (no arithmetic on them here: it would wait for the loads just issued)
                bm[Q][i] = p.mean[row];
            }
        }
    };
    // `it` past the end of the M slice (the loaders re-read the last stage there): A is committed as zeros, so the
    // stage adds nothing to dW or db
    auto commit = [&](auto QC, int stage, int it, auto HC) {
        constexpr int Q = decltype(QC)::value, H = decltype(HC)::value;
        bf16_t* As = sm + stage * STG + lrow * LDW + lc8;
        bf16_t* Bs = As + TMB * LDW;
        const uint32_t keep = it < n_it ? 0xffffffffu : 0u;
#pragma unroll
        for (int i = 2 * H; i < 2 * H + 2; ++i) {
            u32x4 a = ra[Q][i], b = rb[Q][i];
            a.x &= keep; a.y &= keep; a.z &= keep; a.w &= keep;
            if constexpr (PRO == PRO_DROP) {
                const uint64_t e0 = (uint64_t)(r_begin + (int64_t)it * TMB + lrow + 8 * i) * p.N + n0 + lc8;
                const f32x4 k0v = drop_keep4(p.pro_seed, e0, pro_th, pro_sc);
                const f32x4 k1v = drop_keep4(p.pro_seed, e0 + 4, pro_th, pro_sc);
                float av[8];
                unpack8(a, av);
                av[0] *= k0v.x; av[1] *= k0v.y; av[2] *= k0v.z; av[3] *= k0v.w;
                av[4] *= k1v.x; av[5] *= k1v.y; av[6] *= k1v.z; av[7] *= k1v.w;
#pragma unroll
                for (int e = 0; e < 8; ++e) colsum[e] += av[e] * cs_on;          // the unrounded masked values, as gemm_tn_bf16_k
                a = pack8(av);
            } else {
                const uint32_t w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bf16x2 v = __builtin_bit_cast(bf16x2, w[e]);
                    colsum[2 * e] = __builtin_amdgcn_fdot2_f32_bf16(v, __builtin_bit_cast(bf16x2, one_lo), colsum[2 * e], false);
                    colsum[2 * e + 1] = __builtin_amdgcn_fdot2_f32_bf16(v, __builtin_bit_cast(bf16x2, one_hi), colsum[2 * e + 1], false);
                }
            }
            if constexpr (BLN) {
                float bv[8];
                unpack8(b, bv);
                const float nm = -bm[Q][i] * bs[Q][i];
#pragma unroll
                for (int e = 0; e < 8; ++e) bv[e] = fmaf(fmaf(bv[e], bs[Q][i], nm), lg[e], lb[e]);
                b = pack8(bv);
            }
            *reinterpret_cast<u32x4*>(As + 8 * i * LDW) = a;
            *reinterpret_cast<u32x4*>(Bs + 8 * i * LDW) = b;
        }
    };

    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][jj][e] = 0.f;

    // operand set of one k16 step: 4 A + 4 B fragments (32 registers)
    struct Frag { bf16x8 a[4], b[4]; };
    // lane part of a fragment address: row 8 hh + q of the k-step (q = quad of the 16-lane group), 4 columns per lane,
    // the second 16-lane group 16 columns further on
    const int lane_off = (8 * hh + ((lane & 15) >> 2)) * LDW + (lane & 3) * 4 + 16 * ((lane >> 4) & 1);
    const lds_bf16* smA = (const lds_bf16*)sm + lane_off + wn * 128;
    const lds_bf16* smB = (const lds_bf16*)sm + lane_off + TMB * LDW + wk * 128;
    auto fetch = [&](Frag& f, int stage, auto SC) {
        constexpr int S = decltype(SC)::value;
        const lds_bf16* As = smA + stage * STG;
        const lds_bf16* Bs = smB + stage * STG;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f.a[i] = tr_frag(As + 16 * S * LDW + 32 * i);
            f.b[i] = tr_frag(Bs + 16 * S * LDW + 32 * i);
        }
    };
    auto mfma_step = [&](const Frag& f) {                        // 16 MFMAs
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
                acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i], f.b[jj], acc[i][jj], 0, 0, 0);
    };
    // one stage (iteration `it`, PH = it % NSET): 2 k16 steps x 16 MFMAs.  Stage it+2 waits in register set
    // (PH+2) % NSET, requested NSET - 1/2 stages ago:
    //   step 0: its first half is transformed and written to the ring; the fragments of step 1 are read;
    //   step 1: that half is re-requested for stage it+2+NSET, the second half is transformed, written and
    //           re-requested; the step-0 fragments of stage it+1 are read.
    // ONE branch-free body per phase for every stage (see gemm_tn256_k); the loop runs whole groups of NSET stages.
    // Instruction order pinned per MFMA: the commit's VALU work (5-10 instructions per MFMA) has to sit BETWEEN the
    // MFMAs, not in front of the LDS write it feeds -- left to the scheduler it ran as blocks of 24-30 VALU
    // instructions with the matrix pipe idle behind them.
    Frag f0, f1;
    using K0 = std::integral_constant<int, 0>;
    using K1 = std::integral_constant<int, 1>;
    constexpr int SG_VALU = 0x002;
    int st = 0;                                                 // ring slot of the stage being multiplied (it % NST)
    auto stage_body = [&](auto PHC, int it) {
        constexpr int PH = decltype(PHC)::value;
        using SET = std::integral_constant<int, (PH + 2) % NSET>;
        const int st1 = st + 1 == NST ? 0 : st + 1, st2 = st1 + 1 == NST ? 0 : st1 + 1;
        commit(SET{}, st2, it + 2, K0{});
        fetch(f1, st, K1{});
        mfma_step(f0);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            __builtin_amdgcn_sched_group_barrier(SG_MFMA, 1, 0);
            __builtin_amdgcn_sched_group_barrier(SG_DS_RD, 1, 0);
            __builtin_amdgcn_sched_group_barrier(SG_VALU, VPM, 0);
            if ((q & 3) == 3) __builtin_amdgcn_sched_group_barrier(SG_DS_WR, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        issue(SET{}, it + 2 + NSET, K0{});
        commit(SET{}, st2, it + 2, K1{});
        issue(SET{}, it + 2 + NSET, K1{});
        fetch(f0, st1, K0{});
        mfma_step(f1);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            __builtin_amdgcn_sched_group_barrier(SG_MFMA, 1, 0);
            __builtin_amdgcn_sched_group_barrier(SG_DS_RD, 1, 0);
            __builtin_amdgcn_sched_group_barrier(SG_VALU, VPM, 0);
            if ((q & 3) == 3) __builtin_amdgcn_sched_group_barrier(SG_DS_WR, 1, 0);
            if ((q & 3) == 1) __builtin_amdgcn_sched_group_barrier(SG_VMEM_RD, BLN ? 4 : 2, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        st = st1;
    };

    // prologue: stages 0 and 1 into the ring, stages 2 .. NSET+1 in flight -- requested in the order the loop keeps
    // them in (oldest first: the set the first body commits), so that the compiler's vmcnt bookkeeping at the loop
    // head is the steady state's and not a conservative vmcnt(0)
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    using S2 = std::integral_constant<int, 2>;
    issue(S0{}, 0, K0{}); issue(S0{}, 0, K1{});
    issue(S1{}, 1, K0{}); issue(S1{}, 1, K1{});
    if constexpr (NSET == 3) { issue(S2{}, 2, K0{}); issue(S2{}, 2, K1{}); }
    commit(S0{}, 0, 0, K0{}); commit(S0{}, 0, 0, K1{});
    issue(S0{}, NSET == 2 ? 2 : 3, K0{}); issue(S0{}, NSET == 2 ? 2 : 3, K1{});
    commit(S1{}, 1, 1, K0{}); commit(S1{}, 1, 1, K1{});
    issue(S1{}, NSET == 2 ? 3 : 4, K0{}); issue(S1{}, NSET == 2 ? 3 : 4, K1{});
    __syncthreads();
    fetch(f0, 0, K0{});
    for (int it = 0; it < n_it; it += NSET) {
        stage_body(std::integral_constant<int, 0>{}, it);
        stage_body(std::integral_constant<int, 1>{}, it + 1);
        if constexpr (NSET >= 3) stage_body(std::integral_constant<int, 2>{}, it + 2);
    }

    const bool det = p.det_dw != nullptr;                       // deterministic mode: see TnArgsB
    float* dwo = det ? p.det_dw + (int64_t)split * p.N * p.K : p.dW;
    // D[i = n][j = k]: lane (k = lq, hh), reg r -> dW[n = crow(r,hh)][k]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wn * 128 + i * 32 + crow(r, hh);
                const int k = k0 + wk * 128 + jj * 32 + lq;
                HWGAT_TN_ACC(det, dwo, (int64_t)n * p.K + k, acc[i][jj][r]);
            }
    if (want_db) {
        float* red = reinterpret_cast<float*>(sm);              // [8][256] partial column sums
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) red[lrow * BT + lc8 + e] = colsum[e];
        __syncthreads();
        if (tid < BT) {
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) s += red[q * BT + tid];
            HWGAT_TN_ACC(det, det ? p.det_db + (int64_t)split * p.N : p.db, n0 + tid, s);
        }
    }
}

}  // namespace

int hwgat_launch_tn256_bf16(TnArgsB a, hipStream_t st) {
    if (a.N % BT || a.K % BT || a.M % TMB) return HWGAT_ESHAPE;          // whole 32-row stages
    const int n_tiles = (a.N / BT) * (a.K / BT);
    // equal-sized blocks, one resident per CU: n_split * n_tiles an exact multiple of 256 where the tile count allows,
    // the split count a multiple of 8 (split s lives on XCD s % 8) -- the rule of hwgat_launch_tn256 (gemm_f32_tn256.hip)
    auto gcd = [](int x, int y) { while (y) { int t = x % y; x = y; y = t; } return x; };
    const int r_min = n_tiles / gcd(n_tiles, 256);
    static const int min_rounds = [] { const char* e = lab_env("HWGAT_TN_ROUNDS"); return e ? atoi(e) : 1; }();
    int r = r_min;
    while (r < min_rounds) r += r_min;
    int64_t want = (int64_t)256 * r / n_tiles;
    const int64_t max_split = a.M / (TMB * 16) > 0 ? a.M / (TMB * 16) : 1;
    if (want > max_split) want = max_split;
    if (want < 1) want = 1;
    int64_t rows = (a.M + want - 1) / want;
    rows = (rows + TMB - 1) / TMB * TMB;
    a.n_split = (int)((a.M + rows - 1) / rows);
    a.rows_per_split = rows;
    if (a.det_dw && a.n_split > a.det_cap) return HWGAT_ESHAPE;
    const int grid = ((a.n_split + 7) / 8) * 8 * n_tiles;
    const bool drop = a.pro_p > 0.f, ln = a.mean != nullptr;
    if (drop && ln) return HWGAT_ESHAPE;                         // not used by the model: the 128x128 kernel takes it
    static const int nset = [] { const char* e = lab_env("HWGAT_TN_NSET"); return e ? atoi(e) : 0; }();   // lab switch
    if (drop) {
        if (nset == 3) gemm_tn256_bf16_k<PRO_DROP, false, 3><<<grid, 256, 0, st>>>(a);
        else gemm_tn256_bf16_k<PRO_DROP, false, 2><<<grid, 256, 0, st>>>(a);
    } else if (ln) {
        if (nset == 2) gemm_tn256_bf16_k<PRO_NONE, true, 2><<<grid, 256, 0, st>>>(a);
        else gemm_tn256_bf16_k<PRO_NONE, true, 3><<<grid, 256, 0, st>>>(a);
    } else {
        if (nset == 2) gemm_tn256_bf16_k<PRO_NONE, false, 2><<<grid, 256, 0, st>>>(a);
        else gemm_tn256_bf16_k<PRO_NONE, false, 3><<<grid, 256, 0, st>>>(a);
    }
    HWGAT_LAUNCH_CHECK();
}
