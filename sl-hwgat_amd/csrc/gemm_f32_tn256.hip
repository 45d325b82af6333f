// fp32 weight-gradient GEMM with 128x128 WAVE tiles for HWGAT on gfx950: the large-output companion
// of gemm_tn_k (gemm_f32.hip), same semantics (dW += A^T B over an M slice, db += colsum(A), dropout
// mask on A, LayerNorm on B).
//
// Structure (what the vendor's best fp32 kernel does, per its disassembly -- DESIGN.md section 5):
//   * 256x256 dW tile per block, 4 waves x (128x128) = 4x4 v_mfma_f32_32x32x2_f32 tiles, the 256
//     accumulator registers live in AGPRs, ONE wave per SIMD;
//   * 16-row stages of both operands in a 3-deep LDS ring; the stage two ahead is loaded to registers
//     at the top of an iteration and committed in the second half; one barrier per 128 MFMAs;
//   * the steady-state loop body is branch-free straight-line code whose instruction order is PINNED
//     with sched_group_barrier: 4 MFMAs, then one LDS read / global load / LDS write, repeated --
//     never a cluster of memory instructions with a wait, which is what idles a lone wave's MFMA pipe.
#include <stdlib.h>
#include <type_traits>
#include "common.h"
#include "fused_ops.h"
#include "gemm_f32.h"

namespace {

constexpr int BT = 256, TM = 16, NST = 3;
constexpr int STG = 2 * TM * BT;                                // floats per LDS stage: A[16][256] | B[16][256] = 32 KB

// LLVM SchedGroupMask bits
constexpr int SG_MFMA = 0x008, SG_VMEM_RD = 0x020, SG_DS_RD = 0x100, SG_DS_WR = 0x200;

// pin the instruction order of one 32-MFMA chunk: 8 x { 4 MFMAs, N1 x M1, N2 x M2 }
template <int M1, int N1, int M2, int N2>
__device__ __forceinline__ void il8() {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        __builtin_amdgcn_sched_group_barrier(SG_MFMA, 4, 0);
        if constexpr (N1 > 0) __builtin_amdgcn_sched_group_barrier(M1, N1, 0);
        if constexpr (N2 > 0) __builtin_amdgcn_sched_group_barrier(M2, N2, 0);
    }
}

template <int PRO, bool BLN>
__global__ __launch_bounds__(256, 1) void gemm_tn256_k(TnArgs p) {
    __shared__ __attribute__((aligned(16))) float sm[NST * STG];
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
    const int n_it = (int)((r_end - r_begin) / TM);

    const int lrow = tid >> 6, lc4 = (tid & 63) * 4;            // rows lrow + 4*i (i < 4), floats lc4..lc4+3
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
            ra[i] = *reinterpret_cast<const f32x4*>(p.A + (r0 + 4 * i) * p.N + n0 + lc4);
            rb[i] = *reinterpret_cast<const f32x4*>(p.B + (r0 + 4 * i) * p.K + k0 + lc4);
            if constexpr (BLN) { bm[i] = p.mean[r0 + 4 * i]; bs[i] = p.rstd[r0 + 4 * i]; }
        }
    };
    auto commit = [&](int stage, int it) {
        float* As = sm + stage * STG;
        float* Bs = As + TM * BT;
        const int64_t r0 = r_begin + (int64_t)it * TM + lrow;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x4 a = ra[i], b = rb[i];
            if constexpr (PRO == PRO_DROP)
                a *= drop_keep4(p.pro_seed, (uint64_t)(r0 + 4 * i) * p.N + n0 + lc4, pro_th, pro_sc);
            if constexpr (BLN) b = (b - bm[i]) * bs[i] * lg + lb;
            colsum += a;
            *reinterpret_cast<f32x4*>(As + (lrow + 4 * i) * BT + lc4) = a;
            *reinterpret_cast<f32x4*>(Bs + (lrow + 4 * i) * BT + lc4) = b;
        }
    };

    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][jj][e] = 0.f;

    // operand chunk = 2 k2-steps (4 rows of m): 8 A + 8 B registers
    struct Chunk { float a[2][4], b[2][4]; };
    auto fetch = [&](Chunk& c, int stage, int ch) {
        const float* As = sm + stage * STG + wn * 128 + lq;
        const float* Bs = sm + stage * STG + TM * BT + wk * 128 + lq;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int ro = (2 * (2 * ch + e) + hh) * BT;
#pragma unroll
            for (int i = 0; i < 4; ++i) { c.a[e][i] = As[ro + 32 * i]; c.b[e][i] = Bs[ro + 32 * i]; }
        }
    };
    auto mfma_chunk = [&](const Chunk& c) {                      // 32 MFMAs
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
                    acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(c.a[e][i], c.b[e][jj], acc[i][jj], 0, 0, 0);
    };
    // one stage: 4 chunks x 32 MFMAs.  LOAD: stream stage it+2 in (8 global loads in chunk 0, 8 LDS writes in
    // chunk 2).  NEXT: prefetch the first chunk of stage it+1 during chunk 3.
    Chunk c0, c1;
    auto stage_body = [&](int it, auto LOAD, auto NEXT) {
        constexpr bool kLoad = decltype(LOAD)::value, kNext = decltype(NEXT)::value;
        const int st = it % NST;
        if constexpr (kLoad) issue(it + 2);
        fetch(c1, st, 1);
        mfma_chunk(c0);
        il8<SG_DS_RD, 2, SG_VMEM_RD, kLoad ? 1 : 0>();
        __builtin_amdgcn_sched_barrier(0);
        fetch(c0, st, 2);
        mfma_chunk(c1);
        il8<SG_DS_RD, 2, 0, 0>();
        __builtin_amdgcn_sched_barrier(0);
        fetch(c1, st, 3);
        mfma_chunk(c0);
        if constexpr (kLoad) commit((it + 2) % NST, it + 2);
        il8<SG_DS_RD, 2, SG_DS_WR, kLoad ? 1 : 0>();
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (kNext) fetch(c0, (it + 1) % NST, 0);
        mfma_chunk(c1);
        il8<SG_DS_RD, kNext ? 2 : 0, 0, 0>();
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    };
    using T = std::true_type;
    using F = std::false_type;

    // prologue: stages 0 and 1
    issue(0);
    commit(0, 0);
    if (n_it > 1) { issue(1); commit(1, 1); }
    __syncthreads();
    fetch(c0, 0, 0);
    int it = 0;
    for (; it + 2 < n_it; ++it) stage_body(it, T{}, T{});        // steady state: branch-free body
    if (it + 1 < n_it) { stage_body(it, F{}, T{}); ++it; }
    stage_body(it, F{}, F{});

    // D[i = n][j = k]: lane (k = lq, hh), reg r -> dW[n = crow(r,hh)][k]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wn * 128 + i * 32 + crow(r, hh);
                const int k = k0 + wk * 128 + jj * 32 + lq;
                atomicAdd(p.dW + (int64_t)n * p.K + k, acc[i][jj][r]);
            }
    if (p.db != nullptr && k0 == 0) {
        float* red = sm;                                        // [4][256] partial column sums
        __syncthreads();
        *reinterpret_cast<f32x4*>(red + lrow * BT + lc4) = colsum;
        __syncthreads();
        if (tid < BT) atomicAdd(p.db + n0 + tid, red[tid] + red[BT + tid] + red[2 * BT + tid] + red[3 * BT + tid]);
    }
}

}  // namespace

int hwgat_launch_tn256(TnArgs a, hipStream_t st) {
    if (a.N % BT || a.K % BT || a.M % TM) return HWGAT_ESHAPE;
    const int n_tiles = (a.N / BT) * (a.K / BT);
    // equal-sized blocks, one resident per CU (256 slots): blocks = n_split * n_tiles an exact multiple of 256
    auto gcd = [](int x, int y) { while (y) { int t = x % y; x = y; y = t; } return x; };
    const int r_min = n_tiles / gcd(n_tiles, 256);
    // The smallest whole number of rounds that fills every slot with equal blocks -- ONE round when the tile count
    // divides 256.  Every M slice ends in n_tiles x 256 KB of float atomics (memory-side, ~1.3 TB/s chip-wide); the
    // round-1 rule (at least two rounds) doubled that traffic for nothing: stage 2 dWproj 112.5 -> 118.6, dW1 124.3 ->
    // 128.0, dW2 120.0 -> 123.2 TFLOP/s on one box; four rounds 104-119.  The split count has to stay a multiple of 8:
    // split s lives on XCD s % 8 (its tiles share the M slice through that XCD's L2), and 21 splits x 12 tiles put
    // 36 blocks on five of the XCDs' 32 CUs -- twice the time (measured).
    static const int min_rounds = [] { const char* e = getenv("HWGAT_TN_ROUNDS"); return e ? atoi(e) : 1; }();
    int r = r_min;
    while (r < min_rounds) r += r_min;
    int64_t want = (int64_t)256 * r / n_tiles;
    const int64_t max_split = a.M / (TM * 16) > 0 ? a.M / (TM * 16) : 1;
    if (want > max_split) want = max_split;
    if (want < 1) want = 1;
    int64_t rows = (a.M + want - 1) / want;
    rows = (rows + TM - 1) / TM * TM;
    a.n_split = (int)((a.M + rows - 1) / rows);
    a.rows_per_split = rows;
    const int grid = ((a.n_split + 7) / 8) * 8 * n_tiles;
    const bool drop = a.pro_p > 0.f, ln = a.mean != nullptr;
    if (drop && ln) return HWGAT_ESHAPE;                         // not used by the model
    if (drop) gemm_tn256_k<PRO_DROP, false><<<grid, 256, 0, st>>>(a);
    else if (ln) gemm_tn256_k<PRO_NONE, true><<<grid, 256, 0, st>>>(a);
    else gemm_tn256_k<PRO_NONE, false><<<grid, 256, 0, st>>>(a);
    HWGAT_LAUNCH_CHECK();
}
