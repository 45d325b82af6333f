// fp32 weight-gradient GEMM with 128x128 WAVE tiles for HWGAT on gfx950: the large-output companion
// of gemm_tn_k (gemm_f32.hip), same semantics (dW += A^T B over an M slice, db += colsum(A), dropout
// mask on A, LayerNorm on B).
//
// Structure (what the vendor's best fp32 kernel does, per its disassembly -- DESIGN.md section 5):
//   * 256x256 dW tile per block, 4 waves x (128x128) = 4x4 v_mfma_f32_32x32x2_f32 tiles, the 256
//     accumulator registers live in AGPRs, ONE wave per SIMD;
//   * 16-row stages of both operands in a 3-deep LDS ring; the global loads of the stage THREE ahead are issued at
//     the top of an iteration into one of two register sets and committed to the ring in the second half of the
//     NEXT iteration (12 000 matrix-pipe cycles of lead; with one register set and 4 000 cycles the commit waited
//     on HBM: round 2); one barrier per 128 MFMAs;
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

// SLAB: the block's 256x256 partial tile goes to a workspace slab (lane-linear: every store instruction is 1 KB contiguous)
// instead of 64 MB of memory-side float atomics per launch (~49 us at the ~1.3 TB/s they sustain); tn256_reduce_k then adds
// the slabs of a tile in split order, so these weight gradients are also bit-reproducible (gemm_bf16_tn8w.hip did this first).
//   ws[(split * n_tiles + tile) * 65536 + (((wave * 16 + i * 4 + jj) * 4 + q) * 64 + lane) * 4 + e] = acc[i][jj][4 q + e]
template <int PRO, bool BLN, bool SLAB = false>
__global__ __launch_bounds__(256, 1) void gemm_tn256_k(TnArgs p, float* __restrict__ ws = nullptr) {
    HWGAT_RESOLVE_SEED1(p);
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
    f32x4 ra[2][4], rb[2][4];                                   // two staging register sets (stage k uses set k & 1)
    float bm[2][4], bs[2][4];
    f32x4 colsum = {0.f, 0.f, 0.f, 0.f};
    f32x4 lg = {1.f, 1.f, 1.f, 1.f}, lb = {0.f, 0.f, 0.f, 0.f};
    if constexpr (BLN) {
        lg = *reinterpret_cast<const f32x4*>(p.gamma + k0 + lc4);
        lb = *reinterpret_cast<const f32x4*>(p.beta + k0 + lc4);
    }
    auto issue = [&](auto QC, int it) {
        constexpr int Q = decltype(QC)::value;
        const int64_t r0 = r_begin + (int64_t)it * TM + lrow;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ra[Q][i] = *reinterpret_cast<const f32x4*>(p.A + (r0 + 4 * i) * p.N + n0 + lc4);
            rb[Q][i] = *reinterpret_cast<const f32x4*>(p.B + (r0 + 4 * i) * p.K + k0 + lc4);
            if constexpr (BLN) { bm[Q][i] = p.mean[r0 + 4 * i]; bs[Q][i] = p.rstd[r0 + 4 * i]; }
        }
    };
    auto commit = [&](auto QC, int stage, int it, float live) {
        constexpr int Q = decltype(QC)::value;
        float* As = sm + stage * STG;
        float* Bs = As + TM * BT;
        const int64_t r0 = r_begin + (int64_t)it * TM + lrow;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x4 a = ra[Q][i], b = rb[Q][i];
            if constexpr (PRO == PRO_DROP)
                a *= drop_keep4(p.pro_seed, (uint64_t)(r0 + 4 * i) * p.N + n0 + lc4, pro_th, pro_sc);
            if constexpr (BLN) b = (b - bm[Q][i]) * bs[Q][i] * lg + lb;
            colsum += a * live;
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
    // one stage (iteration `it`, parity PAR = it & 1): 4 chunks x 32 MFMAs.
    //   the global loads of stage it+3 go out in chunk 0, into register set PAR^1;
    //   stage it+2 (requested a whole iteration ago, register set PAR) is written to its ring slot in chunk 2;
    //   the first operand chunk of stage it+1 is prefetched during chunk 3.
    // ONE body per parity for every stage: past the end of the M slice the loaders re-read the last stage and the
    // commit lands in a ring slot nobody reads any more (its column-sum contribution is multiplied by zero), so there
    // are no specialised tail bodies -- with ten of them the register allocator spilled 2 000+ registers.
    Chunk c0, c1;
    auto stage_body = [&](auto PARC, int it) {
        constexpr int PAR = decltype(PARC)::value;
        const int st = it % NST;
        issue(std::integral_constant<int, PAR ^ 1>{}, it + 3 < n_it ? it + 3 : n_it - 1);
        fetch(c1, st, 1);
        mfma_chunk(c0);
        il8<SG_DS_RD, 2, SG_VMEM_RD, 1>();
        __builtin_amdgcn_sched_barrier(0);
        fetch(c0, st, 2);
        mfma_chunk(c1);
        il8<SG_DS_RD, 2, 0, 0>();
        __builtin_amdgcn_sched_barrier(0);
        fetch(c1, st, 3);
        mfma_chunk(c0);
        commit(std::integral_constant<int, PAR>{}, (it + 2) % NST, it + 2 < n_it ? it + 2 : n_it - 1, it + 2 < n_it ? 1.0f : 0.0f);
        il8<SG_DS_RD, 2, SG_DS_WR, 1>();
        __builtin_amdgcn_sched_barrier(0);
        fetch(c0, (it + 1) % NST, 0);
        mfma_chunk(c1);
        il8<SG_DS_RD, 2, 0, 0>();
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;

    // prologue: stages 0 and 1 into the ring, stage 2 in flight (register set 0).  n_it is even and >= 2.
    issue(P0{}, 0);
    commit(P0{}, 0, 0, 1.0f);
    issue(P1{}, 1);
    commit(P1{}, 1, 1, 1.0f);
    issue(P0{}, 2 < n_it ? 2 : n_it - 1);
    __syncthreads();
    fetch(c0, 0, 0);
    for (int it = 0; it < n_it; it += 2) {
        stage_body(P0{}, it);
        stage_body(P1{}, it + 1);
    }

    const bool det = p.det_dw != nullptr;                       // deterministic mode: see TnArgs
    float* dwo = det ? p.det_dw + (int64_t)split * p.N * p.K : p.dW;
    // D[i = n][j = k]: lane (k = lq, hh), reg r -> dW[n = crow(r,hh)][k]
    if constexpr (SLAB) {
        float* slab = ws + ((int64_t)split * n_tiles + tile) * 65536 + wave * 16384 + lane * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 v = {acc[i][jj][4 * q], acc[i][jj][4 * q + 1], acc[i][jj][4 * q + 2], acc[i][jj][4 * q + 3]};
                    *reinterpret_cast<f32x4*>(slab + ((i * 4 + jj) * 4 + q) * 256) = v;
                }
    } else {
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
    }
    if (p.db != nullptr && k0 == 0) {
        float* red = sm;                                        // [4][256] partial column sums
        __syncthreads();
        *reinterpret_cast<f32x4*>(red + lrow * BT + lc4) = colsum;
        __syncthreads();
        if (tid < BT) HWGAT_TN_ACC(det, det ? p.det_db + (int64_t)split * p.N : p.db, n0 + tid, red[tid] + red[BT + tid] + red[2 * BT + tid] + red[3 * BT + tid]);
    }
}

// dW tile += the slabs of its M splits, in split order; one thread per slab float (256 workgroups per tile)
__global__ __launch_bounds__(256) void tn256_reduce_k(const float* __restrict__ ws, float* __restrict__ dW, int n_split,
                                                      int n_tiles, int tiles_k, int K) {
    const int tile = blockIdx.x >> 8;
    const int idx = (blockIdx.x & 255) * 256 + threadIdx.x;
    const float* src = ws + (int64_t)tile * 65536 + idx;
    const int64_t step = (int64_t)n_tiles * 65536;
    float s = 0.f;
    int sp = 0;
    for (; sp + 8 <= n_split; sp += 8) {                       // eight loads in flight, added in split order
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = src[(sp + j) * step];
#pragma unroll
        for (int j = 0; j < 8; ++j) s += v[j];
    }
    for (; sp < n_split; ++sp) s += src[sp * step];
    const int e = idx & 3, lane = (idx >> 2) & 63, q = (idx >> 8) & 3, t = (idx >> 10) & 15, wave = idx >> 14;
    const int i = t >> 2, jj = t & 3, lq = lane & 31, hh = lane >> 5, wn = wave >> 1, wk = wave & 1;
    const int n0 = (tile / tiles_k) * BT, k0 = (tile % tiles_k) * BT;
    dW[(int64_t)(n0 + wn * 128 + i * 32 + crow(4 * q + e, hh)) * K + k0 + wk * 128 + jj * 32 + lq] += s;
}

}  // namespace

// the M split of a launch: equal-sized blocks, one resident per CU (256 slots)
static void tn256_split(int64_t M, int n_tiles, int& n_split, int64_t& rows_per_split) {
    // blocks = n_split * n_tiles an exact multiple of 256
    auto gcd = [](int x, int y) { while (y) { int t = x % y; x = y; y = t; } return x; };
    const int r_min = n_tiles / gcd(n_tiles, 256);
    // The smallest whole number of rounds that fills every slot with equal blocks -- ONE round when the tile count
    // divides 256.  Every M slice ends in n_tiles x 256 KB of partial sums; the round-1 rule (at least two rounds) doubled
    // that traffic for nothing: stage 2 dWproj 112.5 -> 118.6, dW1 124.3 -> 128.0, dW2 120.0 -> 123.2 TFLOP/s on one box;
    // four rounds 104-119.  The split count has to stay a multiple of 8: split s lives on XCD s % 8 (its tiles share the
    // M slice through that XCD's L2), and 21 splits x 12 tiles put 36 blocks on five of the XCDs' 32 CUs -- twice the
    // time (measured).
    static const int min_rounds = [] { const char* e = lab_env("HWGAT_TN_ROUNDS"); return e ? atoi(e) : 1; }();
    int r = r_min;
    while (r < min_rounds) r += r_min;
    int64_t want = (int64_t)256 * r / n_tiles;
    const int64_t max_split = M / (TM * 16) > 0 ? M / (TM * 16) : 1;
    if (want > max_split) want = max_split;
    if (want < 1) want = 1;
    int64_t rows = (M + want - 1) / want;
    rows = (rows + 2 * TM - 1) / (2 * TM) * (2 * TM);
    n_split = (int)((M + rows - 1) / rows);
    rows_per_split = rows;
}

// floats of workspace hwgat_launch_tn256 wants for the slab form of this shape (0: the shape does not take the kernel)
int64_t hwgat_tn256_ws_floats(int64_t M, int N, int K) {
    if (N % BT || K % BT || M % (2 * TM)) return 0;
    const int n_tiles = (N / BT) * (K / BT);
    int n_split;
    int64_t rows;
    tn256_split(M, n_tiles, n_split, rows);
    return (int64_t)n_split * n_tiles * 65536;
}

int hwgat_launch_tn256(TnArgs a, hipStream_t st, float* ws, int64_t ws_floats) {
    if (a.N % BT || a.K % BT || a.M % (2 * TM)) return HWGAT_ESHAPE;     // an even number of 16-row stages per M slice
    const int n_tiles = (a.N / BT) * (a.K / BT);
    tn256_split(a.M, n_tiles, a.n_split, a.rows_per_split);
    if (a.det_dw) {                                              // deterministic mode: plain images, never the slabs
        if (a.n_split > a.det_cap) return HWGAT_ESHAPE;
        ws = nullptr;
    }
    const int grid = ((a.n_split + 7) / 8) * 8 * n_tiles;
    const bool drop = a.pro_p > 0.f, ln = a.mean != nullptr;
    if (drop && ln) return HWGAT_ESHAPE;                         // not used by the model
    if (ws && ws_floats >= (int64_t)a.n_split * n_tiles * 65536) {
        if (drop) gemm_tn256_k<PRO_DROP, false, true><<<grid, 256, 0, st>>>(a, ws);
        else if (ln) gemm_tn256_k<PRO_NONE, true, true><<<grid, 256, 0, st>>>(a, ws);
        else gemm_tn256_k<PRO_NONE, false, true><<<grid, 256, 0, st>>>(a, ws);
        tn256_reduce_k<<<n_tiles * 256, 256, 0, st>>>(ws, a.dW, a.n_split, n_tiles, a.K / BT, a.K);
        HWGAT_LAUNCH_CHECK();
    }
    if (drop) gemm_tn256_k<PRO_DROP, false><<<grid, 256, 0, st>>>(a);
    else if (ln) gemm_tn256_k<PRO_NONE, true><<<grid, 256, 0, st>>>(a);
    else gemm_tn256_k<PRO_NONE, false><<<grid, 256, 0, st>>>(a);
    HWGAT_LAUNCH_CHECK();
}
