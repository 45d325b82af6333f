// bf16 weight-gradient GEMM for HWGAT on gfx950 (BASELINE config 3), round-3 form:
//   dW[N,K] (fp32) += A[M,N]^T . B[M,K] over an M slice,  db[N] += colsum(A)      (A = gradient, B = layer input; both bf16)
// i.e. the autograd of every nn.Linear of hwgat/models/HWGATE.py:77-79,125-127 with respect to its parameters.
//
// Same machinery as gemm_bf16_nt8w.hip (read that header first): 256x256 dW tile per block, EIGHT waves (two per SIMD) in
// a ping-pong offset by one barrier, both operands HBM -> LDS by LDS-DMA behind counted `s_waitcnt vmcnt(6)`, raw
// s_barrier, 16 x v_mfma_f32_16x16x32_bf16 per phase.  What differs:
//   * the reduction index m is the ROW index of both operands, so an MFMA fragment (8 consecutive m of one column) is a
//     transposed read: 2 x ds_read_b64_tr_b16 from the row-major LDS image, which the DMA fills in whole 512-byte rows
//     (a piece = 2 rows x 256 columns).  A 32-lane half of such a read touches 8 rows x 32 bytes; rows are 512 bytes
//     apart (same banks), so the 32-byte column blocks of row r are XOR-permuted by f(r) = (r&3) | ((r>>3)&1)<<2 -- on
//     the DMA's per-lane SOURCE address, the LDS write itself is linear -- which makes every read conflict free;
//   * a phase is one half (32 rows = one MFMA depth) of a 64-row step against one half of the wave's eight column
//     tiles: fragments of a phase are 4 + 4 operands (32 registers), all sixteen MFMAs distinct accumulators;
//   * one long main loop per block (its M slice), then the 256x256 fp32 tile is added to dW with global atomics
//     (consecutive lanes = consecutive k); the bias gradient falls out of the A fragments (v_dot2 against ones),
//     the eight column tiles dealt over the four waves of a group, in the blocks that own k-tile 0.
// Needs N % 256 == K % 256 == 0, M % 128 == 0 and plain operands (no dropout mask on A, no LayerNorm on B: the
// LayerNorm backward writes LN(x) for this launch, hwgat_ln_bwd_xn); everything else stays on gemm_tn256_bf16_k /
// gemm_tn_bf16_k (hwgat_linear_tn_bf16 decides).
#include <type_traits>
#include "common.h"
#include "fused_ops.h"
#include "gemm_bf16.h"

namespace {

constexpr int BT = 256, BM = 64;              // dW tile edge; rows of both operands per K-tile (two MFMA depths)
constexpr int ROWB = 2 * BT;                  // bytes of one LDS row (256 columns of one operand row)
constexpr int OPB = BM * ROWB;                // one operand tile: 32 KiB
constexpr int BUFB = 2 * OPB;                 // A | B of one 64-row step: 64 KiB
constexpr int SMEM = 2 * BUFB;

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wait_lds() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void wg_barrier() {          // raw: no implicit vmcnt(0), the DMA queue survives it
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

// SLAB: the block's 256x256 fp32 tile goes to a workspace slab instead of global atomics, lane-linear (every store
// instruction is 1 KiB contiguous): ws[((split * n_tiles + tile) * 8 + wave) * 32 + a * 4 + c][lane] = acc[a][c]; dw_reduce_k
// adds the slabs of a tile in split order.  64 MB of memory-side atomics per launch (~49 us at the ~1.3 TB/s they sustain)
// become 64 MB of streaming stores + 64 MB of streaming reads, and the sum's order is fixed: dW is bit-reproducible.
template <bool SLAB>
__global__ __launch_bounds__(512, 2) void gemm_tn8w_bf16_k(TnArgsB p, float* __restrict__ ws) {
    __shared__ __attribute__((aligned(16))) unsigned char smb[SMEM];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int gm = wave >> 2, wk = wave & 3;               // ping-pong group = half of the tile's n range; quarter of its k range
    const int fr = lane & 15, fq = lane >> 4;
    const int tiles_k = p.K / BT, n_tiles = (p.N / BT) * tiles_k;
    const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int tile = jb % n_tiles;
    const int split = (jb / n_tiles) * 8 + xcd;            // the tiles of one M slice run on one XCD: the slice streams from HBM once
    if (split >= p.n_split) return;
    const int n0 = (tile / tiles_k) * BT, k0 = (tile % tiles_k) * BT;
    const int64_t r_begin = (int64_t)split * p.rows_per_split;
    const int64_t r_end = r_begin + p.rows_per_split < p.M ? r_begin + p.rows_per_split : p.M;
    if (r_begin >= r_end) return;
    const int n_it = (int)((r_end - r_begin) / (2 * BM));   // main-loop iterations: two 64-row steps each
    const int sa = p.N * 2, sb = p.K * 2;                   // row strides in bytes

    // ---- LDS-DMA staging.  A stage = 32 rows of one operand = 16 pieces of 2 rows x 512 B; this wave issues pieces
    // `wave` and `wave + 8` (rows 2 wave, 2 wave + 1 and 16 more).  Lane l of a piece writes LDS chunk l (row l>>5, 16-byte
    // chunk l&31) and fetches the chunk that belongs there: 32-byte block ((l&31)>>1) ^ f(row), f as in the header; for
    // both of this wave's pieces f = 2 (wave&1) | 4 ((wave>>2)&1) | (l>>5).
    const int fpiece = 2 * (wave & 1) + 4 * ((wave >> 2) & 1) + (lane >> 5);
    const int chunk = ((((lane & 31) >> 1) ^ fpiece) << 5) + ((lane & 1) << 4);
    const int voff_a = (lane >> 5) * sa + chunk, voff_b = (lane >> 5) * sb + chunk;
    auto stage = [&](int buf, int op, int h, __amdgpu_buffer_rsrc_t rs, int row0) {   // rows row0 + h*32 .. +31 of the slice
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int r = h * 32 + 2 * (wave + 8 * j);      // row within the 64-row step
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(smb + buf * BUFB + op * OPB + r * ROWB), 16,
                                                     op ? voff_b : voff_a, (row0 + r) * (op ? sb : sa), 0, 0);
        }
    };
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + r_begin * p.N + n0), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)(p.B + r_begin * p.K + k0), 0, 0x7fffffff, 0x00020000);

    // ---- fragment reads.  ds_read_b64_tr_b16: lane 4q + p of a 16-lane group addresses row q, columns 4p..4p+3 of a
    // 4-row x 16-column block and receives column (lane & 15) of its four rows.  Fragment (column tile c, depth half kc) of
    // lane group fq: rows kc*32 + fq*8 + {0..3} and + {4..7}.  Row r keeps its 32-byte block b at b ^ f(r); here
    // f = (fr>>2) | (fq&1)<<2 for every row this lane addresses.
    const int q = fr >> 2, pp = fr & 3;
    const int flane = q | ((fq & 1) << 2);
    const int lrow = (fq * 8 + q) * ROWB + 8 * pp;
    int adr_a[8], adr_b[4];                                 // per column tile: the XOR makes the block offset lane dependent
#pragma unroll
    for (int a = 0; a < 8; ++a) adr_a[a] = lrow + (((gm * 8 + a) ^ flane) << 5);
#pragma unroll
    for (int c = 0; c < 4; ++c) adr_b[c] = OPB + lrow + (((wk * 4 + c) ^ flane) << 5);
    auto tr_frag = [&](int off) {
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(smb + off));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(smb + off + 4 * ROWB));
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };

    f32x4 acc[8][4];
    bf16x8 af[4], bfr[4];
    float dbacc[2] = {0.f, 0.f};
    const bool want_db = p.db != nullptr && k0 == 0;
    const bf16x2_t ones = {(__bf16)1.0f, (__bf16)1.0f};
    auto read_a = [&](int buf, int kc, int half) {
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = tr_frag(buf * BUFB + kc * (32 * ROWB) + adr_a[half * 4 + i]);
    };
    auto read_b = [&](int buf, int kc) {
#pragma unroll
        for (int c = 0; c < 4; ++c) bfr[c] = tr_frag(buf * BUFB + kc * (32 * ROWB) + adr_b[c]);
    };
    auto dot8 = [&](const bf16x8& f, float s) {
        s = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(f, f, 0, 1), ones, s, false);
        s = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(f, f, 2, 3), ones, s, false);
        s = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(f, f, 4, 5), ones, s, false);
        return __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(f, f, 6, 7), ones, s, false);
    };
    // wave wk of a group sums column tile half*4 + wk of every phase (dbacc[half]); uniform branches, static registers
    auto colsum = [&](int half) {
        if (want_db) {
            if (wk == 0) dbacc[half] = dot8(af[0], dbacc[half]);
            else if (wk == 1) dbacc[half] = dot8(af[1], dbacc[half]);
            else if (wk == 2) dbacc[half] = dot8(af[2], dbacc[half]);
            else dbacc[half] = dot8(af[3], dbacc[half]);
        }
    };
    auto mfma16 = [&](int half) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c)
                acc[half * 4 + i][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[c], acc[half * 4 + i][c], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };

    // ---- prologue: what phases 3..8 of a previous iteration would have issued for the first two 64-row steps
    stage(0, 1, 0, rb, 0); stage(0, 0, 0, ra, 0); stage(0, 1, 1, rb, 0); stage(0, 0, 1, ra, 0);
    stage(1, 1, 0, rb, BM); stage(1, 0, 0, ra, BM);
    wait_vm<4>();
    wg_barrier();
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[i][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (gm == 1) wg_barrier();                              // waves 4-7 run one barrier behind waves 0-3

    for (int it = 0; it < n_it; ++it) {
        const bool more = it + 1 < n_it;
        const int rc = (2 * it + 1) * BM, rn = (2 * it + 2) * BM;   // first row of this iteration's second step / of the next pair
        // one phase = { fragment reads + one DMA stage + counted wait | barrier | 16 MFMAs | barrier }; a stage issued in
        // phase q is first read in phase q + 5 and retired by the wait of phase q + 3 (see gemm_bf16_nt8w.hip)
#define HWGAT_WAIT(NLAST) do { if (more) wait_vm<6>(); else wait_vm<NLAST>(); } while (0)
        // phases 1-4: step A (buffer 0)
        read_a(0, 0, 0); read_b(0, 0);
        stage(1, 1, 1, rb, rc);
        HWGAT_WAIT(6);
        wg_barrier(); wait_lds(); __builtin_amdgcn_sched_barrier(0); colsum(0); mfma16(0); wg_barrier();
        read_a(0, 0, 1);
        stage(1, 0, 1, ra, rc);
        HWGAT_WAIT(6);
        wg_barrier(); wait_lds(); __builtin_amdgcn_sched_barrier(0); colsum(1); mfma16(1); wg_barrier();
        read_a(0, 1, 0); read_b(0, 1);
        if (more) stage(0, 1, 0, rb, rn);
        HWGAT_WAIT(4);
        wg_barrier(); wait_lds(); __builtin_amdgcn_sched_barrier(0); colsum(0); mfma16(0); wg_barrier();
        read_a(0, 1, 1);
        if (more) stage(0, 0, 0, ra, rn);
        HWGAT_WAIT(2);
        wg_barrier(); wait_lds(); __builtin_amdgcn_sched_barrier(0); colsum(1); mfma16(1); wg_barrier();
        // phases 5-8: step B (buffer 1)
        read_a(1, 0, 0); read_b(1, 0);
        if (more) stage(0, 1, 1, rb, rn);
        HWGAT_WAIT(0);
        wg_barrier(); wait_lds(); __builtin_amdgcn_sched_barrier(0); colsum(0); mfma16(0); wg_barrier();
        read_a(1, 0, 1);
        if (more) stage(0, 0, 1, ra, rn);
        HWGAT_WAIT(0);
        wg_barrier(); wait_lds(); __builtin_amdgcn_sched_barrier(0); colsum(1); mfma16(1); wg_barrier();
        read_a(1, 1, 0); read_b(1, 1);
        if (more) stage(1, 1, 0, rb, rn + BM);
        HWGAT_WAIT(0);
        wg_barrier(); wait_lds(); __builtin_amdgcn_sched_barrier(0); colsum(0); mfma16(0); wg_barrier();
        read_a(1, 1, 1);
        if (more) stage(1, 0, 0, ra, rn + BM);
        HWGAT_WAIT(0);
        wg_barrier(); wait_lds(); __builtin_amdgcn_sched_barrier(0); colsum(1); mfma16(1); wg_barrier();
#undef HWGAT_WAIT
    }
    if (gm == 0) wg_barrier();                              // matches the partner's last barrier (equal barrier counts)

    // ---- epilogue: acc[a][c][r] is dW[n0 + gm*128 + a*16 + fq*4 + r][k0 + wk*64 + c*16 + fr]
    if constexpr (SLAB) {
        float* slab = ws + ((((int64_t)split * n_tiles + tile) * 8 + wave) * 32) * 256 + lane * 4;
#pragma unroll
        for (int a = 0; a < 8; ++a)
#pragma unroll
            for (int c = 0; c < 4; ++c) *reinterpret_cast<f32x4*>(slab + (a * 4 + c) * 256) = acc[a][c];
    } else {
        const bool det = p.det_dw != nullptr;                   // deterministic mode: see TnArgsB
        float* dst = (det ? p.det_dw + (int64_t)split * p.N * p.K : p.dW) + (int64_t)(n0 + gm * 128 + fq * 4) * p.K + k0 + wk * 64 + fr;
#pragma unroll
        for (int a = 0; a < 8; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) HWGAT_TN_ACC(det, dst, (int64_t)(a * 16 + r) * p.K + c * 16, acc[a][c][r]);
    }
    if (want_db) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            float s = dbacc[e];
            s += __shfl_xor(s, 16, 64);
            s += __shfl_xor(s, 32, 64);
            if (fq == 0) HWGAT_TN_ACC(p.det_dw != nullptr, p.det_dw ? p.det_db + (int64_t)split * p.N : p.db, n0 + gm * 128 + (e * 4 + wk) * 16 + fr, s);
        }
    }
}

// dW tile += the slabs of its M splits, in split order.  One thread per slab float (a tile = 65 536 floats = 256 blocks: with
// one 16-byte slot per thread a single-tile launch had 64 blocks and read its 64 MB at a quarter of the chip's bandwidth):
// float idx of a slab = (((wave*8 + a)*4 + c)*64 + lane)*4 + r.
__global__ __launch_bounds__(256) void dw_reduce_k(const float* __restrict__ ws, float* __restrict__ dW, int n_split,
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
    const int r = idx & 3, slot = idx >> 2;
    const int lane = slot & 63, c = (slot >> 6) & 3, a = (slot >> 8) & 7, wave = slot >> 11;
    const int gm = wave >> 2, wk = wave & 3, fr = lane & 15, fq = lane >> 4;
    const int n0 = (tile / tiles_k) * BT, k0 = (tile % tiles_k) * BT;
    dW[(int64_t)(n0 + gm * 128 + a * 16 + fq * 4 + r) * K + k0 + wk * 64 + c * 16 + fr] += s;
}

}  // namespace

// the M split of a shape: one round of equal blocks, one per CU: the split count a multiple of 8 (split s lives on XCD s % 8,
// the tiles of a slice share its rows through that XCD's L2) with splits x tiles <= 256; slices are whole 128-row iterations
static void tn8w_split(int64_t M, int n_tiles, int64_t* rows_per_split, int* n_split) {
    int want = 256 / n_tiles / 8 * 8;
    if (want < 8) want = 8;
    const int64_t its = M / (2 * BM);
    if (want > its) want = (int)its;
    const int64_t per = (its + want - 1) / want;
    *rows_per_split = per * (2 * BM);
    *n_split = (int)((its + per - 1) / per);
}

// shapes this kernel takes; everything else (incl. a slice too long for the 32-bit DMA offsets) goes to the older kernels
bool hwgat_tn8w_bf16_takes(int64_t M, int N, int K, float pro_p, const float* mean) {
    if (!(N % BT == 0 && K % BT == 0 && M % (2 * BM) == 0 && M >= 2 * BM && pro_p == 0.f && mean == nullptr)) return false;
    int64_t rows;
    int n_split;
    tn8w_split(M, (N / BT) * (K / BT), &rows, &n_split);
    return rows * (N > K ? N : K) * 2 <= 0x7fffffff;             // 32-bit DMA offsets within a slice
}

// floats of workspace the slab form needs for this shape (every block's 256x256 tile), 0 if the kernel does not take it
int64_t hwgat_tn8w_bf16_ws_floats(int64_t M, int N, int K) {
    if (!hwgat_tn8w_bf16_takes(M, N, K, 0.f, nullptr)) return 0;
    const int n_tiles = (N / BT) * (K / BT);
    int want = 256 / n_tiles / 8 * 8;
    if (want < 8) want = 8;
    return (int64_t)want * n_tiles * 65536;
}

int hwgat_launch_tn8w_bf16(TnArgsB a, hipStream_t st, float* ws) {
    if (!hwgat_tn8w_bf16_takes(a.M, a.N, a.K, a.pro_p, a.mean)) return HWGAT_ESHAPE;
    const int n_tiles = (a.N / BT) * (a.K / BT);
    int64_t rows;
    tn8w_split(a.M, n_tiles, &rows, &a.n_split);
    a.rows_per_split = rows;
    if (a.det_dw) {                                              // deterministic mode: plain images, never the slabs
        if (a.n_split > a.det_cap) return HWGAT_ESHAPE;
        ws = nullptr;
    }
    const int grid = ((a.n_split + 7) / 8) * 8 * n_tiles;
    if (ws) {
        gemm_tn8w_bf16_k<true><<<grid, 512, 0, st>>>(a, ws);
        dw_reduce_k<<<n_tiles * 256, 256, 0, st>>>(ws, a.dW, a.n_split, n_tiles, a.K / BT, a.K);
    } else {
        gemm_tn8w_bf16_k<false><<<grid, 512, 0, st>>>(a, nullptr);
    }
    HWGAT_LAUNCH_CHECK();
}
