// bf16 NT linear GEMM for HWGAT on gfx950 (BASELINE config 3), round-3 main loop:
//   C[M,N] = A[M,K] . W[N,K]^T (+ fused epilogue), A / W / C / C2 / res / aux bf16, bias / LayerNorm scalars fp32,
// v_mfma_f32_16x16x32_bf16 with fp32 accumulation (reference hwgat/models/HWGATE.py:86,115-116,131-135,203,217,219).
//
// Why another kernel.  gemm_nt256_bf16_k (one wave per SIMD, operands HBM/L2 -> VGPR -> ds_write_b128 -> LDS) tops out
// at ~890 TFLOP/s on the plain launches, and its fused launches pay every per-element vector instruction of loaders and
// epilogues at the single-wave issue rate with the matrix pipe idle (LABLOG.md section 8).  This one changes both:
//   * operands go HBM/L2 -> LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`): no staging registers, no ds_write pass, the
//     loads of the next five phases stay in flight across barriers behind COUNTED `s_waitcnt vmcnt(6)` (never 0 in the
//     steady state), workgroup barriers are raw `s_barrier` (a __syncthreads() would drain the DMA queue);
//   * EIGHT waves per 256x256 tile, two per SIMD, in a ping-pong: waves 0-3 (columns 0-127 of the tile) and waves 4-7
//     (columns 128-255) are offset by one barrier, so while one wave of a SIMD runs its 16-MFMA cluster its partner issues
//     its LDS fragment reads and its share of the DMA for a later K-tile;
//   * the epilogue needs NO LDS and every global access of it is whole lines: MFMA B-operand column j = lane & 15 is
//     weight row  n = (lane & 15) * 8 + t  for the wave's eight 16-column tiles t, so a lane ends with 8 CONSECUTIVE
//     output columns (16 bytes of bf16) of each of its rows, 16 neighbouring lanes = 256 contiguous bytes, four rows per
//     instruction; the residual / auxiliary operand is read in the same shape.  (Round-3 history: a transposed product
//     with direct stores issued 64 separate 16-byte requests per instruction, 6.7 us of store tail per tile; turning the
//     accumulators through wave-private LDS strips fixed the requests but cost 256 KiB of ds_write_b128 per tile at
//     ~80 B/clk/CU -- as long.  The weight-row deal makes the turn unnecessary.)
//
// LDS image of one K-tile (BK = 64): activations [256 rows][128 B] | weights [256 rows][128 B], 16-byte chunks of a row
// XOR-swizzled by (row & 7) so that every ds_read_b128 fragment read is bank-conflict free; the DMA writes LDS linearly
// (wave base + lane x 16 B), so swizzle AND the weight-row deal are applied to the per-lane SOURCE address (a piece =
// 8 rows x 128 B, whole lines).  Weight rows sit tile-major: LDS row (n>>7)*128 + (n&7)*16 + ((n>>3)&15) holds weight row
// n of the tile, i.e. the 16 rows of one MFMA tile are consecutive.  Two K-tile buffers (128 KiB) + 4 KiB row statistics.
//
// Needs M % 256 == N % 256 == K % 128 == 0, no A-side prologue (PRO_NONE or the folded LayerNorm); everything else stays
// on gemm_nt256_bf16_k / gemm_nt_bf16_k (hwgat_linear_nt_bf16 decides).
#include <type_traits>
#include "common.h"
#include "fused_ops.h"
#include "gemm_bf16.h"

namespace {

constexpr int BT = 256, BK = 64;
constexpr int ROWB = 2 * BK;                  // bytes of one LDS row (one K-tile of one matrix row)
constexpr int OPB = BT * ROWB;                // one operand tile: 32 KiB
constexpr int BUFB = 2 * OPB;                 // activations | weights of one K-tile: 64 KiB
constexpr int STAT_OFF = 2 * BUFB;            // row statistics: [2 column halves][256 rows] x (sum, sum of squares)
constexpr int SMEM = STAT_OFF + 2 * BT * 2 * 4;

typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wait_lds() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void wg_barrier() {          // raw: no implicit vmcnt(0), the DMA queue survives it
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

// STAT: 0 = plain epilogue, 1 = + row statistics, 2 = + row statistics and the merged store, 3 = folded LayerNorm.
// DBG (lab builds only, wrong results by design): 1 = no epilogue stores, 2 = no DMA waits, 3 = no DMA at all,
// 4 = no DMA and no fragment reads -- what each part of the schedule costs (tools/nt8w_lab.py).
template <int EPI, int STAT, int DBG = 0>
__global__ __launch_bounds__(512, 2) void gemm_nt8w_bf16_k(NtArgsB p) {
    HWGAT_RESOLVE_SEEDS2(p);
    __shared__ __attribute__((aligned(16))) unsigned char smb[SMEM];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int gm = wave >> 2, wm = wave & 3;               // ping-pong group = column half of the tile; row quarter
    const int fr = lane & 15, fq = lane >> 4;
    const int tiles_n = p.N / BT;
    const int row_blocks = (int)(p.M / BT);
    const int n_tiles = row_blocks * tiles_n;
    const int n_it = p.K / (2 * BK);                        // main-loop iterations per tile: two K-tiles each
    const int K2 = p.K * 2;                                 // bytes per matrix row

    // ---- tile order: XCD-aware (the n-tiles of a 256-row block run on one XCD at the same time), see gemm_nt_k.
    // (Round 3 also tried contiguous runs -- one block takes all column tiles of a row block one after the other: the
    // re-reads then MISS this XCD's L2, 32 CUs stream 8 MB per tile time through 4 MB: FETCH_SIZE x2 of the qkv launch
    // 1 046 MB against 355 MB in this order and 168 MB algorithmic; same time, the misses land in the Infinity Cache.)
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

    // ---- LDS-DMA staging.  A stage = 16 pieces of 8 LDS rows x 128 B; this wave issues two of them.  Lane l of a piece
    // writes LDS row (l>>3), 16-byte chunk (l&7) and fetches chunk (l&7) ^ (l>>3) of the matrix row that belongs there (the
    // read side applies the same XOR; pieces start at multiples of 8 rows).
    //   activations, half h: LDS rows = tile rows wm'*64 + h*32 + q*8 .. +7 for every row quarter wm' (what the mt in
    //     {2h, 2h+1} fragments of all waves read): this wave's pieces are wm' = wave>>1, q = (wave&1)*2 + j;
    //   weights, half h: the LDS rows of column tiles t in {4h .. 4h+3} of both column halves: this wave's pieces are
    //     half g' = wave>>2, tile t = 4h + (wave&3), LDS rows g'*128 + t*16 + j*8 .. +7 = weight rows g'*128 + (j*8 + l>>3)*8 + t.
    const int pr = lane >> 3;
    const int voff_x = pr * K2 + (((lane & 7) ^ pr) << 4);
    const int voff_w = pr * 8 * K2 + (((lane & 7) ^ pr) << 4);
    auto stage_x = [&](int buf, int h, __amdgpu_buffer_rsrc_t rs, int kb) {
        if constexpr (DBG >= 3) return;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int r = (wave >> 1) * 64 + h * 32 + ((wave & 1) * 2 + j) * 8;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(smb + buf * BUFB + r * ROWB), 16, voff_x, r * K2 + kb, 0, 0);
        }
    };
    auto stage_w = [&](int buf, int h, __amdgpu_buffer_rsrc_t rs, int kb) {
        if constexpr (DBG >= 3) return;
        const int tt = 4 * h + (wave & 3);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int rho = (wave >> 2) * 128 + tt * 16 + j * 8;             // LDS row
            const int n = (wave >> 2) * 128 + j * 64 + tt;                   // weight row of the piece's first lane
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(smb + buf * BUFB + OPB + rho * ROWB), 16, voff_w, n * K2 + kb, 0, 0);
        }
    };
    auto rsrc_x = [&](int64_t m0) {
        return __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + m0 * p.K), 0, 0x7fffffff, 0x00020000);
    };
    auto rsrc_w = [&](int n0) {
        return __builtin_amdgcn_make_buffer_rsrc((void*)(p.W + (int64_t)n0 * p.K), 0, 0x7fffffff, 0x00020000);
    };

    // ---- fragment reads.  MFMA A operand (rows i): activation row wm*64 + mt*16 + fr; B operand (columns j): LDS weight row
    // gm*128 + t*16 + fr = weight row gm*128 + fr*8 + t.  Chunk kc*4 + fq of either, XOR (fr & 7).  Accumulator register r of
    // lane (fr, fq), tile (mt, t): row wm*64 + mt*16 + fq*4 + r, column gm*128 + fr*8 + t.
    const int lx[2] = {(wm * 64 + fr) * ROWB + (((0 + fq) ^ (fr & 7)) << 4), (wm * 64 + fr) * ROWB + (((4 + fq) ^ (fr & 7)) << 4)};
    const int lw[2] = {OPB + (gm * 128 + fr) * ROWB + (((0 + fq) ^ (fr & 7)) << 4), OPB + (gm * 128 + fr) * ROWB + (((4 + fq) ^ (fr & 7)) << 4)};

    f32x4 acc[4][8];
    bf16x8 xf[4][2], wf[4][2];
    if constexpr (DBG >= 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int kc = 0; kc < 2; ++kc) { xf[i][kc] = bf16x8{}; wf[i][kc] = bf16x8{}; }
    }
    auto read_x = [&](int buf, int half) {                  // row tiles mt = 2 half, 2 half + 1
        if constexpr (DBG >= 4) return;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int kc = 0; kc < 2; ++kc)
                xf[half * 2 + i][kc] = *reinterpret_cast<const bf16x8*>(smb + buf * BUFB + lx[kc] + (half * 2 + i) * (16 * ROWB));
    };
    auto read_w = [&](int buf, int half) {                  // column tiles t = 4 half .. 4 half + 3, always into wf[0..3]
        if constexpr (DBG >= 4) return;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int kc = 0; kc < 2; ++kc)
                wf[i][kc] = *reinterpret_cast<const bf16x8*>(smb + buf * BUFB + lw[kc] + (half * 4 + i) * (16 * ROWB));
    };
    auto mfma16 = [&](int mh, int th) {                     // quadrant (row half mh of the wave tile, column half th) x K = 64
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kc = 0; kc < 2; ++kc)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    acc[mh * 2 + i][th * 4 + t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[mh * 2 + i][kc], wf[t][kc], acc[mh * 2 + i][th * 4 + t], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };

    // DBG 5 (lab): s_memtime stamps of waves 0 and 4 into C2 (uint64 [block][tile slot < 8][wave half][6])
    unsigned long long* stamps = reinterpret_cast<unsigned long long*>(p.C2);
    int stamp_slot = 0;
    auto stamp = [&](int which) {
        if constexpr (DBG == 5) {
            if ((wave & 3) == 0 && lane == 0 && stamp_slot < 8)
                stamps[((blockIdx.x * 8 + stamp_slot) * 2 + gm) * 6 + which] = __builtin_amdgcn_s_memtime();
        }
    };
    // Epilogue stores and the counted DMA waits (round 4).  vmcnt retires IN ORDER and counts stores: a wait that retires a
    // DMA stage issued after an epilogue store also waits for that store's acknowledgement, i.e. for the tile's output to
    // drain to L2 / HBM.  So (a) the stages of the next tile that can be issued BEFORE the stores are (the second halves
    // of its K-tile 1: their LDS regions are free once phase 8 is through), and (b) the waits of the next tile's phases
    // 1-5 allow for the NST store instructions that sit between the stages they retire and the ones they may leave in
    // flight: the first wait that has to see the stores acknowledged is phase 6's -- five phases of main loop later.
    constexpr int NST = (EPI == EPI_BIAS_GELU_DROP || EPI == EPI_BIAS_GELU_DROP_G) ? 32 : 16;   // store instructions per wave and tile (lower bound)
    int t = blockIdx.x;
    if (t >= n_tiles) return;
    int64_t m0; int n0;
    tile_origin(t, m0, n0);
    __amdgpu_buffer_rsrc_t xc = rsrc_x(m0), wc = rsrc_w(n0), xn = xc, wnx = wc;

    // ---- prologue: what phases 3..8 of a previous iteration would have issued for this tile's first two K-tiles
    stage_x(0, 0, xc, 0); stage_w(0, 0, wc, 0); stage_x(0, 1, xc, 0); stage_w(0, 1, wc, 0);
    stage_x(1, 0, xc, 2 * BK); stage_w(1, 0, wc, 2 * BK); stage_x(1, 1, xc, 2 * BK); stage_w(1, 1, wc, 2 * BK);
    wait_vm<0>();                                           // the first tile starts with both K-tiles landed: its first-iteration
    wg_barrier();                                           // waits (which allow for NST stores that are not there) have nothing to retire

    while (true) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        stamp(0);
        if (gm == 1) wg_barrier();                          // waves 4-7 run one barrier behind waves 0-3
        stamp(1);

        const int tn = t + gridDim.x;
        int64_t mn = m0; int nn = n0;
        for (int it = 0; it < n_it; ++it) {
            // the K-tile pair staged from phase 3 on: the next pair of this tile, or the first pair of the next tile
            const bool last_it = it + 1 == n_it;
            bool more = true;
            int kb_n = (2 * it + 2) * (2 * BK);             // byte offset of that pair's first K-tile within a row
            if (last_it) {
                kb_n = 0;
                more = tn < n_tiles;
                if (more) { tile_origin(tn, mn, nn); xn = rsrc_x(mn); wnx = rsrc_w(nn); }
            }
            const int kb_c = (2 * it + 1) * (2 * BK);       // the current pair's second K-tile

            // one phase = { fragment reads + one DMA stage + counted wait | barrier | 16 MFMAs | barrier }.  Quadrant order
            // (X0,W0) (X1,W0) (X1,W1) (X0,W1): both activation halves stay in registers, the weight halves share theirs.
            // A stage issued in phase q is first read in phase q + 5 or later; the wait of phase q + 3 (after that phase's
            // own issue: three younger stages = 6 DMA instructions may stay in flight) retires it, the barrier behind it
            // makes every wave's pieces visible, one whole phase before the first read.  A region is restaged two or
            // more phases after its last read.
#define HWGAT_WAIT(NLAST) do { if constexpr (DBG < 2) { if (more) wait_vm<6>(); else wait_vm<NLAST>(); } } while (0)
            // first iteration after an epilogue: K-tile 1's second halves were staged ahead of the stores (see NST above), phases
            // 1-2 issue nothing, and the waits count the stores in: outstanding and younger than the stage a wait retires are
            //   phase 1: sx(1,0) sw(1,0) sx(1,1) sw(1,1) + stores      phase 2: one stage less      phase 3: + this phase's stage ...
#define HWGAT_WAIT_R(NMORE, NLAST) do { if constexpr (DBG < 2) { if (more) wait_vm<NMORE + NST>(); else wait_vm<NLAST + NST>(); } } while (0)
            const bool res_it = it == 0;                    // K-tile B's second halves are already staged (prologue / before the epilogue)
            // phase 1: K-tile A (buffer 0), quadrant (X0, W0)
            read_x(0, 0); read_w(0, 0);
            if (res_it) { HWGAT_WAIT_R(8, 8); } else { stage_x(1, 1, xc, kb_c); HWGAT_WAIT(6); }
            wg_barrier(); wait_lds(); __builtin_amdgcn_sched_barrier(0); mfma16(0, 0); wg_barrier();
            // phase 2: (X1, W0)
            read_x(0, 1);
            if (res_it) { HWGAT_WAIT_R(6, 6); } else { stage_w(1, 1, wc, kb_c); HWGAT_WAIT(6); }
            wg_barrier(); wait_lds(); __builtin_amdgcn_sched_barrier(0); mfma16(1, 0); wg_barrier();
            // phase 3: (X1, W1)
            read_w(0, 1);
            if (more) stage_x(0, 0, xn, kb_n);
            if (res_it) { HWGAT_WAIT_R(6, 4); } else { HWGAT_WAIT(4); }
            wg_barrier(); wait_lds(); __builtin_amdgcn_sched_barrier(0); mfma16(1, 1); wg_barrier();
            // phase 4: (X0, W1) -- nothing to read
            if (more) stage_w(0, 0, wnx, kb_n);
            if (res_it) { HWGAT_WAIT_R(6, 2); } else { HWGAT_WAIT(2); }
            wg_barrier(); __builtin_amdgcn_sched_barrier(0); mfma16(0, 1); wg_barrier();
            // phases 5-8: the same on K-tile B (buffer 1)
            read_x(1, 0); read_w(1, 0);
            if (more) stage_x(0, 1, xn, kb_n);
            if (res_it) { HWGAT_WAIT_R(6, 0); } else { HWGAT_WAIT(0); }
            wg_barrier(); wait_lds(); __builtin_amdgcn_sched_barrier(0); mfma16(0, 0); wg_barrier();
            read_x(1, 1);
            if (more) stage_w(0, 1, wnx, kb_n);
            HWGAT_WAIT(0);
            wg_barrier(); wait_lds(); __builtin_amdgcn_sched_barrier(0); mfma16(1, 0); wg_barrier();
            read_w(1, 1);
            if (more) stage_x(1, 0, xn, kb_n + 2 * BK);
            HWGAT_WAIT(0);
            wg_barrier(); wait_lds(); __builtin_amdgcn_sched_barrier(0); mfma16(1, 1); wg_barrier();
            if (more) stage_w(1, 0, wnx, kb_n + 2 * BK);
            HWGAT_WAIT(0);
            wg_barrier(); __builtin_amdgcn_sched_barrier(0); mfma16(0, 1); wg_barrier();
#undef HWGAT_WAIT
#undef HWGAT_WAIT_R
        }
        stamp(2);
        if (gm == 0) wg_barrier();                          // waves 0-3 wait for the partner's last cluster: both groups
                                                            // run the epilogue together (two waves per SIMD share the vector pipe)
        // the next tile's K-tile 1, second halves (buffer 1, last read in phases 6 / 7): issued here, AHEAD of the epilogue's
        // stores, so that the waits which retire them need not see the stores acknowledged
        if (tn < n_tiles) { stage_x(1, 1, xn, 2 * BK); stage_w(1, 1, wnx, 2 * BK); }

        // ---- epilogue, straight from the accumulators: acc[mt][t][r] is row m0 + wm*64 + mt*16 + fq*4 + r, column
        // n0 + gm*128 + fr*8 + t -- a lane's eight tiles t of one (mt, r) are 8 consecutive columns (16 bytes), the 16 lanes
        // fr of a row 256 contiguous bytes, an instruction = 4 whole-line row segments.
        {
            const uint32_t epi_th = drop_thresh(p.epi_p);
            const float epi_sc = 1.0f / (1.0f - p.epi_p);
            const int col = n0 + gm * 128 + fr * 8;
            float cb[8], cs[8];                             // bias (or c_n of the folded LayerNorm); s_n of the fold
#pragma unroll
            for (int e = 0; e < 8; ++e) { cb[e] = 0.f; cs[e] = 0.f; }
            if constexpr (STAT == X_LNFOLD) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const f32x4 s4 = *reinterpret_cast<const f32x4*>(p.gamma + col + 4 * q);
                    const f32x4 c4 = *reinterpret_cast<const f32x4*>(p.beta + col + 4 * q);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { cs[4 * q + e] = s4[e]; cb[4 * q + e] = c4[e]; }
                }
            } else if constexpr (epi_has_bias(EPI)) {
                if (p.bias) {
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const f32x4 b4 = *reinterpret_cast<const f32x4*>(p.bias + col + 4 * q);
#pragma unroll
                        for (int e = 0; e < 4; ++e) cb[4 * q + e] = b4[e];
                    }
                }
            }
            stamp(3);
            float* rowstat = reinterpret_cast<float*>(smb + STAT_OFF);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int64_t row0 = m0 + wm * 64 + mt * 16 + fq * 4;
                // the residual / auxiliary operand of the piece's four rows is requested before any of it is used
                u32x4 ex[4];
                if constexpr (epi_reads_extra(EPI)) {
                    const bf16_t* src = (EPI == EPI_BIAS_DROP_RES ? p.res : p.aux) + row0 * p.N + col;
#pragma unroll
                    for (int r = 0; r < 4; ++r) ex[r] = *reinterpret_cast<const u32x4*>(src + r * (int64_t)p.N);
                }
                MergeWalk mw;
                if constexpr (STAT == X_STAT_MERGE) mw.start(row0, p.mg_F, p.mg_K, 1);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t row = row0 + r;
                    const int64_t off = row * p.N + col;
                    float o[8];
#pragma unroll
                    for (int tt = 0; tt < 8; ++tt) o[tt] = acc[mt][tt][r];
                    if constexpr (STAT == X_LNFOLD) {
                        const float rr = p.rstd[row], tm = p.mean[row] * rr;
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = o[e] * rr + (cb[e] - cs[e] * tm);
                    } else if constexpr (epi_has_bias(EPI)) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] += cb[e];
                    }
                    float dk[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
                    if constexpr (epi_drops(EPI)) {
                        if (epi_th) {
                            const f32x4 k0 = drop_keep4(p.epi_seed, (uint64_t)off, epi_th, epi_sc), k1 = drop_keep4(p.epi_seed, (uint64_t)off + 4, epi_th, epi_sc);
#pragma unroll
                            for (int e = 0; e < 4; ++e) { dk[e] = k0[e]; dk[4 + e] = k1[e]; }
                        }
                    }
                    if constexpr (EPI == EPI_BIAS_DROP_RES) {
                        float rs[8];
                        unpack8(ex[r], rs);
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = rs[e] + o[e] * dk[e];
                    } else if constexpr (EPI == EPI_BIAS_GELU_DROP) {
                        const u32x4 pre = pack8(o);
                        *reinterpret_cast<u32x4*>(p.C2 + off) = pre;
                        float h[8];
                        unpack8(pre, h);                    // gelu on the bf16-rounded pre-activation that backward will see
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = gelu_f(h[e]) * dk[e];
                    } else if constexpr (EPI == EPI_BIAS_GELU_DROP_G) {
                        float g8[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) gelu_fwd_grad(o[e], dk[e], o[e], g8[e]);
                        *reinterpret_cast<u32x4*>(p.C2 + off) = pack8(g8);
                    } else if constexpr (EPI == EPI_MUL_AUX) {
                        float h[8];
                        unpack8(ex[r], h);
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] *= h[e];
                    } else if constexpr (EPI == EPI_GELU_BWD) {
                        float h[8];
                        unpack8(ex[r], h);
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = o[e] * dk[e] * gelu_grad(h[e]);
                    }
                    const u32x4 outv = pack8(o);
                    if constexpr (STAT == X_STAT || STAT == X_STAT_MERGE) {   // statistics of what the next LayerNorm reads: the bf16 values
                        float q[8];
                        unpack8(outv, q);
                        float s1 = 0.f, s2 = 0.f;
#pragma unroll
                        for (int e = 0; e < 8; ++e) { s1 += q[e]; s2 += q[e] * q[e]; }
                        s1 = group_sum<16>(s1); s2 = group_sum<16>(s2);
                        if (fr == 0) {
                            f32x2 st = {s1, s2};
                            *reinterpret_cast<f32x2*>(rowstat + (gm * BT + wm * 64 + mt * 16 + fq * 4 + r) * 2) = st;
                        }
                    }
                    bf16_t* dst = p.C + off;
                    if constexpr (STAT == X_STAT_MERGE) { dst = p.C + mw.off(p.N) + col; mw.next(); }
                    if constexpr (DBG == 1) asm volatile("" ::"v"(outv));
                    else *reinterpret_cast<u32x4*>(dst) = outv;
                }
            }
            if constexpr (STAT == X_STAT || STAT == X_STAT_MERGE) {
                wait_lds();
                wg_barrier();
                if (tid < BT) {                             // the two column halves in a fixed order, one atomic per row and tile
                    int64_t mr = m0 + tid;
                    if constexpr (STAT == X_STAT_MERGE) { MergeWalk w; w.start(m0 + tid, p.mg_F, p.mg_K, 1); mr = w.mrow(); }
                    atomicAdd(p.stat_sum + mr, rowstat[2 * tid] + rowstat[2 * (BT + tid)]);
                    atomicAdd(p.stat_sq + mr, rowstat[2 * tid + 1] + rowstat[2 * (BT + tid) + 1]);
                }
                wait_lds();
                wg_barrier();                               // the scratch is rewritten by the next tile's epilogue
            }
        }
        stamp(4);
        ++stamp_slot;
        t = tn;
        if (t >= n_tiles) break;
        m0 = mn; n0 = nn; xc = xn; wc = wnx;
    }
}

template <int EPI, int STAT>
int go(const NtArgsB& a, int grid, hipStream_t st) {
    gemm_nt8w_bf16_k<EPI, STAT><<<grid, 512, 0, st>>>(a);
    HWGAT_LAUNCH_CHECK();
}

}  // namespace

// true iff the shape / prologue / epilogue combination is one this kernel takes
bool hwgat_nt8w_bf16_takes(const NtArgsB& a, int pro, int epi) {
    if (a.M % BT || a.N % BT || a.K % (2 * BK) || a.M < BT) return false;
    if ((int64_t)BT * a.K * 2 > 0x3fffffff) return false;   // 32-bit DMA offsets within a tile's rows
    if (pro != PRO_NONE && pro != PRO_LN_FOLD) return false;
    if (a.stat_sum != nullptr) return pro == PRO_NONE && epi == EPI_BIAS_DROP_RES;
    if (pro == PRO_LN_FOLD) return epi == EPI_BIAS || epi == EPI_BIAS_GELU_DROP || epi == EPI_BIAS_GELU_DROP_G;
    return epi == EPI_NONE || epi == EPI_BIAS || epi == EPI_BIAS_DROP_RES || epi == EPI_BIAS_GELU_DROP_G || epi == EPI_MUL_AUX;
}

int hwgat_launch_nt8w_bf16(const NtArgsB& a, int pro, int epi, hipStream_t st) {
    if (!hwgat_nt8w_bf16_takes(a, pro, epi)) return HWGAT_ESHAPE;
    const int64_t tiles = (a.M / BT) * (a.N / BT);
    if (tiles > 0x7fffffff) return HWGAT_ESHAPE;
    const int grid = (int)(tiles < 256 ? tiles : 256);          // persistent: one 8-wave block per CU
    if (a.stat_sum != nullptr) return a.mg_K > 0 ? go<EPI_BIAS_DROP_RES, X_STAT_MERGE>(a, grid, st) : go<EPI_BIAS_DROP_RES, X_STAT>(a, grid, st);
    if (pro == PRO_LN_FOLD) {
        if (epi == EPI_BIAS) return go<EPI_BIAS, X_LNFOLD>(a, grid, st);
        if (epi == EPI_BIAS_GELU_DROP) return go<EPI_BIAS_GELU_DROP, X_LNFOLD>(a, grid, st);
        return go<EPI_BIAS_GELU_DROP_G, X_LNFOLD>(a, grid, st);
    }
#ifdef HWGAT_LAB
    if (epi == EPI_NONE) {
        const char* e = lab_env("HWGAT_NT8W_DBG");
        const int dbg = e ? atoi(e) : 0;
        auto run = [&](auto tag) { gemm_nt8w_bf16_k<EPI_NONE, X_NONE, decltype(tag)::value><<<grid, 512, 0, st>>>(a); };
        if (dbg == 1) { run(std::integral_constant<int, 1>{}); HWGAT_LAUNCH_CHECK(); }
        if (dbg == 2) { run(std::integral_constant<int, 2>{}); HWGAT_LAUNCH_CHECK(); }
        if (dbg == 3) { run(std::integral_constant<int, 3>{}); HWGAT_LAUNCH_CHECK(); }
        if (dbg == 4) { run(std::integral_constant<int, 4>{}); HWGAT_LAUNCH_CHECK(); }
        if (dbg == 5 && a.C2) { run(std::integral_constant<int, 5>{}); HWGAT_LAUNCH_CHECK(); }
    }
#endif
    switch (epi) {
        case EPI_NONE: return go<EPI_NONE, X_NONE>(a, grid, st);
        case EPI_BIAS: return go<EPI_BIAS, X_NONE>(a, grid, st);
        case EPI_BIAS_DROP_RES: return go<EPI_BIAS_DROP_RES, X_NONE>(a, grid, st);
        case EPI_BIAS_GELU_DROP_G: return go<EPI_BIAS_GELU_DROP_G, X_NONE>(a, grid, st);
        case EPI_MUL_AUX: return go<EPI_MUL_AUX, X_NONE>(a, grid, st);
        default: return HWGAT_EINVAL;
    }
}
