// fp32 NT linear GEMM for HWGAT on gfx950 (BASELINE config 2, the headline): the fp32 twin of gemm_bf16_nt8w.hip -- read
// that file's header first.
//   C[M,N] = A[M,K] . W[N,K]^T (+ fused epilogue), everything fp32, v_mfma_f32_16x16x4_f32 (exact fp32 products and
// sums: `dtype: f32` means fp32 MFMA arithmetic, no split-operand emulation) -- the nn.Linear of
// hwgat/models/HWGATE.py:86,115-116,131-135 with LayerNorm :203,:219 folded in.
//
// Same structure: 256x256 tile, EIGHT waves (two per SIMD) in a ping-pong offset by one barrier, operands HBM/L2 -> LDS by
// LDS-DMA behind counted vmcnt waits, transposed product (a lane ends with 16 consecutive output columns of a row),
// epilogue through wave-private LDS strips so that every global access is a whole line.  The byte geometry of the LDS
// image is identical (a K-tile is 32 floats = 128 bytes per row).  What differs from the bf16 kernel:
//   * a 16-byte fragment read is 4 consecutive k of one row; v_mfma_f32_16x16x4_f32 wants ONE k per lane group, so the
//     four floats feed four successive MFMAs: MFMA t of a chunk sums k = {t, 4+t, 8+t, 12+t} of its 16 -- both operands
//     use the same assignment, the order of a sum's terms is free;
//   * a phase is 64 MFMAs of 32 cycles (2 048 matrix-pipe cycles) against the same few hundred cycles of fragment reads
//     and DMA issue: the partner wave's memory slot hides completely, the pipe stays busy through the main loop, and
//     what the one-wave-per-SIMD kernels (gemm_f32_nt256.hip) lose in their epilogues -- the other 10-20 % of a fused
//     launch -- runs here with two waves per SIMD sharing the vector pipe.
// Needs M % 256 == N % 256 == K % 64 == 0, no A-side prologue (PRO_NONE or the folded LayerNorm); everything else stays
// on gemm_nt256_k / gemm_nt_k (hwgat_linear_nt_f32 decides).
#include <type_traits>
#include "common.h"
#include "fused_ops.h"
#include "gemm_f32.h"

namespace {

constexpr int BT = 256, BK = 32;
constexpr int ROWB = 4 * BK;                  // bytes of one LDS row (one K-tile of one matrix row)
constexpr int OPB = BT * ROWB;                // one operand tile: 32 KiB
constexpr int BUFB = 2 * OPB;                 // activations | weights of one K-tile: 64 KiB
constexpr int STRIP_OFF = 2 * BUFB;           // epilogue: one 16-row x 64-fp32 strip per wave (the row statistics alias them)
constexpr int SMEM = STRIP_OFF + 8 * 16 * 256;   // 160 KiB, all of the CU's LDS

typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wait_lds() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void wg_barrier() {          // raw: no implicit vmcnt(0), the DMA queue survives it
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

// STAT: 0 = plain epilogue, 1 = + row statistics, 2 = + row statistics and the merged store, 3 = folded LayerNorm
template <int EPI, int STAT>
__global__ __launch_bounds__(512, 2) void gemm_nt8w_f32_k(NtArgs p) {
    __shared__ __attribute__((aligned(16))) unsigned char smb[SMEM];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int gm = wave >> 2, wn = wave & 3;               // ping-pong group = row half of the tile; column quarter
    const int fr = lane & 15, fq = lane >> 4;
    const int tiles_n = p.N / BT;
    const int row_blocks = (int)(p.M / BT);
    const int n_tiles = row_blocks * tiles_n;
    const int n_it = p.K / (2 * BK);                        // main-loop iterations per tile: two K-tiles each
    const int K2 = p.K * 4;                                 // bytes per matrix row

    // ---- tile order: XCD-aware (the n-tiles of a 256-row block run on one XCD at the same time), see gemm_nt_k
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

    // ---- LDS-DMA staging.  A stage = 16 pieces of 8 rows x 128 B; this wave issues two of them.
    // activations, half h (rows h*64 .. h*64+63 of BOTH row halves of the tile): this wave's pieces are rows
    //   gm*128 + h*64 + ((wave&3)*2 + j)*8 .. +7;   weights, half h (the rows the t in {2h, 2h+1} MFMA tiles of every
    //   wave read): rows (wave>>1)*64 + ((wave&1)*2 + j)*16 + h*8 .. +7.
    // per-lane source offset inside a piece: row (lane>>3), 16-byte chunk (lane&7) ^ key(row) -- the read side applies
    // the same XOR.  key = row & 7 for activations; for weights key = ((row>>4)&3)*2 + ((row>>1)&1), which is what makes
    // the permuted row set of a weight fragment (below) conflict free.
    const int pr = lane >> 3;
    const int voff_x = pr * K2 + (((lane & 7) ^ pr) << 4);
    int voff_w[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int fqp = (wave & 1) * 2 + j;
        voff_w[j] = pr * K2 + (((lane & 7) ^ (fqp * 2 + ((lane >> 4) & 1))) << 4);
    }
    auto x_row = [&](int h, int j) { return gm * 128 + h * 64 + ((wave & 3) * 2 + j) * 8; };
    auto w_row = [&](int h, int j) { return (wave >> 1) * 64 + ((wave & 1) * 2 + j) * 16 + h * 8; };
    auto stage_x = [&](int buf, int h, __amdgpu_buffer_rsrc_t rs, int kb) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int r = x_row(h, j);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(smb + buf * BUFB + r * ROWB), 16, voff_x, r * K2 + kb, 0, 0);
        }
    };
    auto stage_w = [&](int buf, int h, __amdgpu_buffer_rsrc_t rs, int kb) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int r = w_row(h, j);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(smb + buf * BUFB + OPB + r * ROWB), 16, voff_w[j], r * K2 + kb, 0, 0);
        }
    };
    auto rsrc_x = [&](int64_t m0) {
        return __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + m0 * p.K), 0, 0x7fffffff, 0x00020000);
    };
    auto rsrc_w = [&](int n0) {
        return __builtin_amdgcn_make_buffer_rsrc((void*)(p.W + (int64_t)n0 * p.K), 0, 0x7fffffff, 0x00020000);
    };

    // ---- fragment reads.  Activation fragment (mt, kc): row gm*128 + mt*16 + fr, chunk kc*4 + fq.  Weight fragment (t, kc):
    // MFMA row i = fr reads weight row wn*64 + (fr>>2)*16 + t*4 + (fr&3), so accumulator register r of lane (fr, fq) is
    // output column wn*64 + fq*16 + t*4 + r: the four tiles t interleave to 16 consecutive columns per lane.
    const int lx[2] = {gm * (128 * ROWB) + fr * ROWB + (((0 + fq) ^ (fr & 7)) << 4),
                       gm * (128 * ROWB) + fr * ROWB + (((4 + fq) ^ (fr & 7)) << 4)};
    const int wkey = (fr >> 2) * 2 + ((fr >> 1) & 1);
    const int wrow = wn * 64 + (fr >> 2) * 16 + (fr & 3);
    const int lw[2] = {OPB + wrow * ROWB + (((0 + fq) ^ wkey) << 4), OPB + wrow * ROWB + (((4 + fq) ^ wkey) << 4)};

    f32x4 acc[8][4];
    f32x4 xf[4][2], wf[4][2];
    auto read_x = [&](int buf, int half) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int kc = 0; kc < 2; ++kc)
                xf[i][kc] = *reinterpret_cast<const f32x4*>(smb + buf * BUFB + lx[kc] + (half * 4 + i) * (16 * ROWB));
    };
    auto read_w = [&](int buf, int half) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int kc = 0; kc < 2; ++kc)
                wf[half * 2 + i][kc] = *reinterpret_cast<const f32x4*>(smb + buf * BUFB + lw[kc] + (half * 2 + i) * (4 * ROWB));
    };
    auto mfma16 = [&](int mh, int th) {                     // quadrant (row half mh of the wave tile, column half th) x K = 32: 64 MFMAs
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kc = 0; kc < 2; ++kc)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int t = 0; t < 2; ++t)
                        acc[mh * 4 + i][th * 2 + t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[th * 2 + t][kc][e], xf[i][kc][e], acc[mh * 4 + i][th * 2 + t], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };

    int t = blockIdx.x;
    if (t >= n_tiles) return;
    int64_t m0; int n0;
    tile_origin(t, m0, n0);
    __amdgpu_buffer_rsrc_t xc = rsrc_x(m0), wc = rsrc_w(n0), xn = xc, wnx = wc;

    // ---- prologue: what phases 3..8 of a previous iteration would have issued for this tile's first two K-tiles
    stage_x(0, 0, xc, 0); stage_w(0, 0, wc, 0); stage_w(0, 1, wc, 0); stage_x(0, 1, xc, 0);
    stage_x(1, 0, xc, ROWB); stage_w(1, 0, wc, ROWB);
    wait_vm<4>();                                           // K-tile 0 has landed (this wave's pieces) ...
    wg_barrier();                                           // ... and everyone's

    while (true) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (gm == 1) wg_barrier();                          // waves 4-7 run one barrier behind waves 0-3

        const int tn = t + gridDim.x;
        int64_t mn = m0; int nn = n0;
        for (int it = 0; it < n_it; ++it) {
            // the K-tile pair staged from phase 3 on: the next pair of this tile, or the first pair of the next tile
            const bool last_it = it + 1 == n_it;
            bool more = true;
            int kb_n = (2 * it + 2) * ROWB;             // byte offset of that pair's first K-tile within a row
            if (last_it) {
                kb_n = 0;
                more = tn < n_tiles;
                if (more) { tile_origin(tn, mn, nn); xn = rsrc_x(mn); wnx = rsrc_w(nn); }
            }
            const int kb_c = (2 * it + 1) * ROWB;       // the current pair's second K-tile

            // one phase = { fragment reads + one DMA stage + counted wait | barrier | 16 MFMAs | barrier }.
            // A stage issued in phase q is first read in phase q + 5; the wait of phase q + 3 (after that phase's own
            // issue: three younger stages = 6 DMA instructions may stay in flight) retires it, the barrier behind it makes
            // every wave's pieces visible, one whole phase before the first read.
#define HWGAT_WAIT(NLAST) do { if (more) wait_vm<6>(); else wait_vm<NLAST>(); } while (0)
            // phase 1: quadrant (0,0) of K-tile A (buffer 0)
            read_x(0, 0); read_w(0, 0);
            stage_w(1, 1, wc, kb_c);
            HWGAT_WAIT(6);
            wg_barrier(); wait_lds(); __builtin_amdgcn_sched_barrier(0); mfma16(0, 0); wg_barrier();
            // phase 2: quadrant (0,1)
            read_w(0, 1);
            stage_x(1, 1, xc, kb_c);
            HWGAT_WAIT(6);
            wg_barrier(); wait_lds(); __builtin_amdgcn_sched_barrier(0); mfma16(0, 1); wg_barrier();
            // phase 3: quadrant (1,1)
            read_x(0, 1);
            if (more) stage_x(0, 0, xn, kb_n);
            HWGAT_WAIT(4);
            wg_barrier(); wait_lds(); __builtin_amdgcn_sched_barrier(0); mfma16(1, 1); wg_barrier();
            // phase 4: quadrant (1,0) -- the column-half-0 weight fragments are still in registers
            if (more) stage_w(0, 0, wnx, kb_n);
            HWGAT_WAIT(2);
            wg_barrier(); __builtin_amdgcn_sched_barrier(0); mfma16(1, 0); wg_barrier();
            // phases 5-8: the same on K-tile B (buffer 1)
            read_x(1, 0); read_w(1, 0);
            if (more) stage_w(0, 1, wnx, kb_n);
            HWGAT_WAIT(0);
            wg_barrier(); wait_lds(); __builtin_amdgcn_sched_barrier(0); mfma16(0, 0); wg_barrier();
            read_w(1, 1);
            if (more) stage_x(0, 1, xn, kb_n);
            HWGAT_WAIT(0);
            wg_barrier(); wait_lds(); __builtin_amdgcn_sched_barrier(0); mfma16(0, 1); wg_barrier();
            read_x(1, 1);
            if (more) stage_x(1, 0, xn, kb_n + ROWB);
            HWGAT_WAIT(0);
            wg_barrier(); wait_lds(); __builtin_amdgcn_sched_barrier(0); mfma16(1, 1); wg_barrier();
            if (more) stage_w(1, 0, wnx, kb_n + ROWB);
            HWGAT_WAIT(0);
            wg_barrier(); __builtin_amdgcn_sched_barrier(0); mfma16(1, 0); wg_barrier();
#undef HWGAT_WAIT
        }
        if (gm == 0) wg_barrier();                          // waves 0-3 wait for the partner's last cluster: both groups
                                                            // run the epilogue together (two waves per SIMD share the vector pipe)

        // ---- epilogue.  acc[mt][t][r] is row m0 + gm*128 + mt*16 + fr, column n0 + wn*64 + fq*16 + t*4 + r: the four lanes
        // that hold one output row are 16 lanes apart, and a store / load instruction whose neighbouring lanes touch
        // different rows is issued as 64 separate 16-byte requests (measured: 6.7 us of store tail per tile, a third of a
        // K = 512 tile).  So each 16-row piece first turns through a WAVE-PRIVATE 4 KiB LDS strip (16 rows x 64 fp32,
        // 16-byte chunks XOR-swizzled by the row: conflict-free both ways, no barrier -- a wave's DS operations execute in
        // order), after which lane l owns row (l>>3) of 8-row pass ps, columns (l&7)*8 .. +7: eight neighbouring lanes
        // = one whole 128-byte line of C (and of the residual / auxiliary operand), 8 lines per instruction.
        {
            const uint32_t epi_th = drop_thresh(p.epi_p);
            const float epi_sc = 1.0f / (1.0f - p.epi_p);
            const int er = lane >> 3, ec = (lane & 7) * 8;
            const int col = n0 + wn * 64 + ec;
            unsigned char* strip = smb + STRIP_OFF + wave * (16 * 256);
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
            float st1[16], st2[16];                         // row statistics of this lane's 16 (piece, pass) rows
            MergeWalk mw;
            if constexpr (STAT == X_STAT_MERGE) mw.start(m0 + gm * 128 + er, p.mg_F, p.mg_K, 8);
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) {
                // the residual / auxiliary operand of both passes is requested before the piece is parked
                f32x4 ex[2][2];
                if constexpr (epi_reads_extra(EPI)) {
                    const float* src = (EPI == EPI_BIAS_DROP_RES ? p.res : p.aux) + (m0 + gm * 128 + mt * 16 + er) * p.N + col;
                    ex[0][0] = *reinterpret_cast<const f32x4*>(src);
                    ex[0][1] = *reinterpret_cast<const f32x4*>(src + 4);
                    ex[1][0] = *reinterpret_cast<const f32x4*>(src + 8 * (int64_t)p.N);
                    ex[1][1] = *reinterpret_cast<const f32x4*>(src + 8 * (int64_t)p.N + 4);
                }
#pragma unroll
                for (int tt = 0; tt < 4; ++tt)
                    *reinterpret_cast<f32x4*>(strip + fr * 256 + (((fq * 4 + tt) ^ fr) << 4)) = acc[mt][tt];
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // other lanes' writes, this lane's reads: keep the order
                __builtin_amdgcn_wave_barrier();
                f32x4 pv[2][2];
#pragma unroll
                for (int ps = 0; ps < 2; ++ps) {
                    const int R = ps * 8 + er;
                    pv[ps][0] = *reinterpret_cast<const f32x4*>(strip + R * 256 + (((ec >> 2) ^ R) << 4));
                    pv[ps][1] = *reinterpret_cast<const f32x4*>(strip + R * 256 + ((((ec >> 2) + 1) ^ R) << 4));
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // ... and the next piece's writes behind these reads
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int ps = 0; ps < 2; ++ps) {
                    const int R = ps * 8 + er;
                    const int64_t row = m0 + gm * 128 + mt * 16 + R;
                    const int64_t off = row * p.N + col;
                    const f32x4 v0 = pv[ps][0], v1 = pv[ps][1];
                    float o[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
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
                    const float h[8] = {ex[ps][0][0], ex[ps][0][1], ex[ps][0][2], ex[ps][0][3], ex[ps][1][0], ex[ps][1][1], ex[ps][1][2], ex[ps][1][3]};
                    if constexpr (EPI == EPI_BIAS_DROP_RES) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = h[e] + o[e] * dk[e];
                    } else if constexpr (EPI == EPI_BIAS_GELU_DROP) {
                        *reinterpret_cast<f32x4*>(p.C2 + off) = f32x4{o[0], o[1], o[2], o[3]};
                        *reinterpret_cast<f32x4*>(p.C2 + off + 4) = f32x4{o[4], o[5], o[6], o[7]};
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = gelu_f(o[e]) * dk[e];
                    } else if constexpr (EPI == EPI_BIAS_GELU_DROP_G) {
                        float g8[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) gelu_fwd_grad(o[e], dk[e], o[e], g8[e]);
                        *reinterpret_cast<f32x4*>(p.C2 + off) = f32x4{g8[0], g8[1], g8[2], g8[3]};
                        *reinterpret_cast<f32x4*>(p.C2 + off + 4) = f32x4{g8[4], g8[5], g8[6], g8[7]};
                    } else if constexpr (EPI == EPI_MUL_AUX) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] *= h[e];
                    } else if constexpr (EPI == EPI_GELU_BWD) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = o[e] * dk[e] * gelu_grad(h[e]);
                    }
                    if constexpr (STAT == X_STAT || STAT == X_STAT_MERGE) {   // statistics of what the next LayerNorm reads
                        float s1 = 0.f, s2 = 0.f;
#pragma unroll
                        for (int e = 0; e < 8; ++e) { s1 += o[e]; s2 += o[e] * o[e]; }
                        st1[mt * 2 + ps] = group_sum<8>(s1);
                        st2[mt * 2 + ps] = group_sum<8>(s2);
                    }
                    float* dst = p.C + off;
                    if constexpr (STAT == X_STAT_MERGE) { dst = p.C + mw.off(p.N) + col; mw.next(); }
                    *reinterpret_cast<f32x4*>(dst) = f32x4{o[0], o[1], o[2], o[3]};
                    *reinterpret_cast<f32x4*>(dst + 4) = f32x4{o[4], o[5], o[6], o[7]};
                }
            }
            if constexpr (STAT == X_STAT || STAT == X_STAT_MERGE) {
                // [4 column quarters][256 rows] x (sum, sum of squares) in the strip area, once every wave is done with its strip
                float* rowstat = reinterpret_cast<float*>(smb + STRIP_OFF);
                wait_lds();
                wg_barrier();
                if ((lane & 7) == 0) {
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        f32x2 st = {st1[q], st2[q]};
                        *reinterpret_cast<f32x2*>(rowstat + (wn * BT + gm * 128 + (q >> 1) * 16 + (q & 1) * 8 + er) * 2) = st;
                    }
                }
                wait_lds();
                wg_barrier();
                if (tid < BT) {                             // fixed order over the four column quarters, one atomic per row and tile
                    int64_t mr = m0 + tid;
                    if constexpr (STAT == X_STAT_MERGE) { MergeWalk w; w.start(m0 + tid, p.mg_F, p.mg_K, 1); mr = w.mrow(); }
                    const float a1 = (rowstat[2 * tid] + rowstat[2 * (BT + tid)]) + (rowstat[2 * (2 * BT + tid)] + rowstat[2 * (3 * BT + tid)]);
                    const float a2 = (rowstat[2 * tid + 1] + rowstat[2 * (BT + tid) + 1]) + (rowstat[2 * (2 * BT + tid) + 1] + rowstat[2 * (3 * BT + tid) + 1]);
                    atomicAdd(p.stat_sum + mr, a1);
                    atomicAdd(p.stat_sq + mr, a2);
                }
                wait_lds();
                wg_barrier();                               // the strips are rewritten by the next tile's epilogue
            }
        }
        t = tn;
        if (t >= n_tiles) break;
        m0 = mn; n0 = nn; xc = xn; wc = wnx;
    }
}

template <int EPI, int STAT>
int go(const NtArgs& a, int grid, hipStream_t st) {
    gemm_nt8w_f32_k<EPI, STAT><<<grid, 512, 0, st>>>(a);
    HWGAT_LAUNCH_CHECK();
}

}  // namespace

// true iff the shape / prologue / epilogue combination is one this kernel takes
bool hwgat_nt8w_f32_takes(const NtArgs& a, int pro, int epi) {
    if (a.M % BT || a.N % BT || a.K % (2 * BK) || a.M < BT) return false;
    if ((int64_t)BT * a.K * 4 > 0x3fffffff) return false;   // 32-bit DMA offsets within a tile's rows
    if (pro != PRO_NONE && pro != PRO_LN_FOLD) return false;
    if (a.stat_sum != nullptr) return pro == PRO_NONE && epi == EPI_BIAS_DROP_RES;
    if (pro == PRO_LN_FOLD) return epi == EPI_BIAS || epi == EPI_BIAS_GELU_DROP || epi == EPI_BIAS_GELU_DROP_G;
    return epi == EPI_NONE || epi == EPI_BIAS || epi == EPI_BIAS_DROP_RES || epi == EPI_BIAS_GELU_DROP_G || epi == EPI_MUL_AUX;
}

int hwgat_launch_nt8w_f32(const NtArgs& a, int pro, int epi, hipStream_t st) {
    if (!hwgat_nt8w_f32_takes(a, pro, epi)) return HWGAT_ESHAPE;
    const int64_t tiles = (a.M / BT) * (a.N / BT);
    if (tiles > 0x7fffffff) return HWGAT_ESHAPE;
    const int grid = (int)(tiles < 256 ? tiles : 256);          // persistent: one 8-wave block per CU
    if (a.stat_sum != nullptr) return a.mg_K > 0 ? go<EPI_BIAS_DROP_RES, X_STAT_MERGE>(a, grid, st) : go<EPI_BIAS_DROP_RES, X_STAT>(a, grid, st);
    if (pro == PRO_LN_FOLD) {
        if (epi == EPI_BIAS) return go<EPI_BIAS, X_LNFOLD>(a, grid, st);
        if (epi == EPI_BIAS_GELU_DROP) return go<EPI_BIAS_GELU_DROP, X_LNFOLD>(a, grid, st);
        return go<EPI_BIAS_GELU_DROP_G, X_LNFOLD>(a, grid, st);
    }
    switch (epi) {
        case EPI_NONE: return go<EPI_NONE, X_NONE>(a, grid, st);
        case EPI_BIAS: return go<EPI_BIAS, X_NONE>(a, grid, st);
        case EPI_BIAS_DROP_RES: return go<EPI_BIAS_DROP_RES, X_NONE>(a, grid, st);
        case EPI_BIAS_GELU_DROP_G: return go<EPI_BIAS_GELU_DROP_G, X_NONE>(a, grid, st);
        case EPI_MUL_AUX: return go<EPI_MUL_AUX, X_NONE>(a, grid, st);
        default: return HWGAT_EINVAL;
    }
}
