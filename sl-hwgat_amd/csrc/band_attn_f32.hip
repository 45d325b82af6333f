// Fused band graph-attention for the WGATE sibling model, fp32 storage and fp32 MFMA arithmetic, head_dim 16, on gfx950
// (MI355X): the workgroup-staged form.
//
// Same contract as band_attn.hip (MSA.forward's attention core of the reference's hwgat/models/WGATE.py:87-108 with
// window_partition / window_reverse, WGATE.py:32-65, folded into index arithmetic; block-tridiagonal 0 / -10000 mask of
// model_params.py:209-228 as 48-bit rows) and the same arithmetic (v_mfma_f32_16x16x4_f32, softmax in fp32 registers).  What
// changes is how the operands arrive.  There, every wave fetches its own head: 64-byte pieces of the token rows, sixteen per
// load instruction, the column forms by a second, element-wise pass over the same data, the results back in 4-byte stores:
// 10 (forward) and 28 (backward) memory instructions per frame and wave for 4 / 7 KB, and HBM held at 0.53 / 0.49 of its
// rate.  Here, as in band_attn_bf16.hip:
//
//   workgroup = (clip, part window, frame segment, NHW = 8 (forward) or 4 neighbouring heads); wave w owns head NHW hg + w
//          and walks the frames of the segment in order.  The workgroup fetches the q / k / v (/ dO) tiles of a frame
//          TOGETHER: 16 token rows x (NHW heads x 16 channels x 4 bytes) = whole 512- / 256-byte row pieces, by LDS-DMA
//          (16 bytes per lane, 1 KB per instruction, no registers), one frame group ahead into a double buffer.
//   row operand  X[row = l&15][4 (l>>4) + 0..3] of the wave's head: one ds_read_b128 = the four k-steps of a head-dim
//          contraction (S^T = K Q^T, dP^T = V dO^T).
//   column operand X[row = 4 (l>>4) + r][l&15]: four ds_read_b32 from the SAME image = the A operand of the contractions over
//          tokens (O^T = V^T P^T, dQ^T = K^T dS^T, dK^T = Q^T dS, dV^T = dO^T P) -- no second pass over global memory.
//   results come out as [channel 4g + r][token l&15] = four consecutive channels of one token per lane: one 16-byte store
//          per lane for o, dq, dk, dv, and the row's 1 / sum sits in the very lane that stores it.
//   backward: P and dS are transposed through wave-private 16 x 16 LDS tiles (one 16-byte write, four 4-byte reads); the
//          clip may be cut into frame segments with a one-query-frame halo on either side (recomputed, not stored).
//
// Every LDS read is conflict-free under one XOR swizzle of the sixteen 16-byte chunks of a 256-byte row, applied on the
// DMA's source side (xrf: the swizzle of blk_attn_f32.hip).  HBM traffic stays the algorithmic 4 E s / 7 E s plus the
// segment halos.  head_dim 32 keeps the kernels of band_attn.hip.
#include <stdlib.h>
#include "band_common.h"

namespace {
using namespace band;

typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) f32x4v lds_f32x4;
typedef __attribute__((address_space(3))) float lds_f32;
typedef __attribute__((address_space(3))) void* lds_void;

constexpr int HD = 16;
// a staged tile = 16 token rows x NHW heads (the waves of a workgroup) x 16 channels x 4 bytes
template <int NHW> struct Tile {
    static constexpr int RB = NHW * HD * 4;      // bytes per token row: 256 (4 heads) / 512 (8 heads: with d = 128 the whole q, k or v row)
    static constexpr int BYTES = 16 * RB;        // 4 / 8 KB
    static constexpr int NI = BYTES / 1024;      // DMA instructions per tile
    static constexpr int RPI = 1024 / RB;        // token rows per DMA instruction
};
constexpr int XLD = 20;                  // floats per row of a transposing tile (16 + 4: conflict-free 16-byte writes and 4-byte column reads)
constexpr int XTILE = 16 * XLD * 4;      // bytes
constexpr float NEG_INF = -__builtin_inff();

// D(16x16) += A(16x4) B(4x16): lane l supplies A[i = l&15][k = l>>4] and B[k = l>>4][j = l&15]; register r of lane l is
// D[i = 4 (l>>4) + r][j = l&15]
__device__ __forceinline__ f32x4v mfma16(float a, float b, f32x4v c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// reductions over the 4 lanes l, l^16, l^32, l^48 without LDS
__device__ __forceinline__ float xg_max(float v) {
    u32x2v r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __builtin_fmaxf(__uint_as_float(r.x), __uint_as_float(r.y));
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __builtin_fmaxf(__uint_as_float(r.x), __uint_as_float(r.y));
}
__device__ __forceinline__ float xg_sum(float v) {
    u32x2v r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r.x) + __uint_as_float(r.y);
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r.x) + __uint_as_float(r.y);
}
__device__ __forceinline__ void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void wg_barrier() { asm volatile("s_barrier" ::: "memory"); }
__device__ __forceinline__ void wave_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// chunk c (0..15) of tile row `row` sits in slot c ^ xrf(row): conflict-free ds_read_b128 of one chunk column of the 16 rows
// and conflict-free ds_read_b32 of 16 consecutive channels of two rows 4 apart (see blk_attn_f32.hip)
__device__ __forceinline__ int xrf(int row) { return (row & 7) | ((((row >> 2) ^ (row >> 3)) & 1) << 3); }
// source byte offset (token rows of `row_bytes`) of this lane's 16 bytes in DMA instruction `ins` (RPI rows) of a tile
// (the swizzle touches the low four bits of the chunk index: with 512-byte rows a chunk stays in its 256-byte half, and the
//  halves are a whole number of bank rows apart)
template <int NHW> __device__ __forceinline__ uint32_t dma_src(int lane, int ins, uint32_t row_bytes) {
    using TL = Tile<NHW>;
    const int row = TL::RPI * ins + (lane * 16) / TL::RB, cp = ((lane * 16) % TL::RB) >> 4;
    return row * row_bytes + ((cp ^ xrf(row)) << 4);
}

// operands of head w from a staged tile.  Row: X[row = l&15][4g .. 4g+3]; column: X[row = 4g + r][l&15], r = 0..3
template <int NHW> struct Ops {
    static constexpr int RB = Tile<NHW>::RB;
    uint32_t row, col[4];
    __device__ __forceinline__ Ops(int lane, int w) {
        const int lr = lane & 15, g = lane >> 4;
        row = lr * RB + (((4 * w + g) ^ xrf(lr)) << 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) col[r] = (4 * g + r) * RB + (((4 * w + (lr >> 2)) ^ xrf(4 * g + r)) << 4) + (lr & 3) * 4;
    }
    __device__ __forceinline__ f32x4v read_row(const char* tile) const { return *(const lds_f32x4*)(tile + row); }
    __device__ __forceinline__ f32x4v read_col(const char* tile) const {
        return f32x4v{*(const lds_f32*)(tile + col[0]), *(const lds_f32*)(tile + col[1]), *(const lds_f32*)(tile + col[2]),
                      *(const lds_f32*)(tile + col[3])};
    }
};
// the same operands of one head straight from memory (the segment's first key frames): `base` wave-uniform, per-lane offsets
__device__ __forceinline__ f32x4v load_row(const float* base, uint32_t roff) { return *reinterpret_cast<const f32x4v*>(base + roff); }
__device__ __forceinline__ f32x4v load_col(const float* base, uint32_t coff, int64_t row_stride) {
    return f32x4v{base[coff], base[coff + row_stride], base[coff + 2 * row_stride], base[coff + 3 * row_stride]};
}

// D[i][j] = sum_c X[i][c] Y[j][c] of two row operands: lane (j = l&15, g), register r -> D[4g + r][j]
__device__ __forceinline__ f32x4v dot_rows(const f32x4v& x, const f32x4v& y) {
    f32x4v acc = {0.f, 0.f, 0.f, 0.f};
    acc = mfma16(x.x, y.x, acc);
    acc = mfma16(x.y, y.y, acc);
    acc = mfma16(x.z, y.z, acc);
    return mfma16(x.w, y.w, acc);
}
// acc[r] (lane j) += sum_k Y[k][4g + r] a[j][k]: Y a column operand over the 16 tokens k (yc[e] = Y[k = 4g + e][l&15]),
// a[e] = A[j = l&15][k = 4g + e]
__device__ __forceinline__ f32x4v mul_cols(const f32x4v& yc, const f32x4v& a, f32x4v acc) {
    acc = mfma16(yc.x, a.x, acc);
    acc = mfma16(yc.y, a.y, acc);
    acc = mfma16(yc.z, a.z, acc);
    return mfma16(yc.w, a.w, acc);
}

// additive visibility of the 3 x 4 keys (tile t, joint 4g + r) this lane holds for query joint l&15: 0 or -inf
__device__ __forceinline__ void band_bias(uint64_t mrow, int g, float (&bias)[3][4]) {
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) bias[t][r] = ((mrow >> (16 * t + 4 * g + r)) & 1ull) ? 0.f : NEG_INF;
}
// e[t][r] = exp(scale s - max) over the visible keys of one query row (invisible / out-of-clip keys: exactly 0, as
// exp(s - 10000 - max) is in the reference's fp32 softmax, WGATE.py:97-103); returns 1 / sum
__device__ __forceinline__ float band_exp(const f32x4v (&s)[3], const float (&bias)[3][4], bool has_prev, bool has_next,
                                          f32x4v (&e)[3]) {
    constexpr float c1 = band_scale<HD>() * 1.4426950408889634f;        // scale * log2(e)
    const float edge[3] = {has_prev ? 0.f : NEG_INF, 0.f, has_next ? 0.f : NEG_INF};
    float m = NEG_INF;
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float x = __builtin_fmaf(s[t][r], c1, bias[t][r]);
            if (t != 1) x += edge[t];
            e[t][r] = x;
            m = __builtin_fmaxf(m, x);
        }
    m = xg_max(m);                                                       // the diagonal is always visible: m is finite
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            e[t][r] = __builtin_amdgcn_exp2f(e[t][r] - m);
            sum += e[t][r];
        }
    return 1.0f / xg_sum(sum);
}

// unit = ((clip, window), segment, group of NHW heads): one workgroup; wave w of it owns head NHW hg + w
struct GroupF {
    int64_t tok0;          // token index of (clip, frame 0, first joint of the window)
    int hg, w, f0, f1;     // owned frames [f0, f1)
    int bw;                // clip * nW + window (see BandUnit)
};
__device__ __forceinline__ GroupF decode_group(const BandGeom& g, int blk, int nhw) {
    GroupF r;
    const int n_hg = (g.nH + nhw - 1) / nhw;
    r.hg = blk % n_hg;
    int t = blk / n_hg;
    const int sgi = t % g.n_seg;
    t /= g.n_seg;
    r.bw = t;
    r.w = t % g.nW;
    const int b = t / g.nW;
    r.tok0 = (int64_t)b * g.F * g.K + r.w * 16;
    r.f0 = sgi * g.seg;
    r.f1 = min(g.F, r.f0 + g.seg);
    return r;
}

// =============================================================== forward
// PF = frames per staged group (1 or 2): group = PF x (Q, K, V) tiles, double buffered
template <int NHW, int PF, int MINW, bool ADROP>
__global__ __launch_bounds__(NHW * 64, MINW) void band_fwd_f32st_k(const float* __restrict__ qkv, float* __restrict__ o,
                                                              const uint64_t* __restrict__ maskrows, BandGeom g,
                                                              int64_t qkv_bytes, AttnDrop ad) {
    if constexpr (ADROP) ad.seed += seed_base_of(ad.base);
    using TL = Tile<NHW>;
    constexpr int TILE = TL::BYTES;
    constexpr int WPF = NHW / PF;                                // waves sharing the DMA work of one frame
    constexpr int IPW = TL::NI / WPF;                            // DMA instructions per tile and wave
    static_assert((PF == 1 || PF == 2 || PF == 4) && TL::NI % WPF == 0, "frame group = 1, 2 or 4 frames");
    constexpr int GROUP = PF * 3 * TILE;
    __shared__ __attribute__((aligned(1024))) char sm[2 * GROUP];    // [2][PF][Q K V][TILE]
    const int lane = threadIdx.x & 63, lr = lane & 15, gq = lane >> 4;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const GroupF un = decode_group(g, blockIdx.x, NHW);
    const int head = NHW * un.hg + wib;
    const bool live = head < g.nH;                               // waves past the last head stage tiles but compute nothing
    const int hd_eff = min(head, g.nH - 1);
    const int64_t rs = 3 * (int64_t)g.d;                         // qkv row stride (elements)
    const float* gb = qkv + un.tok0 * rs + NHW * un.hg * HD;     // the group's q columns of (frame 0, joint 0)
    const float* qb = qkv + un.tok0 * rs + hd_eff * HD;          // this wave's head (prologue loads)
    float* ob = o + un.tok0 * (int64_t)g.d + hd_eff * HD;
    const int64_t fs = (int64_t)g.K * rs;                        // frame stride in qkv
    const uint32_t roff = lr * (uint32_t)rs + 4 * gq;            // per-lane offsets in qkv: row operand ...
    const uint32_t coff = 4 * gq * (uint32_t)rs + lr;            // ... column operand
    const uint32_t ooff = lr * (uint32_t)g.d + 4 * gq;           // ... and in o
    float bias[3][4];
    band_bias(maskrows[un.w * 16 + lr], gq, bias);
    const Ops<NHW> ops(lane, wib);

    // DMA: resource = from the group's first byte to the end of the tensor (lanes past it read zeros: a last head group
    // of fewer than NHW heads stages columns that belong to no head)
    const int64_t left = qkv_bytes - ((const char*)gb - (const char*)qkv);
    const int span = (int)min(left, (int64_t)0x7fffffff);
    uint32_t voff[IPW];
#pragma unroll
    for (int j = 0; j < IPW; ++j) voff[j] = dma_src<NHW>(lane, (wib % WPF) * IPW + j, (uint32_t)rs * 4);
    const uint32_t fs4 = (uint32_t)fs * 4, d4 = (uint32_t)g.d * 4;
    // wave w stages frame (w / WPF) of the group: Q of that frame, K and V of the frame after it (nontemporal loads, aux = 2:
    // + 3 % here, + 8 % in the bf16 kernel, + 30 % in the bf16 block-attention forward; the backward kernels are indifferent)
    auto stage = [&](int buf, int fb) {
        const int i = wib / WPF;
        const int fq = min(fb + i, g.F - 1), fk = min(fb + i + 1, g.F - 1);    // clamped: such tiles are masked or unused
        char* dst = sm + buf * GROUP + i * 3 * TILE + (wib % WPF) * IPW * 1024;
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)gb, 0, span, 0x00020000);
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int j = 0; j < IPW; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void)(dst + t * TILE + j * 1024), 16, (int)voff[j],
                                                         (t ? fk : fq) * fs4 + t * d4, 0, 2);
    };

    stage(0, un.f0);
    // sliding window: K (row operand) and V (column operand) of frames f-1, f, f+1; the first two straight from memory
    f32x4v kw[3], vw[3];
    {
        const int fp = max(un.f0 - 1, 0);                        // frame 0 stands in when there is none: masked by `edge`
        kw[0] = load_row(qb + fp * fs + g.d, roff);
        vw[0] = load_col(qb + fp * fs + 2 * g.d, coff, rs);
        kw[1] = load_row(qb + un.f0 * fs + g.d, roff);
        vw[1] = load_col(qb + un.f0 * fs + 2 * g.d, coff, rs);
    }

    int buf = 0;
    for (int fb = un.f0; fb < un.f0 + g.seg; fb += PF, buf ^= 1) {
        wait_vm0();                                              // this wave's share of the group has landed ...
        wg_barrier();                                            // ... and everyone's; the other buffer is free again
        stage(buf ^ 1, fb + PF);
        const char* grp = sm + buf * GROUP;
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            const int f = fb + i;
            const char* tq = grp + i * 3 * TILE;
            const f32x4v q = ops.read_row(tq);
            kw[2] = ops.read_row(tq + TILE);
            vw[2] = ops.read_col(tq + 2 * TILE);
            if (f < un.f1 && live) {
                f32x4v s[3], e[3];
#pragma unroll
                for (int t = 0; t < 3; ++t) s[t] = dot_rows(kw[t], q);       // s[t][r] = S[q = lr][key = 4g + r]
                const float inv = band_exp(s, bias, f > 0, f + 1 < g.F, e);
                if constexpr (ADROP) {                           // WGATE.py:103 (on the numerators: 1 / sum is applied to O)
                    f32x4v keep[3];
                    band_keep(keep, ad, un.bw, g.nH, head, g.F, f, lr, gq);
#pragma unroll
                    for (int t = 0; t < 3; ++t) e[t] *= keep[t];
                }
                f32x4v oacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int t = 0; t < 3; ++t) oacc = mul_cols(vw[t], e[t], oacc);
                // lane (q = lr, g), reg r -> O[q][4g + r]; the row's 1 / sum is in this very lane
                // (plain stores: the four heads of a row arrive from four waves, and nontemporal ones, which the whole-row
                //  stores of blk_attn_f32.hip gain from, cost these 64-byte pieces 8 % forward and 18 % backward)
                *reinterpret_cast<f32x4v*>(ob + (int64_t)f * g.K * g.d + ooff) = oacc * inv;
            }
            kw[0] = kw[1]; kw[1] = kw[2];
            vw[0] = vw[1]; vw[1] = vw[2];
        }
    }
    wait_vm0();                                                  // the last, unused prefetch group must land before the LDS is released
}

// =============================================================== backward
// Staged per query frame: Q, dO of that frame and K, V of the frame after it (4 tiles of 4 heads); the column forms come
// from the same LDS images, so only P and dS pass through the wave-private transposing tiles (three tiles, P then dS).
template <int PF, int MINW, bool ADROP>
__global__ __launch_bounds__(256, MINW) void band_bwd_f32st_k(const float* __restrict__ qkv, const float* __restrict__ dO,
                                                              float* __restrict__ dqkv,
                                                              const uint64_t* __restrict__ maskrows, BandGeom g,
                                                              int64_t qkv_bytes, int64_t do_bytes, AttnDrop ad) {
    if constexpr (ADROP) ad.seed += seed_base_of(ad.base);
    constexpr int NHW = 4;                                       // (eight heads per workgroup would leave room for one workgroup per CU)
    using TL = Tile<NHW>;
    constexpr int TILE = TL::BYTES;
    constexpr int WPF = NHW / PF;
    constexpr int IPW = TL::NI / WPF;
    static_assert(PF == 1 || PF == 2, "frame group = 1 or 2 frames");
    constexpr int GROUP = PF * 4 * TILE;                         // one frame group: PF x (Q, K, V, dO)
    __shared__ __attribute__((aligned(1024))) char sm[2 * GROUP + 4 * 3 * XTILE];
    const int lane = threadIdx.x & 63, lr = lane & 15, gq = lane >> 4;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* xt = reinterpret_cast<float*>(sm + 2 * GROUP + wib * (3 * XTILE));   // [3 key tiles][q][XLD]
    const GroupF un = decode_group(g, blockIdx.x, NHW);
    const int head = 4 * un.hg + wib;
    const bool live = head < g.nH;
    const int hd_eff = min(head, g.nH - 1);
    const int64_t rs = 3 * (int64_t)g.d;
    const int64_t fs = (int64_t)g.K * rs, gs = (int64_t)g.K * g.d;
    const float* gqb = qkv + un.tok0 * rs + 4 * un.hg * HD;      // the group's columns: qkv ...
    const float* ggb = dO + un.tok0 * (int64_t)g.d + 4 * un.hg * HD;    // ... and dO
    const float* qb = qkv + un.tok0 * rs + hd_eff * HD;          // this wave's head (prologue loads)
    float* db = dqkv + un.tok0 * rs + hd_eff * HD;
    const uint32_t roff = lr * (uint32_t)rs + 4 * gq;            // lane offsets in qkv / dqkv: row operand / result ...
    const uint32_t coff = 4 * gq * (uint32_t)rs + lr;            // ... column operand
    float bias[3][4];
    band_bias(maskrows[un.w * 16 + lr], gq, bias);
    const Ops<NHW> ops(lane, wib);
    const int fa = max(un.f0 - 1, 0), fz = min(un.f1, g.F - 1);  // query frames fa .. fz (inclusive)

    const int span_q = (int)min(qkv_bytes - ((const char*)gqb - (const char*)qkv), (int64_t)0x7fffffff);
    const int span_g = (int)min(do_bytes - ((const char*)ggb - (const char*)dO), (int64_t)0x7fffffff);
    uint32_t voff_q[IPW], voff_g[IPW];
#pragma unroll
    for (int j = 0; j < IPW; ++j) {
        voff_q[j] = dma_src<NHW>(lane, (wib % WPF) * IPW + j, (uint32_t)rs * 4);
        voff_g[j] = dma_src<NHW>(lane, (wib % WPF) * IPW + j, (uint32_t)g.d * 4);
    }
    const uint32_t fs4 = (uint32_t)fs * 4, gs4 = (uint32_t)gs * 4, d4 = (uint32_t)g.d * 4;
    // wave w stages frame (w / WPF) of the group: Q, dO of that frame, K and V of the frame after it
    auto stage = [&](int buf, int fb) {
        const int i = wib / WPF;
        const int fq = min(fb + i, g.F - 1), fk = min(fb + i + 1, g.F - 1);
        char* dst = sm + buf * GROUP + i * 4 * TILE + (wib % WPF) * IPW * 1024;
        const auto rq = __builtin_amdgcn_make_buffer_rsrc((void*)gqb, 0, span_q, 0x00020000);
        const auto rg = __builtin_amdgcn_make_buffer_rsrc((void*)ggb, 0, span_g, 0x00020000);
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int j = 0; j < IPW; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rq, (lds_void)(dst + t * TILE + j * 1024), 16, (int)voff_q[j],
                                                         (t ? fk : fq) * fs4 + t * d4, 0, 0);
#pragma unroll
        for (int j = 0; j < IPW; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rg, (lds_void)(dst + 3 * TILE + j * 1024), 16, (int)voff_g[j], fq * gs4, 0, 0);
    };

    stage(0, fa);
    struct KeyFrame { f32x4v k, v, kc; };
    KeyFrame kw[3];
    {
        const int fp = max(fa - 1, 0);                           // frame 0 stands in when there is none: masked by `edge`
        kw[0].k = load_row(qb + fp * fs + g.d, roff);
        kw[0].v = load_row(qb + fp * fs + 2 * g.d, roff);
        kw[0].kc = load_col(qb + fp * fs + g.d, coff, rs);
        kw[1].k = load_row(qb + fa * fs + g.d, roff);
        kw[1].v = load_row(qb + fa * fs + 2 * g.d, roff);
        kw[1].kc = load_col(qb + fa * fs + g.d, coff, rs);
    }
    f32x4v dk[3], dv[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) { dk[t] = f32x4v{0.f, 0.f, 0.f, 0.f}; dv[t] = f32x4v{0.f, 0.f, 0.f, 0.f}; }

    auto store_key = [&](int f, const f32x4v& k, const f32x4v& v) {
        float* row = db + f * fs + roff;                         // lane (key = lr, g), reg r -> [key][4g + r]
        *reinterpret_cast<f32x4v*>(row + g.d) = k * band_scale<HD>();
        *reinterpret_cast<f32x4v*>(row + 2 * g.d) = v;
    };
    // transposed tile: lane (q = lr, g) wrote x[q][4g .. 4g+3]; lane (key = lr, g) reads x[q = 4g + r][key = lr]
    auto xpose3 = [&](const f32x4v (&x)[3], f32x4v (&y)[3]) {
#pragma unroll
        for (int t = 0; t < 3; ++t) *reinterpret_cast<f32x4v*>(xt + (t * 16 + lr) * XLD + 4 * gq) = x[t];
        wave_fence();
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) y[t][r] = xt[(t * 16 + 4 * gq + r) * XLD + lr];
        wave_fence();
    };

    // every workgroup runs the same number of rounds (seg + 2 query frames at most)
    int buf = 0;
    for (int it = 0; it < g.seg + 2; it += PF, buf ^= 1) {
        wait_vm0();                                              // this wave's share of the group has landed ...
        wg_barrier();                                            // ... and everyone's; the other buffer is free again
        stage(buf ^ 1, fa + it + PF);
        const char* grp = sm + buf * GROUP;
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            const int f = fa + it + i;
            const char* tq = grp + i * 4 * TILE;
            const f32x4v q = ops.read_row(tq), go = ops.read_row(tq + 3 * TILE);
            const f32x4v qc = ops.read_col(tq), gc = ops.read_col(tq + 3 * TILE);
            kw[2].k = ops.read_row(tq + TILE);
            kw[2].v = ops.read_row(tq + 2 * TILE);
            kw[2].kc = ops.read_col(tq + TILE);
            if (f <= fz && live) {
                const bool hp = f > 0, hn = f + 1 < g.F;
                // ---- lane = query joint lr, registers = key joints 4g + r
                f32x4v s[3], p[3], ds[3];
#pragma unroll
                for (int t = 0; t < 3; ++t) s[t] = dot_rows(kw[t].k, q);
                const float inv = band_exp(s, bias, hp, hn, p);
                // attention dropout: A = D o P went into O = A V, so dP = D o dA (dA = dO V^T) and dV = A^T dO; mask recomputed
                f32x4v keep[ADROP ? 3 : 1];
                if constexpr (ADROP) band_keep(keep, ad, un.bw, g.nH, head, g.F, f, lr, gq);
                float delta = 0.f;
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    p[t] *= inv;
                    ds[t] = dot_rows(kw[t].v, go);                             // dP[q = lr][key = 4g + r]
                    if constexpr (ADROP) ds[t] *= keep[t];
#pragma unroll
                    for (int r = 0; r < 4; ++r) delta = __builtin_fmaf(p[t][r], ds[t][r], delta);
                }
                delta = xg_sum(delta);
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    ds[t] = p[t] * (ds[t] - delta);                            // dS (the score scale goes on dq / dk)
                    if constexpr (ADROP) p[t] *= keep[t];                      // the transposed P feeds dV only: A = D o P
                }
                // dQ[q = lr][4g + r] = scale * sum_key dS[q][key] K[key][c]
                if (f >= un.f0 && f < un.f1) {
                    f32x4v acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int t = 0; t < 3; ++t) acc = mul_cols(kw[t].kc, ds[t], acc);
                    *reinterpret_cast<f32x4v*>(db + f * fs + roff) = acc * band_scale<HD>();
                }
                // ---- lane = key joint lr, registers = query joints 4g + r
                f32x4v p2[3], ds2[3];
                xpose3(p, p2);
                xpose3(ds, ds2);
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    dk[t] = mul_cols(qc, ds2[t], dk[t]);                       // dK[key][c] += sum_q dS[q][key] Q[q][c]
                    dv[t] = mul_cols(gc, p2[t], dv[t]);                        // dV[key][c] += sum_q P[q][key] dO[q][c]
                }
                // key frame f-1 has now seen query frames f-2, f-1, f: done
                if (hp && f - 1 >= un.f0) store_key(f - 1, dk[0], dv[0]);      // (f - 1 < f1 always: f <= f1)
                dk[0] = dk[1]; dk[1] = dk[2]; dk[2] = f32x4v{0.f, 0.f, 0.f, 0.f};
                dv[0] = dv[1]; dv[1] = dv[2]; dv[2] = f32x4v{0.f, 0.f, 0.f, 0.f};
                kw[0] = kw[1]; kw[1] = kw[2];
            }
        }
    }
    wait_vm0();                                                  // the last, unused prefetch group must land before the LDS is released
    // the clip's last key frame has no query frame after it: after the rotation it sits in slot 0
    if (live && un.f1 == g.F) store_key(g.F - 1, dk[0], dv[0]);
}

// frame segments per clip so that the grid holds `want` wavefronts; segments of at least `min_seg` frames
int segments(int64_t base_units, int F, int64_t want, int min_seg, const char* lab_name) {
    int n_seg = 1;
    while (base_units * n_seg < want && F / (n_seg * 2) >= min_seg) n_seg *= 2;
    if (const char* e = lab_env(lab_name)) n_seg = max(1, atoi(e));
    return n_seg;
}

}  // namespace

int hwgat_launch_band_fwd_f32(const void* qkv, void* o, const uint64_t* maskrows, int B, int F, int nW, int nH,
                              uint32_t drop_seed, float drop_p, const uint32_t* seed_base, hipStream_t st) {
    const AttnDrop ad = make_drop(drop_seed, drop_p, seed_base);
    const int64_t base_units = (int64_t)B * nW * nH;
    int n_seg = segments(base_units, F, 256 * 4 * 2, 16, "HWGAT_BAND_FSEG");
    const int seg = (F + n_seg - 1) / n_seg;
    n_seg = (F + seg - 1) / seg;
    BandGeom g{F, nW * 16, nW, nH, nH * HD, seg, n_seg};
    const int64_t clip_bytes = (int64_t)F * nW * 16 * 3 * nH * HD * 4;
    // eight heads per workgroup where the head count allows it: with d = 128 the tile rows are then whole q / k / v rows, the
    // 16 joints of a window one contiguous 24 KB piece of qkv per frame (LABLOG 10.9)
    const int nhw = lab_env("HWGAT_BAND_NHW") ? atoi(lab_env("HWGAT_BAND_NHW")) : (nH % 8 == 0 ? 8 : 4);
    const int64_t blocks = (int64_t)B * nW * n_seg * ((nH + nhw - 1) / nhw);
    if (blocks > 0x7fffffff || clip_bytes > 0x7fffffff) return HWGAT_ESHAPE;
    const int64_t bytes = clip_bytes * B;
    const int pf = lab_env("HWGAT_BAND_PF") ? atoi(lab_env("HWGAT_BAND_PF")) : (nhw == 8 ? 1 : 2);
#define FWD(NHW, PF, MINW, MINWD)                                                                                                 \
    do {                                                                                                                          \
        if (ad.p > 0.f) band_fwd_f32st_k<NHW, PF, MINWD, true><<<(int)blocks, NHW * 64, 0, st>>>((const float*)qkv, (float*)o, maskrows, g, bytes, ad); \
        else band_fwd_f32st_k<NHW, PF, MINW, false><<<(int)blocks, NHW * 64, 0, st>>>((const float*)qkv, (float*)o, maskrows, g, bytes, ad);  \
    } while (0)
    if (nhw == 8) { if (pf == 2) FWD(8, 2, 2, 2); else FWD(8, 1, 6, 5); }    // 96 KB: one workgroup per CU / 48 KB: three
    else if (pf == 1) FWD(4, 1, 4, 4);
    else FWD(4, 2, 3, 3);
#undef FWD
    HWGAT_LAUNCH_CHECK();
}

int hwgat_launch_band_bwd_f32(const void* qkv, const void* dO, void* dqkv, const uint64_t* maskrows, int B, int F, int nW,
                              int nH, uint32_t drop_seed, float drop_p, const uint32_t* seed_base, hipStream_t st) {
    const AttnDrop ad = make_drop(drop_seed, drop_p, seed_base);
    const int64_t base_units = (int64_t)B * nW * nH;
    int n_seg = segments(base_units, F, 256 * 4 * 2, 16, "HWGAT_BAND_BSEG");
    const int seg = (F + n_seg - 1) / n_seg;
    n_seg = (F + seg - 1) / seg;
    BandGeom g{F, nW * 16, nW, nH, nH * HD, seg, n_seg};
    const int64_t clip_bytes = (int64_t)F * nW * 16 * 3 * nH * HD * 4;
    const int64_t blocks = (int64_t)B * nW * n_seg * ((nH + 3) / 4);
    if (blocks > 0x7fffffff || clip_bytes > 0x7fffffff) return HWGAT_ESHAPE;
    const int pf = lab_env("HWGAT_BAND_PF") ? atoi(lab_env("HWGAT_BAND_PF")) : 1;
#define BWD(PF, MINW)                                                                                                              \
    do {                                                                                                                          \
        if (ad.p > 0.f)                                                                                                           \
            band_bwd_f32st_k<PF, MINW, true><<<(int)blocks, 256, 0, st>>>((const float*)qkv, (const float*)dO, (float*)dqkv, maskrows, g, clip_bytes * B, clip_bytes * B / 3, ad); \
        else                                                                                                                      \
            band_bwd_f32st_k<PF, MINW, false><<<(int)blocks, 256, 0, st>>>((const float*)qkv, (const float*)dO, (float*)dqkv, maskrows, g, clip_bytes * B, clip_bytes * B / 3, ad); \
    } while (0)
    if (pf == 2) BWD(2, 1);                                      // 80 KB of LDS: one workgroup per CU (lab A/B)
    else BWD(1, 3);                                              // 47 KB: three
#undef BWD
    HWGAT_LAUNCH_CHECK();
}
