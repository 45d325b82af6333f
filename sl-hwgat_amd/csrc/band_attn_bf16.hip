// Fused band graph-attention for the WGATE sibling model, bf16 storage, on gfx950 (MI355X).
//
// Same contract as band_attn.hip (MSA.forward's attention core of the reference's hwgat/models/WGATE.py:87-108 with
// window_partition / window_reverse, WGATE.py:32-65, folded into index arithmetic; block-tridiagonal 0 / -10000 mask of
// model_params.py:209-228 as 48-bit rows), but every product runs on v_mfma_f32_16x16x16_bf16 instead of four
// v_mfma_f32_16x16x4_f32, and nothing is converted to fp32 on the way in:
//
//   workgroup = (clip, part window, frame segment, 4 neighbouring heads); wave w owns head 4 hg + w and walks the frames
//          of the segment in order.  The workgroup fetches the q / k / v (/ dO) tiles of its frames TOGETHER, as whole
//          128 / 256-byte row pieces, by LDS-DMA one frame group ahead (see "workgroup-staged tiles" below): a wave
//          fetching only its own head's 32-byte pieces holds HBM at 3.6 TB/s, whole lines reach 5.1 TB/s.
//   row operand  X[row = l&15][16 ch + 4 (l>>4) + 0..3]: one ds_read_b64 per lane and 16 channels, the A or B operand of a
//          head-dim contraction (S^T = K Q^T, dP^T = V dO^T).
//   column operand X[row = 4 (l>>4) + 0..3][16 ct + (l&15)]: the A operand of every contraction over tokens
//          (O^T = V^T P^T, dQ^T = K^T dS^T, dK^T = Q^T dS, dV^T = dO^T P): one ds_read_b64_tr_b16 from the SAME staged
//          image -- no second, element-wise global load of the same data as in the fp32 kernel.
//   results come out of the MFMA as [channel 4g + r][token l&15] = four consecutive channels of one token per lane: one
//          8-byte store per lane and 16 channels for o, dq, dk, dv.
//   softmax in fp32 registers (base-2 exponent, visibility as an additive 0 / -inf bias, cross-lane steps with
//          v_permlane16/32_swap); P and dS are rounded to bf16 as MFMA operands (what a bf16 matmul of the reference
//          does with them); scores are scaled in fp32 AFTER the product, so q is not rounded a second time.
//   backward: P / dS are transposed through wave-private 16 x 16 LDS tiles (8-byte write, one transposed read); the clip
//          may be cut into frame segments with a one-query-frame halo on either side (recomputed, not stored).
//
// HBM traffic stays the algorithmic 4 E s (fwd) / 7 E s (bwd) plus the segment halos.
#include <stdlib.h>
#include "band_common.h"

namespace {
using namespace band;

typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
typedef __attribute__((address_space(3))) u32x2v lds_u32x2;

constexpr int TROW = 40;                 // bytes per row of a transposing tile: 16 bf16 + 8 pad (8-byte aligned rows)
constexpr int TTILE = 16 * TROW;         // one 16 x 16 tile
constexpr float NEG_INF = -__builtin_inff();

// D(16x16) += A(16x16) B(16x16): lane l supplies A[i = l&15][k = 4 (l>>4) + e] and B[k = 4 (l>>4) + e][j = l&15], e = 0..3;
// register r of lane l is D[i = 4 (l>>4) + r][j = l&15].
// (operands travel as two dwords = 4 packed bf16: a bf16x4 bit-cast at the MFMA makes the compiler convert one value
// per instruction and merge with v_perm)
typedef u32x2v pk4;
__device__ __forceinline__ f32x4v mfma_bf(pk4 a, pk4 b, f32x4v c) {
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, a), __builtin_bit_cast(s16x4, b), c, 0, 0, 0);
}

__device__ __forceinline__ uint32_t pk2(float a, float b) {           // round to nearest even, v_cvt_pk_bf16_f32
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{a, b}, bf16x2));
}
__device__ __forceinline__ pk4 to_bf(const f32x4v& v) { return pk4{pk2(v.x, v.y), pk2(v.z, v.w)}; }

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

template <int NC> struct Row16 { pk4 c[NC]; };     // X[row = l&15][16 ch + 4 g + e]
template <int NC> struct Col16 { pk4 v[NC]; };     // X[row = 4 g + e][16 ct + (l&15)]

// `base` is wave-uniform, `off` a 32-bit per-lane element offset (scalar base + vector offset addressing)
template <int NC> __device__ __forceinline__ Row16<NC> load_row16(const bf16_t* base, uint32_t off) {
    Row16<NC> t;
#pragma unroll
    for (int ch = 0; ch < NC; ++ch) t.c[ch] = *reinterpret_cast<const pk4*>(base + off + 16 * ch);
    return t;
}
template <int NC> __device__ __forceinline__ Row16<NC> zero_row16() {
    Row16<NC> t;
#pragma unroll
    for (int ch = 0; ch < NC; ++ch) t.c[ch] = pk4{0u, 0u};
    return t;
}

// D[i][j] = sum_c X[i][c] Y[j][c] of two row operands: lane (j = l&15, g), register r -> D[4g + r][j]
template <int NC> __device__ __forceinline__ f32x4v dot_rows16(const Row16<NC>& x, const Row16<NC>& y) {
    f32x4v acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ch = 0; ch < NC; ++ch) acc = mfma_bf(x.c[ch], y.c[ch], acc);
    return acc;
}
// acc[ct][r] (lane j) += sum_k Y[k][16 ct + 4g + r] a[j][k]: Y a column operand over the 16 tokens k, a[e] = A[j = l&15][k = 4g + e]
template <int NC> __device__ __forceinline__ void mul_cols16(const Col16<NC>& y, pk4 a, f32x4v (&acc)[NC]) {
#pragma unroll
    for (int ct = 0; ct < NC; ++ct) acc[ct] = mfma_bf(y.v[ct], a, acc[ct]);
}

// wave-private transposing tile: lane (row = l&15, g) writes its 4 elements of columns 4g .. 4g+3; lane (col = l&15, g)
// reads back the elements of rows 4g .. 4g+3 of its column (ds_read_b64_tr_b16: 16-lane group g takes the 4 x 16 block of
// rows 4g .. 4g+3; lane 4q + p of the group supplies the address of row 4g + q, columns 4p .. 4p+3)
struct TileXpose {
    uint32_t put, get;
    __device__ __forceinline__ TileXpose(int lane) {
        const int lr = lane & 15, g = lane >> 4;
        put = lr * TROW + g * 8;
        get = (4 * g + (lr >> 2)) * TROW + (lr & 3) * 8;
    }
    __device__ __forceinline__ void write(char* tile, pk4 v) const { *(lds_u32x2*)(tile + put) = v; }
    __device__ __forceinline__ pk4 read(char* tile) const {
        return __builtin_bit_cast(pk4, __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(tile + get)));
    }
};
__device__ __forceinline__ void wave_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
template <int NC> __device__ __forceinline__ Col16<NC> to_col(const TileXpose& x, char* tiles, const Row16<NC>& r) {
    Col16<NC> c;
#pragma unroll
    for (int ch = 0; ch < NC; ++ch) x.write(tiles + ch * TTILE, r.c[ch]);
    wave_fence();
#pragma unroll
    for (int ch = 0; ch < NC; ++ch) c.v[ch] = x.read(tiles + ch * TTILE);
    wave_fence();
    return c;
}

// unit = ((clip, window), segment, head), head fastest: the 4 waves of a workgroup are neighbouring heads of one segment
struct Unit16 {
    int64_t tok0;          // token index of (clip, frame 0, first joint of the window)
    int head, w, f0, f1;   // owned frames [f0, f1)
};
__device__ __forceinline__ Unit16 decode16(const BandGeom& g, int u) {
    Unit16 r;
    r.head = u % g.nH;
    int t = u / g.nH;
    const int sgi = t % g.n_seg;
    t /= g.n_seg;
    r.w = t % g.nW;
    const int b = t / g.nW;
    r.tok0 = (int64_t)b * g.F * g.K + r.w * 16;
    r.f0 = sgi * g.seg;
    r.f1 = min(g.F, r.f0 + g.seg);
    return r;
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
template <int HD>
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
    return __builtin_amdgcn_rcpf(xg_sum(sum));
}

// =============================================================== workgroup-staged tiles
// One head's piece of a token row is 32 (hd 16) or 64 bytes (hd 32): a wavefront that fetches its own head reads 16
// such pieces per load instruction, and HBM delivers 3.6 TB/s to that pattern against 5.1 TB/s for whole 128-byte lines
// (profiles/r03_band_bf16_lab.txt).  So a workgroup = 4 neighbouring heads fetches the q / k / v tiles of its frames TOGETHER:
// 16 token rows x (4 heads x hd) = rows of RB = 128 / 256 contiguous bytes, moved by LDS-DMA (16 bytes per lane, 1 KB per
// instruction, no registers) one frame group ahead into a double buffer; each wave then reads its head's operands out of
// LDS: row operands with ds_read_b64, column operands with ds_read_b64_tr_b16 straight from the same image.
// The DMA writes LDS lane-linearly, so the bank swizzle is applied on the SOURCE side: LDS 16-byte slot c' of row r holds
// the row's chunk c' ^ x(r), x = the row-pair index (hd 16) / row index (hd 32) rotated left by one bit: both kinds of
// read are then conflict-free.
template <int HD> struct Staged {
    static constexpr int RB = 4 * HD * 2;                    // bytes per token row of a 4-head tile
    static constexpr int TILE = 16 * RB;                     // 2 KB / 4 KB
    static constexpr int NI = TILE / 1024;                   // DMA instructions per tile: 2 / 4
    static constexpr int RPI = 16 / NI;                      // token rows per DMA instruction: 8 / 4
    __device__ static __forceinline__ int xr(int row) {
        if constexpr (HD == 16) { const int h = (row >> 1) & 7; return ((h & 3) << 1) | (h >> 2); }
        else { const int h = row & 15; return ((h & 7) << 1) | (h >> 3); }
    }
    // byte offset inside a tile of the 8-byte piece p8 (0..3) of channels 16 ch .. 16 ch + 15 of head w in token row `row`
    __device__ static __forceinline__ uint32_t piece(int row, int w, int ch, int p8) {
        const int c = w * (HD / 8) + 2 * ch + (p8 >> 1);
        return row * RB + ((c ^ xr(row)) << 4) + (p8 & 1) * 8;
    }
    // source byte offset (token rows of rs2 bytes) of this lane's 16 bytes in DMA instruction `ins` of a tile
    __device__ static __forceinline__ uint32_t src(int lane, int ins, uint32_t rs2) {
        const int row = ins * RPI + (lane * 16) / RB, cp = ((lane * 16) % RB) >> 4;
        return row * rs2 + ((cp ^ xr(row)) << 4);
    }
};

typedef __attribute__((address_space(3))) void* lds_void;
__device__ __forceinline__ void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void wg_barrier() { asm volatile("s_barrier" ::: "memory"); }

// unit = ((clip, window), segment, group of 4 heads): one workgroup; wave w of it owns head 4 hg + w
struct Group16 {
    int64_t tok0;
    int hg, w, f0, f1;
    int bw;                // clip * nW + window (see BandUnit)
};
__device__ __forceinline__ Group16 decode_group(const BandGeom& g, int blk) {
    Group16 r;
    const int n_hg = (g.nH + 3) >> 2;
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
// DBG (kernel lab only): 1 = memory only (stage, read, store; no products, no softmax)
template <int HD, int PF, int MINW, int DBG = 0, bool ADROP = false>
__global__ __launch_bounds__(256, MINW) void band_fwd_st_k(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ o,
                                                           const uint64_t* __restrict__ maskrows, BandGeom g,
                                                           int64_t qkv_bytes, AttnDrop ad) {
    if constexpr (ADROP) ad.seed += seed_base_of(ad.base);
    constexpr int NC = HD / 16;
    using St = Staged<HD>;
    constexpr int WPF = 4 / PF;                                  // waves sharing the DMA work of one frame
    constexpr int IPW = St::NI / WPF;                            // DMA instructions per tile and wave
    static_assert(PF == 4 || PF == 2, "frame group = 4 or 2 frames");
    static_assert(St::NI % WPF == 0, "tile instructions split evenly over the waves of a frame");
    constexpr int GROUP = PF * 3 * St::TILE;                     // one frame group: PF x (Q, K, V)
    __shared__ __attribute__((aligned(1024))) char sm[2 * GROUP + 4 * NC * TTILE];   // [2][PF][3][TILE] | transposing tiles (prologue)
    const int lane = threadIdx.x & 63, lr = lane & 15, gq = lane >> 4;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char* vt = sm + 2 * GROUP + wib * (NC * TTILE);
    const TileXpose xp(lane);
    const Group16 un = decode_group(g, blockIdx.x);
    const int head = 4 * un.hg + wib;
    const bool live = head < g.nH;                               // waves past the last head stage tiles but compute nothing
    const int hd_eff = min(head, g.nH - 1);
    const int64_t rs = 3 * (int64_t)g.d;                         // qkv row stride (elements)
    const bf16_t* gb = qkv + un.tok0 * rs + 4 * un.hg * HD;      // the group's q columns of (frame 0, joint 0)
    const bf16_t* qb = qkv + un.tok0 * rs + hd_eff * HD;         // this wave's head (prologue loads)
    bf16_t* ob = o + un.tok0 * (int64_t)g.d + hd_eff * HD;
    const int64_t fs = (int64_t)g.K * rs;                        // frame stride in qkv
    const uint32_t roff = lr * (uint32_t)rs + 4 * gq;            // per-lane offsets in qkv ...
    const uint32_t ooff = lr * (uint32_t)g.d + 4 * gq;           // ... and in o
    float bias[3][4];
    band_bias(maskrows[un.w * 16 + lr], gq, bias);

    // DMA: resource = from the group's first byte to the end of the tensor (lanes past it read zeros: a last head group
    // of fewer than 4 heads stages columns that belong to no head)
    const int64_t left = qkv_bytes - ((const char*)gb - (const char*)qkv);
    const int span = (int)min(left, (int64_t)0x7fffffff);
    uint32_t voff[IPW];
#pragma unroll
    for (int j = 0; j < IPW; ++j) voff[j] = St::src(lane, (wib % WPF) * IPW + j, (uint32_t)rs * 2);
    const uint32_t fs2 = (uint32_t)fs * 2, d2 = (uint32_t)g.d * 2;
    // wave w stages frame (w / WPF) of the group: Q of that frame, K and V of the frame after it
    auto stage = [&](int buf, int fb) {
        const int i = wib / WPF;
        const int fq = min(fb + i, g.F - 1), fk = min(fb + i + 1, g.F - 1);    // clamped: such tiles are masked or unused
        char* dst = sm + buf * GROUP + i * 3 * St::TILE + (wib % WPF) * IPW * 1024;
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)gb, 0, span, 0x00020000);
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int j = 0; j < IPW; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void)(dst + t * St::TILE + j * 1024), 16, (int)voff[j],   // (the cast matters: without it clang drops the host stub of this kernel)
                                                         (t ? fk : fq) * fs2 + t * d2, 0, 2);
    };
    // this lane's read offsets inside a tile
    uint32_t r_row[NC], r_col[NC];
#pragma unroll
    for (int ch = 0; ch < NC; ++ch) {
        r_row[ch] = St::piece(lr, wib, ch, gq);
        r_col[ch] = St::piece(4 * gq + (lr >> 2), wib, ch, lr & 3);
    }
    auto read_row = [&](const char* tile) {
        Row16<NC> t;
#pragma unroll
        for (int ch = 0; ch < NC; ++ch) t.c[ch] = *(const lds_u32x2*)(tile + r_row[ch]);
        return t;
    };
    auto read_col = [&](const char* tile) {
        Col16<NC> t;
#pragma unroll
        for (int ch = 0; ch < NC; ++ch)
            t.v[ch] = __builtin_bit_cast(pk4, __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(tile + r_col[ch])));
        return t;
    };

    stage(0, un.f0);
    // sliding window: K (row operand) and V (column operand) of frames f-1, f, f+1; the first two straight from memory
    Row16<NC> kw[3];
    Col16<NC> vw[3];
    kw[0] = zero_row16<NC>();
    {
        const int fp = max(un.f0 - 1, 0);                        // frame 0 stands in when there is none: masked by `edge`
        const Row16<NC> vp = load_row16<NC>(qb + fp * fs + 2 * g.d, roff);
        if (un.f0 > 0) kw[0] = load_row16<NC>(qb + fp * fs + g.d, roff);
        vw[0] = to_col<NC>(xp, vt, vp);
        kw[1] = load_row16<NC>(qb + un.f0 * fs + g.d, roff);
        vw[1] = to_col<NC>(xp, vt, load_row16<NC>(qb + un.f0 * fs + 2 * g.d, roff));
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
            const char* tq = grp + i * 3 * St::TILE;
            const Row16<NC> q = read_row(tq);
            kw[2] = read_row(tq + St::TILE);
            vw[2] = read_col(tq + 2 * St::TILE);                 // outside the branch: the transposed read wants all lanes
            if (DBG == 1) {
                if (f < un.f1 && live) {
                    bf16_t* of = ob + (int64_t)f * g.K * g.d;
#pragma unroll
                    for (int ct = 0; ct < NC; ++ct) *reinterpret_cast<pk4*>(of + ooff + 16 * ct) = q.c[ct] ^ kw[2].c[ct] ^ vw[2].v[ct];
                }
            } else if (f < un.f1 && live) {
                f32x4v s[3], e[3];
#pragma unroll
                for (int t = 0; t < 3; ++t) s[t] = dot_rows16<NC>(kw[t], q);   // s[t][r] = S[q = lr][key = 4g + r]
                const float inv = band_exp<HD>(s, bias, f > 0, f + 1 < g.F, e);
                if constexpr (ADROP) {                           // WGATE.py:103 (on the numerators: 1 / sum is applied to O)
                    f32x4v keep[3];
                    band_keep(keep, ad, un.bw, g.nH, head, g.F, f, lr, gq);
#pragma unroll
                    for (int t = 0; t < 3; ++t) e[t] *= keep[t];
                }
                f32x4v oacc[NC];
#pragma unroll
                for (int ct = 0; ct < NC; ++ct) oacc[ct] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int t = 0; t < 3; ++t) mul_cols16<NC>(vw[t], to_bf(e[t]), oacc);
                // lane (q = lr, g), reg r -> O[q][16 ct + 4g + r]; the row's 1 / sum is in this very lane
                bf16_t* of = ob + (int64_t)f * g.K * g.d;
#pragma unroll
                for (int ct = 0; ct < NC; ++ct)
                    *reinterpret_cast<pk4*>(of + ooff + 16 * ct) = to_bf(oacc[ct] * inv);
            }
            kw[0] = kw[1]; kw[1] = kw[2];
            vw[0] = vw[1]; vw[1] = vw[2];
        }
    }
    wait_vm0();                                                  // the last, unused prefetch group must land before the LDS is released
}

#ifdef HWGAT_LAB
// ---- first form (lab A/B only): every wave fetches its own head (32-byte row pieces) into a register ring; kept as the
// reference point of profiles/r03_band_bf16_lab.txt
// DBG (kernel lab only): 1 = memory only (loads, transposes, stores; no products, no softmax), 2 = products without the
// softmax, 3 = no loads inside the frame loop
template <int HD, int PF, int MINW, int DBG = 0>
__global__ __launch_bounds__(256, MINW) void band_fwd_b16_k(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ o,
                                                            const uint64_t* __restrict__ maskrows, BandGeom g,
                                                            int n_units) {
    constexpr int NC = HD / 16;
    __shared__ __attribute__((aligned(16))) char sm[4 * NC * TTILE];
    const int lane = threadIdx.x & 63, lr = lane & 15, gq = lane >> 4;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char* vt = sm + wib * (NC * TTILE);
    const TileXpose xp(lane);
    const int u_raw = blockIdx.x * 4 + wib;
    const bool live = u_raw < n_units;                           // tail waves shadow the last unit (no stores): every
    const int u = live ? u_raw : n_units - 1;                    // wave of a workgroup reaches the per-group barrier
    const Unit16 un = decode16(g, u);
    const int64_t rs = 3 * (int64_t)g.d;                         // qkv row stride (elements)
    const bf16_t* qb = qkv + un.tok0 * rs + un.head * HD;
    bf16_t* ob = o + un.tok0 * (int64_t)g.d + un.head * HD;
    const int64_t fs = (int64_t)g.K * rs;                        // frame stride in qkv
    const uint32_t roff = lr * (uint32_t)rs + 4 * gq;            // per-lane offsets in qkv ...
    const uint32_t ooff = lr * (uint32_t)g.d + 4 * gq;           // ... and in o
    float bias[3][4];
    band_bias(maskrows[un.w * 16 + lr], gq, bias);

    // sliding window: K (row operand) and V (column operand) of frames f-1, f, f+1
    Row16<NC> kw[3];
    Col16<NC> vw[3];
    kw[0] = zero_row16<NC>();
    {
        const int fp = max(un.f0 - 1, 0);                        // frame 0 stands in when there is none: masked by `edge`
        const Row16<NC> vp = load_row16<NC>(qb + fp * fs + 2 * g.d, roff);
        if (un.f0 > 0) kw[0] = load_row16<NC>(qb + fp * fs + g.d, roff);
        vw[0] = to_col<NC>(xp, vt, vp);
        kw[1] = load_row16<NC>(qb + un.f0 * fs + g.d, roff);
        vw[1] = to_col<NC>(xp, vt, load_row16<NC>(qb + un.f0 * fs + 2 * g.d, roff));
    }

    // prefetch ring: slot i holds Q of frame f+i and K, V of frame f+i+1
    Row16<NC> rq[PF], rk[PF], rv[PF];
    // loads are unconditional (a branch around a load makes the compiler drain the whole ring): frames past the clip
    // re-read the last frame; such tiles are either never consumed or masked out
    auto fill = [&](int i, int f) {                              // f = query frame of the slot
        const int fq = min(f, g.F - 1), fk = min(f + 1, g.F - 1);
        rq[i] = load_row16<NC>(qb + fq * fs, roff);
        rk[i] = load_row16<NC>(qb + fk * fs + g.d, roff);
        rv[i] = load_row16<NC>(qb + fk * fs + 2 * g.d, roff);
    };
#pragma unroll
    for (int i = 0; i < PF; ++i) fill(i, un.f0 + i);

    for (int fb = un.f0; fb < un.f0 + g.seg; fb += PF) {
        // neighbouring heads stay within PF frames of each other: the line they share is touched while still cached
        __syncthreads();
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            const int f = fb + i;
            const Row16<NC> q = rq[i];
            kw[2] = rk[i];
            vw[2] = to_col<NC>(xp, vt, rv[i]);                   // outside the branch: the transposed read wants all lanes
            if (DBG != 3) fill(i, f + PF);
            if (DBG == 1) {
                if (f < un.f1 && live) {
                    bf16_t* of = ob + (int64_t)f * g.K * g.d;
#pragma unroll
                    for (int ct = 0; ct < NC; ++ct) *reinterpret_cast<pk4*>(of + ooff + 16 * ct) = q.c[ct] ^ kw[2].c[ct] ^ vw[2].v[ct];
                }
            } else if (f < un.f1 && live) {
                f32x4v s[3], e[3];
#pragma unroll
                for (int t = 0; t < 3; ++t) s[t] = dot_rows16<NC>(kw[t], q);   // s[t][r] = S[q = lr][key = 4g + r]
                float inv = 1.f;
                if (DBG == 2) { e[0] = s[0]; e[1] = s[1]; e[2] = s[2]; }
                else inv = band_exp<HD>(s, bias, f > 0, f + 1 < g.F, e);
                f32x4v oacc[NC];
#pragma unroll
                for (int ct = 0; ct < NC; ++ct) oacc[ct] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int t = 0; t < 3; ++t) mul_cols16<NC>(vw[t], to_bf(e[t]), oacc);
                // lane (q = lr, g), reg r -> O[q][16 ct + 4g + r]; the row's 1 / sum is in this very lane
                bf16_t* of = ob + (int64_t)f * g.K * g.d;
#pragma unroll
                for (int ct = 0; ct < NC; ++ct)
                    *reinterpret_cast<pk4*>(of + ooff + 16 * ct) = to_bf(oacc[ct] * inv);
            }
            kw[0] = kw[1]; kw[1] = kw[2];
            vw[0] = vw[1]; vw[1] = vw[2];
        }
    }
}

template <int HD, int PF, int MINW>
__global__ __launch_bounds__(256, MINW) void band_bwd_b16_k(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dO,
                                                            bf16_t* __restrict__ dqkv,
                                                            const uint64_t* __restrict__ maskrows, BandGeom g,
                                                            int n_units) {
    constexpr int NC = HD / 16;
    constexpr int NT = 3 * NC + 6;                               // tiles per wave: Q^T, dO^T, K^T chunks | P x 3 | dS x 3
    __shared__ __attribute__((aligned(16))) char sm[4 * NT * TTILE];
    const int lane = threadIdx.x & 63, lr = lane & 15, gq = lane >> 4;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char* tq = sm + wib * (NT * TTILE);
    char* tg = tq + NC * TTILE;
    char* tk = tg + NC * TTILE;
    char* tp = tk + NC * TTILE;
    char* td = tp + 3 * TTILE;
    const TileXpose xp(lane);
    const int u_raw = blockIdx.x * 4 + wib;
    const bool live = u_raw < n_units;                           // tail waves shadow the last unit without storing
    const int u = live ? u_raw : n_units - 1;
    const Unit16 un = decode16(g, u);
    const int64_t rs = 3 * (int64_t)g.d;
    const int64_t fs = (int64_t)g.K * rs, gs = (int64_t)g.K * g.d;
    const bf16_t* qb = qkv + un.tok0 * rs + un.head * HD;
    const bf16_t* gb = dO + un.tok0 * (int64_t)g.d + un.head * HD;
    bf16_t* db = dqkv + un.tok0 * rs + un.head * HD;
    const uint32_t roff = lr * (uint32_t)rs + 4 * gq;            // lane offsets in qkv / dqkv ...
    const uint32_t groff = lr * (uint32_t)g.d + 4 * gq;          // ... and in dO
    float bias[3][4];
    band_bias(maskrows[un.w * 16 + lr], gq, bias);
    const int fa = max(un.f0 - 1, 0), fz = min(un.f1, g.F - 1);  // query frames fa .. fz (inclusive)

    struct KeyFrame { Row16<NC> k, v; Col16<NC> kc; };
    KeyFrame kw[3];
    {
        const int fp = max(fa - 1, 0);
        kw[0].k = zero_row16<NC>();
        kw[0].v = zero_row16<NC>();
        if (fa > 0) {
            kw[0].k = load_row16<NC>(qb + fp * fs + g.d, roff);
            kw[0].v = load_row16<NC>(qb + fp * fs + 2 * g.d, roff);
        }
        kw[0].kc = to_col<NC>(xp, tk, kw[0].k);
        kw[1].k = load_row16<NC>(qb + fa * fs + g.d, roff);
        kw[1].v = load_row16<NC>(qb + fa * fs + 2 * g.d, roff);
        kw[1].kc = to_col<NC>(xp, tk, kw[1].k);
    }
    f32x4v dk[3][NC], dv[3][NC];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int ct = 0; ct < NC; ++ct) { dk[t][ct] = f32x4v{0.f, 0.f, 0.f, 0.f}; dv[t][ct] = f32x4v{0.f, 0.f, 0.f, 0.f}; }

    // prefetch ring: slot i = query frame f+i: Q, dO and K, V of key frame f+i+1 (all as row operands)
    struct Slot { Row16<NC> q, go, k, v; };
    Slot ring[PF];
    auto fill = [&](int i, int f) {                              // unconditional, clamped (see the forward kernel)
        const int fq = min(f, g.F - 1), fk = min(f + 1, g.F - 1);
        ring[i].q = load_row16<NC>(qb + fq * fs, roff);
        ring[i].go = load_row16<NC>(gb + fq * gs, groff);
        ring[i].k = load_row16<NC>(qb + fk * fs + g.d, roff);
        ring[i].v = load_row16<NC>(qb + fk * fs + 2 * g.d, roff);
    };
#pragma unroll
    for (int i = 0; i < PF; ++i) fill(i, fa + i);

    auto store_key = [&](int f, const f32x4v (&k)[NC], const f32x4v (&v)[NC]) {
        bf16_t* row = db + f * fs + roff;                        // lane (key = lr, g), reg r -> [key][16 ct + 4g + r]
#pragma unroll
        for (int ct = 0; ct < NC; ++ct) {
            *reinterpret_cast<pk4*>(row + g.d + 16 * ct) = to_bf(k[ct] * band_scale<HD>());
            *reinterpret_cast<pk4*>(row + 2 * g.d + 16 * ct) = to_bf(v[ct]);
        }
    };

    // every unit runs the same number of rounds (seg + 2 query frames at most): the workgroup barrier below is uniform
    for (int it = 0; it < g.seg + 2; it += PF) {
        __syncthreads();                                         // neighbouring heads stay within PF frames (see forward)
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            const int f = fa + it + i;
            const Row16<NC> q = ring[i].q, go = ring[i].go;
            kw[2].k = ring[i].k;
            kw[2].v = ring[i].v;
            fill(i, f + PF);
            // column forms of this frame's Q, dO and of the new key frame (all lanes, outside the branch)
#pragma unroll
            for (int ch = 0; ch < NC; ++ch) {
                xp.write(tq + ch * TTILE, q.c[ch]);
                xp.write(tg + ch * TTILE, go.c[ch]);
                xp.write(tk + ch * TTILE, kw[2].k.c[ch]);
            }
            wave_fence();
            Col16<NC> qc, gc;
#pragma unroll
            for (int ch = 0; ch < NC; ++ch) {
                qc.v[ch] = xp.read(tq + ch * TTILE);
                gc.v[ch] = xp.read(tg + ch * TTILE);
                kw[2].kc.v[ch] = xp.read(tk + ch * TTILE);
            }
            wave_fence();
            if (f <= fz && live) {
                const bool hp = f > 0, hn = f + 1 < g.F;
                // ---- lane = query joint lr, registers = key joints 4g + r
                f32x4v s[3], p[3], ds[3];
#pragma unroll
                for (int t = 0; t < 3; ++t) s[t] = dot_rows16<NC>(kw[t].k, q);
                const float inv = band_exp<HD>(s, bias, hp, hn, p);
                float delta = 0.f;
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    p[t] *= inv;
                    ds[t] = dot_rows16<NC>(kw[t].v, go);                       // dP[q = lr][key = 4g + r]
#pragma unroll
                    for (int r = 0; r < 4; ++r) delta = __builtin_fmaf(p[t][r], ds[t][r], delta);
                }
                delta = xg_sum(delta);
                pk4 pb[3], dsb[3];
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    ds[t] = p[t] * (ds[t] - delta);                            // dS (the score scale goes on dq / dk)
                    pb[t] = to_bf(p[t]);
                    dsb[t] = to_bf(ds[t]);
                    xp.write(tp + t * TTILE, pb[t]);
                    xp.write(td + t * TTILE, dsb[t]);
                }
                // dQ[q = lr][16 ct + 4g + r] = scale * sum_key dS[q][key] K[key][c]
                if (f >= un.f0 && f < un.f1) {
                    f32x4v acc[NC];
#pragma unroll
                    for (int ct = 0; ct < NC; ++ct) acc[ct] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int t = 0; t < 3; ++t) mul_cols16<NC>(kw[t].kc, dsb[t], acc);
                    bf16_t* row = db + f * fs + roff;
#pragma unroll
                    for (int ct = 0; ct < NC; ++ct)
                        *reinterpret_cast<pk4*>(row + 16 * ct) = to_bf(acc[ct] * band_scale<HD>());
                }
                // ---- lane = key joint lr, registers = query joints 4g + r
                wave_fence();
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    const pk4 p2 = xp.read(tp + t * TTILE), ds2 = xp.read(td + t * TTILE);
                    mul_cols16<NC>(qc, ds2, dk[t]);                            // dK[key][c] += sum_q dS[q][key] Q[q][c]
                    mul_cols16<NC>(gc, p2, dv[t]);                             // dV[key][c] += sum_q P[q][key] dO[q][c]
                }
                wave_fence();
                // key frame f-1 has now seen query frames f-2, f-1, f: done
                if (hp && f - 1 >= un.f0) store_key(f - 1, dk[0], dv[0]);      // (f - 1 < f1 always: f <= f1)
#pragma unroll
                for (int ct = 0; ct < NC; ++ct) {
                    dk[0][ct] = dk[1][ct]; dk[1][ct] = dk[2][ct]; dk[2][ct] = f32x4v{0.f, 0.f, 0.f, 0.f};
                    dv[0][ct] = dv[1][ct]; dv[1][ct] = dv[2][ct]; dv[2][ct] = f32x4v{0.f, 0.f, 0.f, 0.f};
                }
                kw[0] = kw[1]; kw[1] = kw[2];
            }
        }
    }
    // the clip's last key frame has no query frame after it: after the rotation it sits in slot 0
    if (live && un.f1 == g.F) store_key(g.F - 1, dk[0], dv[0]);
}

// memory pattern probe (lab): the forward kernel's bytes moved in whole 128-byte lines -- a workgroup (4 heads of hd 16)
// reads the q, k, v tiles of its frames as 16-byte lanes, 8 lanes per token row, and writes o the same way; no arithmetic
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
template <int PF>
__global__ __launch_bounds__(256, 4) void band_memtest_k(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ o, BandGeom g,
                                                         int n_units) {
    const int lane = threadIdx.x & 63, wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const Unit16 un = decode16(g, blockIdx.x * 4);               // head = first head of the group of 4
    const int64_t rs = 3 * (int64_t)g.d;
    const bf16_t* qb = qkv + un.tok0 * rs + un.head * 16;
    bf16_t* ob = o + un.tok0 * (int64_t)g.d + un.head * 16;
    const int64_t fs = (int64_t)g.K * rs;
    const uint32_t roff = (lane >> 3) * (uint32_t)rs + 8 * (lane & 7), ooff = (lane >> 3) * (uint32_t)g.d + 8 * (lane & 7);
    // per group of 4 frames: 4 x 3 tiles x 2 halves (8 rows) = 24 loads per workgroup, 6 per wave: wave w takes frame w
    u32x4v ring[6];
    auto fill = [&](int f) {
        const int ff = min(f + wib, g.F - 1);
#pragma unroll
        for (int j = 0; j < 6; ++j)
            ring[j] = *reinterpret_cast<const u32x4v*>(qb + ff * fs + (j >> 1) * g.d + (j & 1) * 8 * rs + roff);
    };
    fill(un.f0);
    for (int fb = un.f0; fb < un.f0 + g.seg; fb += 4) {
        __syncthreads();
        u32x4v a = ring[0] ^ ring[2] ^ ring[4], b = ring[1] ^ ring[3] ^ ring[5];
        fill(fb + 4);
        const int f = fb + wib;
        if (f < un.f1) {
            bf16_t* of = ob + (int64_t)f * g.K * g.d;
            *reinterpret_cast<u32x4v*>(of + ooff) = a;
            *reinterpret_cast<u32x4v*>(of + 8 * g.d + ooff) = b;
        }
    }
}
#endif

// =============================================================== backward
// Workgroup = key frames [f0, f1) of one (clip, window, 4 heads): it walks the query frames f0-1 .. f1 (the two outer ones are
// the halo: their softmax is recomputed in full, only their contribution to the owned key frames is kept), stores dq of
// the query frames [f0, f1) and dk, dv of the key frames [f0, f1).
// Staged per query frame: Q, dO of that frame and K, V of the frame after it (4 tiles of 4 heads); the column forms come
// from the same LDS images by transposed reads, so only P and dS pass through the wave-private transposing tiles.
template <int HD, int PF, int MINW, bool ADROP = false>
__global__ __launch_bounds__(256, MINW) void band_bwd_st_k(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dO,
                                                           bf16_t* __restrict__ dqkv,
                                                           const uint64_t* __restrict__ maskrows, BandGeom g,
                                                           int64_t qkv_bytes, int64_t do_bytes, AttnDrop ad) {
    if constexpr (ADROP) ad.seed += seed_base_of(ad.base);
    constexpr int NC = HD / 16;
    using St = Staged<HD>;
    constexpr int WPF = 4 / PF;                                  // waves sharing the DMA work of one frame
    constexpr int IPW = St::NI / WPF;                            // DMA instructions per tile and wave
    static_assert(PF == 4 || PF == 2, "frame group = 4 or 2 frames");
    static_assert(St::NI % WPF == 0, "tile instructions split evenly over the waves of a frame");
    constexpr int GROUP = PF * 4 * St::TILE;                     // one frame group: PF x (Q, K, V, dO)
    constexpr int NT = 6;                                        // per wave: P x 3, dS x 3 (the prologue's K^T chunks borrow the P tiles)
    static_assert(NC <= 3, "prologue transposes fit the P tiles");
    __shared__ __attribute__((aligned(1024))) char sm[2 * GROUP + 4 * NT * TTILE];
    const int lane = threadIdx.x & 63, lr = lane & 15, gq = lane >> 4;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char* tp = sm + 2 * GROUP + wib * (NT * TTILE);
    char* td = tp + 3 * TTILE;
    char* tk = tp;
    const TileXpose xp(lane);
    const Group16 un = decode_group(g, blockIdx.x);
    const int head = 4 * un.hg + wib;
    const bool live = head < g.nH;
    const int hd_eff = min(head, g.nH - 1);
    const int64_t rs = 3 * (int64_t)g.d;
    const int64_t fs = (int64_t)g.K * rs, gs = (int64_t)g.K * g.d;
    const bf16_t* gqb = qkv + un.tok0 * rs + 4 * un.hg * HD;     // the group's columns: qkv ...
    const bf16_t* ggb = dO + un.tok0 * (int64_t)g.d + 4 * un.hg * HD;   // ... and dO
    const bf16_t* qb = qkv + un.tok0 * rs + hd_eff * HD;         // this wave's head (prologue loads)
    bf16_t* db = dqkv + un.tok0 * rs + hd_eff * HD;
    const uint32_t roff = lr * (uint32_t)rs + 4 * gq;            // lane offsets in qkv / dqkv
    float bias[3][4];
    band_bias(maskrows[un.w * 16 + lr], gq, bias);
    const int fa = max(un.f0 - 1, 0), fz = min(un.f1, g.F - 1);  // query frames fa .. fz (inclusive)

    const int span_q = (int)min(qkv_bytes - ((const char*)gqb - (const char*)qkv), (int64_t)0x7fffffff);
    const int span_g = (int)min(do_bytes - ((const char*)ggb - (const char*)dO), (int64_t)0x7fffffff);
    uint32_t voff_q[IPW], voff_g[IPW];
#pragma unroll
    for (int j = 0; j < IPW; ++j) {
        voff_q[j] = St::src(lane, (wib % WPF) * IPW + j, (uint32_t)rs * 2);
        voff_g[j] = St::src(lane, (wib % WPF) * IPW + j, (uint32_t)g.d * 2);
    }
    const uint32_t fs2 = (uint32_t)fs * 2, gs2 = (uint32_t)gs * 2, d2 = (uint32_t)g.d * 2;
    // wave w stages frame (w / WPF) of the group: Q, dO of that frame, K and V of the frame after it
    auto stage = [&](int buf, int fb) {
        const int i = wib / WPF;
        const int fq = min(fb + i, g.F - 1), fk = min(fb + i + 1, g.F - 1);
        char* dst = sm + buf * GROUP + i * 4 * St::TILE + (wib % WPF) * IPW * 1024;
        const auto rq = __builtin_amdgcn_make_buffer_rsrc((void*)gqb, 0, span_q, 0x00020000);
        const auto rg = __builtin_amdgcn_make_buffer_rsrc((void*)ggb, 0, span_g, 0x00020000);
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int j = 0; j < IPW; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rq, (lds_void)(dst + t * St::TILE + j * 1024), 16, (int)voff_q[j],
                                                         (t ? fk : fq) * fs2 + t * d2, 0, 2);
#pragma unroll
        for (int j = 0; j < IPW; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rg, (lds_void)(dst + 3 * St::TILE + j * 1024), 16, (int)voff_g[j],
                                                     fq * gs2, 0, 2);
    };
    uint32_t r_row[NC], r_col[NC];
#pragma unroll
    for (int ch = 0; ch < NC; ++ch) {
        r_row[ch] = St::piece(lr, wib, ch, gq);
        r_col[ch] = St::piece(4 * gq + (lr >> 2), wib, ch, lr & 3);
    }
    auto read_row = [&](const char* tile) {
        Row16<NC> t;
#pragma unroll
        for (int ch = 0; ch < NC; ++ch) t.c[ch] = *(const lds_u32x2*)(tile + r_row[ch]);
        return t;
    };
    auto read_col = [&](const char* tile) {
        Col16<NC> t;
#pragma unroll
        for (int ch = 0; ch < NC; ++ch)
            t.v[ch] = __builtin_bit_cast(pk4, __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(tile + r_col[ch])));
        return t;
    };

    stage(0, fa);
    struct KeyFrame { Row16<NC> k, v; Col16<NC> kc; };
    KeyFrame kw[3];
    {
        const int fp = max(fa - 1, 0);
        kw[0].k = zero_row16<NC>();
        kw[0].v = zero_row16<NC>();
        if (fa > 0) {
            kw[0].k = load_row16<NC>(qb + fp * fs + g.d, roff);
            kw[0].v = load_row16<NC>(qb + fp * fs + 2 * g.d, roff);
        }
        kw[0].kc = to_col<NC>(xp, tk, kw[0].k);
        kw[1].k = load_row16<NC>(qb + fa * fs + g.d, roff);
        kw[1].v = load_row16<NC>(qb + fa * fs + 2 * g.d, roff);
        kw[1].kc = to_col<NC>(xp, tk, kw[1].k);
    }
    f32x4v dk[3][NC], dv[3][NC];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int ct = 0; ct < NC; ++ct) { dk[t][ct] = f32x4v{0.f, 0.f, 0.f, 0.f}; dv[t][ct] = f32x4v{0.f, 0.f, 0.f, 0.f}; }

    auto store_key = [&](int f, const f32x4v (&k)[NC], const f32x4v (&v)[NC]) {
        bf16_t* row = db + f * fs + roff;                        // lane (key = lr, g), reg r -> [key][16 ct + 4g + r]
#pragma unroll
        for (int ct = 0; ct < NC; ++ct) {
            *reinterpret_cast<pk4*>(row + g.d + 16 * ct) = to_bf(k[ct] * band_scale<HD>());
            *reinterpret_cast<pk4*>(row + 2 * g.d + 16 * ct) = to_bf(v[ct]);
        }
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
            const char* tq = grp + i * 4 * St::TILE;
            const Row16<NC> q = read_row(tq), go = read_row(tq + 3 * St::TILE);
            const Col16<NC> qc = read_col(tq), gc = read_col(tq + 3 * St::TILE);
            kw[2].k = read_row(tq + St::TILE);
            kw[2].v = read_row(tq + 2 * St::TILE);
            kw[2].kc = read_col(tq + St::TILE);
            if (f <= fz && live) {
                const bool hp = f > 0, hn = f + 1 < g.F;
                // ---- lane = query joint lr, registers = key joints 4g + r
                f32x4v s[3], p[3], ds[3];
#pragma unroll
                for (int t = 0; t < 3; ++t) s[t] = dot_rows16<NC>(kw[t].k, q);
                const float inv = band_exp<HD>(s, bias, hp, hn, p);
                // attention dropout: A = D o P went into O = A V, so dP = D o dA (dA = dO V^T) and dV = A^T dO; mask recomputed
                f32x4v keep[ADROP ? 3 : 1];
                if constexpr (ADROP) band_keep(keep, ad, un.bw, g.nH, head, g.F, f, lr, gq);
                float delta = 0.f;
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    p[t] *= inv;
                    ds[t] = dot_rows16<NC>(kw[t].v, go);                       // dP[q = lr][key = 4g + r]
                    if constexpr (ADROP) ds[t] *= keep[t];
#pragma unroll
                    for (int r = 0; r < 4; ++r) delta = __builtin_fmaf(p[t][r], ds[t][r], delta);
                }
                delta = xg_sum(delta);
                pk4 pb[3], dsb[3];
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    ds[t] = p[t] * (ds[t] - delta);                            // dS (the score scale goes on dq / dk)
                    if constexpr (ADROP) pb[t] = to_bf(p[t] * keep[t]);        // the transposed P feeds dV only: A = D o P
                    else pb[t] = to_bf(p[t]);
                    dsb[t] = to_bf(ds[t]);
                    xp.write(tp + t * TTILE, pb[t]);
                    xp.write(td + t * TTILE, dsb[t]);
                }
                // dQ[q = lr][16 ct + 4g + r] = scale * sum_key dS[q][key] K[key][c]
                if (f >= un.f0 && f < un.f1) {
                    f32x4v acc[NC];
#pragma unroll
                    for (int ct = 0; ct < NC; ++ct) acc[ct] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int t = 0; t < 3; ++t) mul_cols16<NC>(kw[t].kc, dsb[t], acc);
                    bf16_t* row = db + f * fs + roff;
#pragma unroll
                    for (int ct = 0; ct < NC; ++ct)
                        *reinterpret_cast<pk4*>(row + 16 * ct) = to_bf(acc[ct] * band_scale<HD>());
                }
                // ---- lane = key joint lr, registers = query joints 4g + r
                wave_fence();
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    const pk4 p2 = xp.read(tp + t * TTILE), ds2 = xp.read(td + t * TTILE);
                    mul_cols16<NC>(qc, ds2, dk[t]);                            // dK[key][c] += sum_q dS[q][key] Q[q][c]
                    mul_cols16<NC>(gc, p2, dv[t]);                             // dV[key][c] += sum_q P[q][key] dO[q][c]
                }
                wave_fence();
                // key frame f-1 has now seen query frames f-2, f-1, f: done
                if (hp && f - 1 >= un.f0) store_key(f - 1, dk[0], dv[0]);      // (f - 1 < f1 always: f <= f1)
#pragma unroll
                for (int ct = 0; ct < NC; ++ct) {
                    dk[0][ct] = dk[1][ct]; dk[1][ct] = dk[2][ct]; dk[2][ct] = f32x4v{0.f, 0.f, 0.f, 0.f};
                    dv[0][ct] = dv[1][ct]; dv[1][ct] = dv[2][ct]; dv[2][ct] = f32x4v{0.f, 0.f, 0.f, 0.f};
                }
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

template <int HD, int PF, int MINW, int DBG = 0>
int launch_fwd_st(const bf16_t* x, bf16_t* o, const uint64_t* maskrows, const BandGeom& g, int blocks, int64_t bytes,
                  const AttnDrop& ad, hipStream_t st) {
    if (ad.p > 0.f) band_fwd_st_k<HD, PF, MINW, DBG, true><<<blocks, 256, 0, st>>>(x, o, maskrows, g, bytes, ad);
    else band_fwd_st_k<HD, PF, MINW, DBG><<<blocks, 256, 0, st>>>(x, o, maskrows, g, bytes, ad);
    HWGAT_LAUNCH_CHECK();
}

}  // namespace

int hwgat_launch_band_fwd_b16(const void* qkv, void* o, const uint64_t* maskrows, int B, int F, int nW, int nH, int hd,
                              uint32_t drop_seed, float drop_p, const uint32_t* seed_base, hipStream_t st) {
    const AttnDrop ad = make_drop(drop_seed, drop_p, seed_base);
    const int64_t base_units = (int64_t)B * nW * nH;
    int n_seg = segments(base_units, F, 256 * 4 * 2, 16, "HWGAT_BAND_FSEG");
    const int seg = (F + n_seg - 1) / n_seg;
    n_seg = (F + seg - 1) / seg;
    BandGeom g{F, nW * 16, nW, nH, nH * hd, seg, n_seg};
    const int64_t units = base_units * n_seg;
    const int64_t clip_bytes = (int64_t)F * nW * 16 * 3 * nH * hd * 2;
    if (units > 0x7fffffff || clip_bytes > 0x7fffffff) return HWGAT_ESHAPE;
    const int64_t bytes = clip_bytes * B;
    const bf16_t* x = (const bf16_t*)qkv;
    const int blocks_st = (int)((int64_t)B * nW * n_seg * ((nH + 3) / 4));
#ifdef HWGAT_LAB
    if (const char* e = lab_env("HWGAT_BAND_DBG")) {
        const int blocks = (int)((units + 3) / 4);
        const int dbg = atoi(e);
        if (dbg == 1) band_fwd_b16_k<16, 4, 4, 1><<<blocks, 256, 0, st>>>(x, (bf16_t*)o, maskrows, g, (int)units);
        else if (dbg == 2) band_fwd_b16_k<16, 4, 4, 2><<<blocks, 256, 0, st>>>(x, (bf16_t*)o, maskrows, g, (int)units);
        else if (dbg == 3) band_fwd_b16_k<16, 4, 4, 3><<<blocks, 256, 0, st>>>(x, (bf16_t*)o, maskrows, g, (int)units);
        else if (dbg == 4) band_memtest_k<4><<<blocks, 256, 0, st>>>(x, (bf16_t*)o, g, (int)units);
        else if (dbg == 5) return launch_fwd_st<16, 4, 3, 1>(x, (bf16_t*)o, maskrows, g, blocks_st, bytes, ad, st);
        else if (dbg == 6) return launch_fwd_st<16, 2, 3>(x, (bf16_t*)o, maskrows, g, blocks_st, bytes, ad, st);
        else if (dbg == 7) band_fwd_b16_k<16, 4, 4><<<blocks, 256, 0, st>>>(x, (bf16_t*)o, maskrows, g, (int)units);
        else return launch_fwd_st<16, 4, 3>(x, (bf16_t*)o, maskrows, g, blocks_st, bytes, ad, st);
        HWGAT_LAUNCH_CHECK();
    }
#endif
    if (hd == 32) return launch_fwd_st<32, 2, 2>(x, (bf16_t*)o, maskrows, g, blocks_st, bytes, ad, st);
    return launch_fwd_st<16, 4, 3>(x, (bf16_t*)o, maskrows, g, blocks_st, bytes, ad, st);
}

int hwgat_launch_band_bwd_b16(const void* qkv, const void* dO, void* dqkv, const uint64_t* maskrows, int B, int F, int nW,
                              int nH, int hd, uint32_t drop_seed, float drop_p, const uint32_t* seed_base, hipStream_t st) {
    const AttnDrop ad = make_drop(drop_seed, drop_p, seed_base);
    const int64_t base_units = (int64_t)B * nW * nH;
    int n_seg = segments(base_units, F, 256 * 4 * 2, 16, "HWGAT_BAND_BSEG");
    const int seg = (F + n_seg - 1) / n_seg;
    n_seg = (F + seg - 1) / seg;
    BandGeom g{F, nW * 16, nW, nH, nH * hd, seg, n_seg};
    const int64_t units = base_units * n_seg;
    const int64_t clip_bytes = (int64_t)F * nW * 16 * 3 * nH * hd * 2;
    if (units > 0x7fffffff || clip_bytes > 0x7fffffff) return HWGAT_ESHAPE;
    const bf16_t* x = (const bf16_t*)qkv;
    const int blocks_st = (int)((int64_t)B * nW * n_seg * ((nH + 3) / 4));
#define BWD_ST(HD, PF, MINW)                                                                                                  \
    do {                                                                                                                      \
        if (ad.p > 0.f) band_bwd_st_k<HD, PF, MINW, true><<<blocks_st, 256, 0, st>>>(x, (const bf16_t*)dO, (bf16_t*)dqkv, maskrows, g, clip_bytes * B, clip_bytes * B / 3, ad); \
        else band_bwd_st_k<HD, PF, MINW><<<blocks_st, 256, 0, st>>>(x, (const bf16_t*)dO, (bf16_t*)dqkv, maskrows, g, clip_bytes * B, clip_bytes * B / 3, ad); \
    } while (0)
#ifdef HWGAT_LAB
    if (const char* e = lab_env("HWGAT_BAND_DBG")) {
        const int blocks = (int)((units + 3) / 4);
        const int dbg = atoi(e);
        if (dbg == 7) band_bwd_b16_k<16, 1, 3><<<blocks, 256, 0, st>>>(x, (const bf16_t*)dO, (bf16_t*)dqkv, maskrows, g, (int)units);
        else if (dbg == 6) BWD_ST(16, 2, 3);
        else BWD_ST(16, 4, 2);
        HWGAT_LAUNCH_CHECK();
    }
#endif
    if (hd == 32) BWD_ST(32, 2, 2);
    else BWD_ST(16, 4, 2);
#undef BWD_ST
    HWGAT_LAUNCH_CHECK();
}
