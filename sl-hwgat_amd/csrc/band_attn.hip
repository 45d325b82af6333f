// Fused band graph-attention for the WGATE sibling model on gfx950 (MI355X).
//
// Replaces MSA.forward's attention core of the reference's hwgat/models/WGATE.py:87-108 together with
// window_partition / window_reverse (WGATE.py:32-65).  A WGATE window is one 16-joint body-part window
// over ALL T frames (T*16 tokens) with an ADDITIVE 0 / -10000 mask built from a block-tridiagonal
// adjacency (model_params.py:209-228): a query in frame f can only see keys of frames f-1, f, f+1 of its
// own part window; everything else gets exp(s - 10000 - max) == 0 exactly in fp32.  The reference
// materialises the dense (T*16)^2 score matrix per (clip, window, head); here it never exists:
//
//   unit = (clip, part window, head [, frame segment]): ONE wavefront walks the frames in order.
//   tile = 16 query joints x 16 key joints of one (query frame, key frame) pair = one
//          v_mfma_f32_16x16x4_f32 accumulator (4 registers per lane); three key tiles per query frame.
//   S^T  = K Q^T : both operands are "lane = row, 4 consecutive head-dim elements" = one 16-byte
//          global load per lane straight from the natural-order qkv tensor -- no LDS anywhere.
//   softmax over the <= 48 candidate keys in registers (12 per lane + 2 cross-lane steps);
//   O    = P V   : P is fed back as the MFMA A operand, V comes in as 4 dwords per lane.
//   K / V tiles live in a 3-frame sliding register window, so every q, k, v element is read from
//   HBM once and every o element written once: traffic = the algorithmic 4*E*s (fwd), 7*E*s (bwd).
//
// Backward: dK / dV need the P and dS tiles transposed (lane = key).  Shipped form (XPOSE): one 16-byte write
// and four 4-byte reads per tile and lane through a 7.5 KB wave-private LDS scratch, wave-level fences only.
// Register-only form (HWGAT_BAND_XPOSE=0): run the S and dP products a second time with the operands swapped
// (same registers) and move the per-query softmax statistics across lanes with ds_bpermute -- 84 instead of 60
// MFMAs per frame, 16 % slower.
// dK / dV of a key frame collect the contributions of query frames f-1, f, f+1 in a sliding
// 3-frame accumulator window and are stored once.
#include <stdlib.h>
#include "band_common.h"

namespace {
using namespace band;

typedef float f32x4v __attribute__((ext_vector_type(4)));

// D(16x16) += A(16x4) B(4x16): lane l supplies A[i = l&15][k = l>>4] and B[k = l>>4][j = l&15];
// register r of lane l is D[i = 4*(l>>4) + r][j = l&15].
__device__ __forceinline__ f32x4v mfma16(float a, float b, f32x4v c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

template <typename T> __device__ __forceinline__ f32x4v ld4(const T* p) {
    if constexpr (sizeof(T) == 4) {
        return *reinterpret_cast<const f32x4v*>(p);
    } else {
        const bf16x4 t = *reinterpret_cast<const bf16x4*>(p);
        f32x4v r = {(float)t.x, (float)t.y, (float)t.z, (float)t.w};
        return r;
    }
}
// N = 1 or 2 consecutive elements <-> floats, one memory instruction
template <typename T, int N> __device__ __forceinline__ void ldn(const T* p, float (&v)[N]) {
    if constexpr (N == 1) v[0] = (float)*p;
    else if constexpr (sizeof(T) == 4) { const f32x2 t = *reinterpret_cast<const f32x2*>(p); v[0] = t.x; v[1] = t.y; }
    else { const bf16x2 t = *reinterpret_cast<const bf16x2*>(p); v[0] = (float)t.x; v[1] = (float)t.y; }
}
template <typename T, int N> __device__ __forceinline__ void stn(T* p, const float (&v)[N]) {
    if constexpr (N == 1) *p = (T)v[0];
    else if constexpr (sizeof(T) == 4) *reinterpret_cast<f32x2*>(p) = f32x2{v[0], v[1]};
    else *reinterpret_cast<bf16x2*>(p) = bf16x2{(bf16_t)v[0], (bf16_t)v[1]};
}

__device__ __forceinline__ float xg_max(float v) {              // over the 4 lanes l, l^16, l^32, l^48
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float xg_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}

// row-operand tile: X[row = l&15][c = 16*ch + 4*g + s]  (A or B^T operand of a head-dim contraction)
template <int NC> struct RowT { f32x4v c[NC]; };
// column-operand tile: X[row = 4*g + r][c = NC*(l&15) + ct]  (B operand of a row contraction: column j of channel tile ct is
// channel NC j + ct, so a lane's NC tiles are NC consecutive elements = one load per row, and the product comes out as NC
// consecutive channels per lane = one store per row; with tile ct = channels 16 ct + j every head_dim-32 column load and
// store was two 4-byte instructions)
template <int NC> struct ColT { float v[NC][4]; };

// `base` is wave-uniform (SGPR pair), `off` a 32-bit per-lane element offset: the loads use the
// "scalar base + vector offset" addressing form and need no 64-bit vector address arithmetic.
template <typename T, int NC>
__device__ __forceinline__ RowT<NC> load_row(const T* base, uint32_t off, float mul) {
    RowT<NC> t;                                                  // off = (l&15) * row_stride + 4 * g
#pragma unroll
    for (int ch = 0; ch < NC; ++ch) t.c[ch] = ld4<T>(base + off + 16 * ch) * mul;
    return t;
}
template <typename T, int NC>
__device__ __forceinline__ ColT<NC> load_col(const T* base, int64_t row_stride, uint32_t off, float mul) {
    ColT<NC> t;                                                  // off = 4 * g * row_stride + NC * (l&15)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float x[NC];
        ldn<T, NC>(base + r * row_stride + off, x);
#pragma unroll
        for (int ct = 0; ct < NC; ++ct) t.v[ct][r] = x[ct] * mul;
    }
    return t;
}
// rows 4g + r of a product in the column layout (acc[ct][r] = channel NC (l&15) + ct of row 4g + r), scaled
template <typename T, int NC>
__device__ __forceinline__ void store_col(T* base, int64_t row_stride, uint32_t off, const f32x4v (&acc)[NC], float mul) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float x[NC];
#pragma unroll
        for (int ct = 0; ct < NC; ++ct) x[ct] = acc[ct][r] * mul;
        stn<T, NC>(base + r * row_stride + off, x);
    }
}
template <int NC> __device__ __forceinline__ RowT<NC> zero_row() {
    RowT<NC> t;
#pragma unroll
    for (int ch = 0; ch < NC; ++ch) t.c[ch] = f32x4v{0.f, 0.f, 0.f, 0.f};
    return t;
}
template <int NC> __device__ __forceinline__ ColT<NC> zero_col() {
    ColT<NC> t;
#pragma unroll
    for (int ct = 0; ct < NC; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) t.v[ct][r] = 0.f;
    return t;
}

// D[i][j] = sum_c X[i][c] Y[j][c] for two row-operand tiles: lane (j = l&15, g), reg r -> D[4g + r][j]
template <int NC>
__device__ __forceinline__ f32x4v dot_rows(const RowT<NC>& x, const RowT<NC>& y) {
    f32x4v acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ch = 0; ch < NC; ++ch) {
        acc = mfma16(x.c[ch].x, y.c[ch].x, acc);
        acc = mfma16(x.c[ch].y, y.c[ch].y, acc);
        acc = mfma16(x.c[ch].z, y.c[ch].z, acc);
        acc = mfma16(x.c[ch].w, y.c[ch].w, acc);
    }
    return acc;
}
// acc[ct] (16 x 16 cols) += A(16x16) Y where a[r] = A[i = l&15][k = 4g + r] and Y is a column-operand tile
template <int NC>
__device__ __forceinline__ void mul_cols(const f32x4v& a, const ColT<NC>& y, f32x4v (&acc)[NC]) {
#pragma unroll
    for (int ct = 0; ct < NC; ++ct) {
        acc[ct] = mfma16(a.x, y.v[ct][0], acc[ct]);
        acc[ct] = mfma16(a.y, y.v[ct][1], acc[ct]);
        acc[ct] = mfma16(a.z, y.v[ct][2], acc[ct]);
        acc[ct] = mfma16(a.w, y.v[ct][3], acc[ct]);
    }
}

// probabilities of one query row from its three key tiles; masked / out-of-clip entries are exactly 0
// (additive -10000 of WGATE.py:97-100 underflows to 0 in fp32; the diagonal is always visible)
__device__ __forceinline__ void band_softmax(const f32x4v (&s)[3], uint32_t vis, f32x4v (&p)[3], float& m_out,
                                             float& linv_out) {
    float m = -3.0e38f;
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if ((vis >> (4 * t + r)) & 1u) m = fmaxf(m, s[t][r]);
    m = xg_max(m);
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float e = ((vis >> (4 * t + r)) & 1u) ? __expf(s[t][r] - m) : 0.f;
            p[t][r] = e;
            sum += e;
        }
    sum = xg_sum(sum);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int t = 0; t < 3; ++t) p[t] *= inv;
    m_out = m;
    linv_out = inv;
}

// 12 visibility bits (3 key tiles x 4 key joints 4g+r) of query joint `lr` from its 48-bit mask row
__device__ __forceinline__ uint32_t vis_q(uint64_t mrow, int g, bool has_prev, bool has_next) {
    uint32_t v = 0;
#pragma unroll
    for (int t = 0; t < 3; ++t) v |= (uint32_t)((mrow >> (16 * t + 4 * g)) & 0xFull) << (4 * t);
    if (!has_prev) v &= ~0x00Fu;
    if (!has_next) v &= ~0xF00u;
    return v;
}

// =============================================================== forward
template <typename T, int HD, int PF, int MINW, bool ADROP = false>
__global__ __launch_bounds__(256, MINW) void band_attn_fwd_k(const T* __restrict__ qkv, T* __restrict__ o,
                                                       const uint64_t* __restrict__ maskrows, BandGeom g,
                                                       int n_units, AttnDrop ad) {
    if constexpr (ADROP) ad.seed += seed_base_of(ad.base);
    constexpr int NC = HD / 16;
    const int lane = threadIdx.x & 63, lr = lane & 15, gq = lane >> 4;
    // the wave index is uniform; say so, so that unit decoding and all base pointers live in SGPRs
    const int u_raw = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool live = u_raw < n_units;                           // tail waves shadow the last unit (no stores): every
    const int u = live ? u_raw : n_units - 1;                    // wave of a workgroup reaches the per-group barrier
    const BandUnit un = decode_band(g, u);
    const int64_t rs = 3 * (int64_t)g.d;                         // qkv row stride (elements)
    const T* qb = qkv + un.tok0 * rs + un.head * HD;
    T* ob = o + un.tok0 * (int64_t)g.d + un.head * HD;
    const int64_t fs = (int64_t)g.K * rs;                        // frame stride in qkv
    const uint64_t mrow = maskrows[un.w * 16 + lr];
    const uint32_t roff = lr * (uint32_t)rs + 4 * gq, coff = 4 * gq * (uint32_t)rs + NC * lr;   // per-lane offsets in qkv
    const uint32_t ooff = 4 * gq * (uint32_t)g.d + NC * lr;                                     // ... and in o

    // sliding window: K (row operand) and V (column operand) of frames f-1, f, f+1
    RowT<NC> kw[3];
    ColT<NC> vw[3];
    kw[0] = zero_row<NC>(); vw[0] = zero_col<NC>();
    if (un.f0 > 0) {
        kw[0] = load_row<T, NC>(qb + (un.f0 - 1) * fs + g.d, roff, 1.0f);
        vw[0] = load_col<T, NC>(qb + (un.f0 - 1) * fs + 2 * g.d, rs, coff, 1.0f);
    }
    kw[1] = load_row<T, NC>(qb + un.f0 * fs + g.d, roff, 1.0f);
    vw[1] = load_col<T, NC>(qb + un.f0 * fs + 2 * g.d, rs, coff, 1.0f);

    // prefetch ring: slot i holds Q of frame f+i and K, V of frame f+i+1
    RowT<NC> rq[PF], rk[PF];
    ColT<NC> rv[PF];
    // loads are unconditional (a branch around a load makes the compiler drain the whole ring): frames
    // past the clip re-read the last frame; such tiles are either never consumed or masked out by vis_q
    auto fill = [&](int i, int f) {                              // f = query frame of the slot
        const int fq = min(f, g.F - 1), fk = min(f + 1, g.F - 1);
        rq[i] = load_row<T, NC>(qb + fq * fs, roff, band_scale<HD>());
        rk[i] = load_row<T, NC>(qb + fk * fs + g.d, roff, 1.0f);
        rv[i] = load_col<T, NC>(qb + fk * fs + 2 * g.d, rs, coff, 1.0f);
    };
#pragma unroll
    for (int i = 0; i < PF; ++i) fill(i, un.f0 + i);

    for (int fb = un.f0; fb < un.f0 + g.seg; fb += PF) {
        // the 4 waves of a workgroup are neighbouring heads of one (clip, window): keep them within PF frames of
        // each other so that the two heads sharing a 128-byte line touch it while it is still cached
        __syncthreads();
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            const int f = fb + i;
            const RowT<NC> q = rq[i];
            kw[2] = rk[i];
            vw[2] = rv[i];
            fill(i, f + PF);
            if (f < un.f1 && live) {
                f32x4v s[3], p[3];
#pragma unroll
                for (int t = 0; t < 3; ++t) s[t] = dot_rows<NC>(kw[t], q);     // s[t][r] = S[q = lr][key = 4g + r]
                float m, linv;
                band_softmax(s, vis_q(mrow, gq, f > 0, f + 1 < g.F), p, m, linv);
                if constexpr (ADROP) {                           // WGATE.py:103
                    f32x4v keep[3];
                    band_keep(keep, ad, un.bw, g.nH, un.head, g.F, f, lr, gq);
#pragma unroll
                    for (int t = 0; t < 3; ++t) p[t] *= keep[t];
                }
                f32x4v oacc[NC];
#pragma unroll
                for (int ct = 0; ct < NC; ++ct) oacc[ct] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int t = 0; t < 3; ++t) mul_cols<NC>(p[t], vw[t], oacc);
                // lane (c = lr, g), reg r -> O[q = 4g + r][NC c + ct]
                store_col<T, NC>(ob + (int64_t)f * g.K * g.d, g.d, ooff, oacc, 1.0f);
            }
            kw[0] = kw[1]; kw[1] = kw[2];
            vw[0] = vw[1]; vw[1] = vw[2];
        }
    }
}

// =============================================================== backward
// XPOSE = true: the transposed P / dS tiles (lane = key) come from a wave-private LDS scratch (one 16-byte write
// and four 4-byte reads per tile and lane) instead of a second pair of MFMA products: 60 instead of 84 MFMAs per
// frame, no cross-lane statistics.  XPOSE = false is the register-only form described above.
template <typename T, int HD, int PF, int MINW, bool XPOSE, bool ADROP = false>
__global__ __launch_bounds__(256, MINW) void band_attn_bwd_k(const T* __restrict__ qkv, const T* __restrict__ dO,
                                                       T* __restrict__ dqkv,
                                                       const uint64_t* __restrict__ maskrows, BandGeom g,
                                                       int n_units, AttnDrop ad) {
    static_assert(XPOSE || !ADROP, "attention dropout exists for the shipped (LDS-transposed) form only");
    if constexpr (ADROP) ad.seed += seed_base_of(ad.base);
    constexpr int NC = HD / 16;
    constexpr int XLD = 20;                                      // scratch row stride (floats): 16-byte aligned rows
    __shared__ __attribute__((aligned(16))) float xsm[XPOSE ? 4 * 6 * 16 * XLD : 4];
    const int lane = threadIdx.x & 63, lr = lane & 15, gq = lane >> 4;
    // the wave index is uniform; say so, so that unit decoding and all base pointers live in SGPRs
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* xs = xsm + (XPOSE ? wib * (6 * 16 * XLD) : 0);        // [P tiles 0..2 | dS tiles 3..5][q][XLD]
    const int u_raw = blockIdx.x * 4 + wib;
    const bool live = u_raw < n_units;                           // tail waves shadow the last unit without storing
    const int u = live ? u_raw : n_units - 1;
    const BandUnit un = decode_band(g, u);                       // backward units always span the whole clip
    const int64_t rs = 3 * (int64_t)g.d;
    const int64_t fs = (int64_t)g.K * rs, gs = (int64_t)g.K * g.d;
    const T* qb = qkv + un.tok0 * rs + un.head * HD;
    const T* gb = dO + un.tok0 * (int64_t)g.d + un.head * HD;
    T* db = dqkv + un.tok0 * rs + un.head * HD;
    const uint64_t mrow = maskrows[un.w * 16 + lr];              // row of query joint lr
    const uint32_t roff = lr * (uint32_t)rs + 4 * gq, coff = 4 * gq * (uint32_t)rs + NC * lr;       // lane offsets in qkv / dqkv
    const uint32_t groff = lr * (uint32_t)g.d + 4 * gq, gcoff = 4 * gq * (uint32_t)g.d + NC * lr;  // ... in dO
    uint64_t mrow2[4];                                           // rows of query joints 4g + r (transposed tiles)
#pragma unroll
    for (int r = 0; r < 4; ++r) mrow2[r] = maskrows[un.w * 16 + 4 * gq + r];

    struct KeyFrame { RowT<NC> k, v; ColT<NC> kc; };
    auto load_key = [&](int f) {
        KeyFrame x;
        x.k = load_row<T, NC>(qb + f * fs + g.d, roff, 1.0f);
        x.v = load_row<T, NC>(qb + f * fs + 2 * g.d, roff, 1.0f);
        x.kc = load_col<T, NC>(qb + f * fs + g.d, rs, coff, 1.0f);
        return x;
    };
    auto zero_key = [&]() {
        KeyFrame x;
        x.k = zero_row<NC>(); x.v = zero_row<NC>(); x.kc = zero_col<NC>();
        return x;
    };
    KeyFrame kw[3];
    kw[0] = zero_key();
    kw[1] = load_key(0);
    kw[2] = zero_key();
    f32x4v dk[3][NC], dv[3][NC];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int ct = 0; ct < NC; ++ct) { dk[t][ct] = f32x4v{0.f, 0.f, 0.f, 0.f}; dv[t][ct] = f32x4v{0.f, 0.f, 0.f, 0.f}; }

    // prefetch ring: slot i = query frame f+i: Q, dO (row + column operands) and the key frame f+i+1
    struct QFrame { RowT<NC> q, go; ColT<NC> qc, gc; };
    QFrame rq[PF];
    KeyFrame rk[PF];
    auto fill = [&](int i, int f) {                              // unconditional, clamped (see the forward kernel)
        const int fq = min(f, g.F - 1), fk = min(f + 1, g.F - 1);
        rq[i].q = load_row<T, NC>(qb + fq * fs, roff, band_scale<HD>());
        rq[i].qc = load_col<T, NC>(qb + fq * fs, rs, coff, band_scale<HD>());
        rq[i].go = load_row<T, NC>(gb + fq * gs, groff, 1.0f);
        rq[i].gc = load_col<T, NC>(gb + fq * gs, g.d, gcoff, 1.0f);
        rk[i] = load_key(fk);
    };
#pragma unroll
    for (int i = 0; i < PF; ++i) fill(i, i);

    auto store_key = [&](int f, const f32x4v (&k)[NC], const f32x4v (&v)[NC]) {
        store_col<T, NC>(db + f * fs + g.d, rs, coff, k, 1.0f);
        store_col<T, NC>(db + f * fs + 2 * g.d, rs, coff, v, 1.0f);
    };

    for (int fb = 0; fb < g.F; fb += PF) {
        __syncthreads();                                         // neighbouring heads stay within PF frames (see forward)
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            const int f = fb + i;
            const QFrame q = rq[i];
            kw[2] = rk[i];
            fill(i, f + PF);
            if (f < g.F && live) {
                const bool hp = f > 0, hn = f + 1 < g.F;
                // ---- orientation 1: lane = query joint lr, registers = key joints 4g + r
                f32x4v s[3], p[3], ds[3];
#pragma unroll
                for (int t = 0; t < 3; ++t) s[t] = dot_rows<NC>(kw[t].k, q.q);
                float m, linv;
                band_softmax(s, vis_q(mrow, gq, hp, hn), p, m, linv);
                // attention dropout: A = D o P went into O = A V, so dP = D o dA (dA = dO V^T) and dV = A^T dO; mask recomputed
                f32x4v keep[ADROP ? 3 : 1];
                if constexpr (ADROP) band_keep(keep, ad, un.bw, g.nH, un.head, g.F, f, lr, gq);
                float delta = 0.f;
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    ds[t] = dot_rows<NC>(kw[t].v, q.go);                       // dP[q = lr][key = 4g + r]
                    if constexpr (ADROP) ds[t] *= keep[t];
#pragma unroll
                    for (int r = 0; r < 4; ++r) delta += p[t][r] * ds[t][r];
                }
                delta = xg_sum(delta);
#pragma unroll
                for (int t = 0; t < 3; ++t) ds[t] = p[t] * (ds[t] - delta);
                // dQ[q = 4g + r][c] = scale * sum_key dS[q][key] K[key][c]
                f32x4v acc[NC];
#pragma unroll
                for (int ct = 0; ct < NC; ++ct) acc[ct] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int t = 0; t < 3; ++t) mul_cols<NC>(ds[t], kw[t].kc, acc);
                store_col<T, NC>(db + f * fs, rs, coff, acc, band_scale<HD>());
                // ---- orientation 2: lane = key joint lr, registers = query joints 4g + r
                if constexpr (XPOSE) {
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        if constexpr (ADROP) *reinterpret_cast<f32x4v*>(xs + (t * 16 + lr) * XLD + 4 * gq) = p[t] * keep[t];   // dV takes A = D o P
                        else *reinterpret_cast<f32x4v*>(xs + (t * 16 + lr) * XLD + 4 * gq) = p[t];
                        *reinterpret_cast<f32x4v*>(xs + ((3 + t) * 16 + lr) * XLD + 4 * gq) = ds[t];
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        f32x4v p2, ds2;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            p2[r] = xs[(t * 16 + 4 * gq + r) * XLD + lr];
                            ds2[r] = xs[((3 + t) * 16 + 4 * gq + r) * XLD + lr];
                        }
                        mul_cols<NC>(ds2, q.qc, dk[t]);                        // dK[key][c] += sum_q dS[q][key] (scale*Q)[q][c]
                        mul_cols<NC>(p2, q.gc, dv[t]);                         // dV[key][c] += sum_q P[q][key] dO[q][c]
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                } else {
                    float m2[4], l2[4], d2[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {                              // statistics of query joint 4g + r
                        m2[r] = __shfl(m, 4 * gq + r, 64);
                        l2[r] = __shfl(linv, 4 * gq + r, 64);
                        d2[r] = __shfl(delta, 4 * gq + r, 64);
                    }
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        const bool tile_ok = (t == 0) ? hp : (t == 2) ? hn : true;
                        const f32x4v s2 = dot_rows<NC>(q.q, kw[t].k);          // S[q = 4g + r][key = lr]
                        const f32x4v dp2 = dot_rows<NC>(q.go, kw[t].v);
                        f32x4v p2, ds2;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const bool vis = tile_ok && ((mrow2[r] >> (16 * t + lr)) & 1ull);
                            p2[r] = vis ? __expf(s2[r] - m2[r]) * l2[r] : 0.f;
                            ds2[r] = p2[r] * (dp2[r] - d2[r]);
                        }
                        mul_cols<NC>(ds2, q.qc, dk[t]);                        // dK[key][c] += sum_q dS[q][key] (scale*Q)[q][c]
                        mul_cols<NC>(p2, q.gc, dv[t]);                         // dV[key][c] += sum_q P[q][key] dO[q][c]
                    }
                }
                // key frame f-1 has now seen query frames f-2, f-1, f: done
                if (hp) store_key(f - 1, dk[0], dv[0]);
#pragma unroll
                for (int ct = 0; ct < NC; ++ct) {
                    dk[0][ct] = dk[1][ct]; dk[1][ct] = dk[2][ct]; dk[2][ct] = f32x4v{0.f, 0.f, 0.f, 0.f};
                    dv[0][ct] = dv[1][ct]; dv[1][ct] = dv[2][ct]; dv[2][ct] = f32x4v{0.f, 0.f, 0.f, 0.f};
                }
                kw[0] = kw[1]; kw[1] = kw[2];
            }
        }
    }
    if (live) store_key(g.F - 1, dk[0], dv[0]);                  // after the rotation the last key frame sits in slot 0
}

__global__ void mfma16_probe_k(const float* a, const float* b, float* out) {
    const int lane = threadIdx.x;
    f32x4v acc = {0.f, 0.f, 0.f, 0.f};
    // A[i][k] supplied by lane (i = lane&15, k = lane>>4); B[k][j] by lane (j = lane&15, k = lane>>4)
    acc = mfma16(a[(lane & 15) * 4 + (lane >> 4)], b[(lane >> 4) * 16 + (lane & 15)], acc);
#pragma unroll
    for (int i = 0; i < 4; ++i) out[lane * 4 + i] = acc[i];
}

}  // namespace

extern "C" int hwgat_debug_mfma16x16x4(const float* a, const float* b, float* out, void* stream) {
    if (!a || !b || !out) return HWGAT_EINVAL;
    mfma16_probe_k<<<1, 64, 0, (hipStream_t)stream>>>(a, b, out);
    HWGAT_LAUNCH_CHECK();
}

extern "C" int hwgat_band_attn_fwd_drop(const void* qkv, void* o, const uint64_t* maskrows, int B, int F, int nW,
                                        int nH, int hd, int dtype, uint32_t drop_seed, float drop_p,
                                        const uint32_t* seed_base, void* stream) {
    if (!qkv || !o || !maskrows || drop_p < 0.f || drop_p >= 1.f) return HWGAT_EINVAL;
    const AttnDrop ad = make_drop(drop_seed, drop_p, seed_base);
    if (!band_ok(B, F, nW, nH, hd)) return HWGAT_ESHAPE;
    // enough wavefronts to fill the chip: split the clip into frame segments (1 halo frame of K, V each)
    const int64_t base_units = (int64_t)B * nW * nH;
    int n_seg = 1;
    // (at B64 T128: 2048 / 4096 / 8192 waves run in 296 / 298 / 334 us and fetch 1.19 / 1.33 / - x the algorithmic bytes)
    while (base_units * n_seg < 256 * 8 && F / (n_seg * 2) >= 8) n_seg *= 2;
    const int seg = (F + n_seg - 1) / n_seg;
    n_seg = (F + seg - 1) / seg;
    BandGeom g{F, nW * 16, nW, nH, nH * hd, seg, n_seg};
    const int64_t units = base_units * n_seg;
    if (units > 0x7fffffff) return HWGAT_ESHAPE;
    hipStream_t st = (hipStream_t)stream;
    // bf16 storage takes the bf16-MFMA kernels of band_attn_bf16.hip; the fp32-MFMA form on bf16 data survives as a lab A/B
    static const bool old_b16 = lab_env("HWGAT_BAND_B16") && lab_env("HWGAT_BAND_B16")[0] == '0';
    const int blocks = (int)((units + 3) / 4);
    // prefetch depth / occupancy: A/B on MI355X at B64 T128 K64 (fp32): PF 8 at 2 waves/SIMD 327 us, PF 4 at 3-4 waves/SIMD 295-302 us
#define FWD_ARGS(T) (const T*)qkv, (T*)o, maskrows, g, (int)units, ad
#define FWD(T)                                                                                       \
    if (ad.p > 0.f) {                                                                                \
        if (hd == 32) band_attn_fwd_k<T, 32, 4, 1, true><<<blocks, 256, 0, st>>>(FWD_ARGS(T));       \
        else band_attn_fwd_k<T, 16, 4, 3, true><<<blocks, 256, 0, st>>>(FWD_ARGS(T));                \
    } else if (hd == 32) band_attn_fwd_k<T, 32, 4, 1><<<blocks, 256, 0, st>>>(FWD_ARGS(T));          \
    else band_attn_fwd_k<T, 16, 4, 3><<<blocks, 256, 0, st>>>(FWD_ARGS(T));
    if (dtype == HWGAT_F32) {
        // head_dim 16 (WGATE): the workgroup-staged kernels of band_attn_f32.hip; the one-wave-per-head form stays for
        // head_dim 32 and as the lab A/B (HWGAT_BAND_F32=0)
        static const bool old_f32 = lab_env("HWGAT_BAND_F32") && lab_env("HWGAT_BAND_F32")[0] == '0';
        if (hd == 16 && !old_f32) return hwgat_launch_band_fwd_f32(qkv, o, maskrows, B, F, nW, nH, ad.seed, ad.p, ad.base, st);
        FWD(float)
    }
    else if (dtype == HWGAT_BF16) {
        if (!old_b16) return hwgat_launch_band_fwd_b16(qkv, o, maskrows, B, F, nW, nH, hd, ad.seed, ad.p, ad.base, st);
        FWD(bf16_t)
    } else return HWGAT_EDTYPE;
#undef FWD
#undef FWD_ARGS
    HWGAT_LAUNCH_CHECK();
}

extern "C" int hwgat_band_attn_fwd(const void* qkv, void* o, const uint64_t* maskrows, int B, int F, int nW,
                                   int nH, int hd, int dtype, void* stream) {
    return hwgat_band_attn_fwd_drop(qkv, o, maskrows, B, F, nW, nH, hd, dtype, 0u, 0.f, nullptr, stream);
}

extern "C" int hwgat_band_attn_bwd_drop(const void* qkv, const void* dO, void* dqkv, const uint64_t* maskrows, int B,
                                        int F, int nW, int nH, int hd, int dtype, uint32_t drop_seed, float drop_p,
                                        const uint32_t* seed_base, void* stream) {
    if (!qkv || !dO || !dqkv || !maskrows || drop_p < 0.f || drop_p >= 1.f) return HWGAT_EINVAL;
    const AttnDrop ad = make_drop(drop_seed, drop_p, seed_base);
    if (!band_ok(B, F, nW, nH, hd)) return HWGAT_ESHAPE;
    BandGeom g{F, nW * 16, nW, nH, nH * hd, F, 1};
    const int64_t units = (int64_t)B * nW * nH;
    if (units > 0x7fffffff) return HWGAT_ESHAPE;
    hipStream_t st = (hipStream_t)stream;
    static const bool old_b16 = lab_env("HWGAT_BAND_B16") && lab_env("HWGAT_BAND_B16")[0] == '0';
    const int blocks = (int)((units + 3) / 4);
    // register-only form: PF 2 at 1 wave/SIMD 644 us, PF 2 at 2-3 waves/SIMD 630 us, PF 1 at 2 waves/SIMD 619 us (all issue-bound
    // alike); with the LDS transposes (XPOSE, the default; HWGAT_BAND_XPOSE=0 selects the register-only form) 514 us
    static const bool xpose = !(lab_env("HWGAT_BAND_XPOSE") && lab_env("HWGAT_BAND_XPOSE")[0] == '0');
#define BWD_ARGS(T) (const T*)qkv, (const T*)dO, (T*)dqkv, maskrows, g, (int)units, ad
#define BWD(T)                                                                                       \
    if (ad.p > 0.f) {                                                                                \
        if (hd == 32) band_attn_bwd_k<T, 32, 1, 1, true, true><<<blocks, 256, 0, st>>>(BWD_ARGS(T)); \
        else band_attn_bwd_k<T, 16, 2, 2, true, true><<<blocks, 256, 0, st>>>(BWD_ARGS(T));          \
    } else                                                                                           \
    if (hd == 32 && xpose) band_attn_bwd_k<T, 32, 1, 1, true><<<blocks, 256, 0, st>>>(BWD_ARGS(T)); \
    else if (hd == 32) band_attn_bwd_k<T, 32, 1, 1, false><<<blocks, 256, 0, st>>>(BWD_ARGS(T));     \
    else if (xpose) band_attn_bwd_k<T, 16, 2, 2, true><<<blocks, 256, 0, st>>>(BWD_ARGS(T));         \
    else band_attn_bwd_k<T, 16, 2, 2, false><<<blocks, 256, 0, st>>>(BWD_ARGS(T));
    if (dtype == HWGAT_F32) {
        static const bool old_f32 = lab_env("HWGAT_BAND_F32") && lab_env("HWGAT_BAND_F32")[0] == '0';
        if (hd == 16 && !old_f32) return hwgat_launch_band_bwd_f32(qkv, dO, dqkv, maskrows, B, F, nW, nH, ad.seed, ad.p, ad.base, st);
        BWD(float)
    }
    else if (dtype == HWGAT_BF16) {
        if (!old_b16) return hwgat_launch_band_bwd_b16(qkv, dO, dqkv, maskrows, B, F, nW, nH, hd, ad.seed, ad.p, ad.base, st);
        BWD(bf16_t)
    } else return HWGAT_EDTYPE;
#undef BWD
#undef BWD_ARGS
    HWGAT_LAUNCH_CHECK();
}

extern "C" int hwgat_band_attn_bwd(const void* qkv, const void* dO, void* dqkv, const uint64_t* maskrows, int B,
                                   int F, int nW, int nH, int hd, int dtype, void* stream) {
    return hwgat_band_attn_bwd_drop(qkv, dO, dqkv, maskrows, B, F, nW, nH, hd, dtype, 0u, 0.f, nullptr, stream);
}
