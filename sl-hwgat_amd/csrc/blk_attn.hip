// Fused block graph-attention for the HGATE sibling model on gfx950 (MI355X).
//
// Replaces MSA.forward's attention core of the reference's hwgat/models/HGATE.py:84-108 together with
// block_partition / block_reverse / torch.roll (HGATE.py:30-47,184-207), which become index arithmetic.
//
// Work unit = one (block, head): 2 frames x KJ joints (KJ <= 32; 29 in HGATEParams) x head_dim.  Each
// frame is one 32-row MFMA tile (rows >= KJ are zero padding that is never loaded or stored), so a unit
// is 2 query tiles x 2 key tiles of the 32-token machinery of win_attn.hip:
//
//   S^T[kt][qt] = K_kt Q_qt^T     v_mfma_f32_32x32x2_f32, lane = query, 2 x 16 registers = this
//                                 lane's half of the 64 key slots (partner lane^32 has the rest)
//   masks + softmax               in registers: adjacency / shift bit rows (2 x u32 per query), the
//                                 "== 0 -> -10000" fill (HGATE.py:104) over the 2*KJ real keys; pad
//                                 key slots get probability exactly 0 and never count in max / sum
//   O_qt = sum_kt P[kt] V_kt      P fed straight back as the MFMA A operand
//
// One 64-lane wavefront (= one workgroup) owns a unit end to end with private LDS: no workgroup
// barriers.  The backward pass keeps Q, K, V, dO (2 tiles each) in LDS, runs the two query tiles as two
// passes and accumulates dK / dV across them in registers; P^T and dS^T go through one LDS scratch tile
// that is reused for both.
//
// HBM traffic is the algorithmic 4*E*s (fwd) / 7*E*s (bwd) like the window kernel; the MFMA work per
// byte is 2 x (64/58)^2 higher (64-slot tiles for 58 tokens), so at head_dim 64 the backward pass sits
// at the fp32 MFMA / HBM ridge rather than clearly HBM-bound (HGATE section).
#include "attn_common.h"

namespace {

struct BlkGeom {
    int F, KJ, nH, f, d, shift;
};

struct BUnit {
    int64_t base[2];        // token index of joint 0 of frame A / frame B
    int head, mrow;         // head index, first row of this unit's mask variant
};
__device__ __forceinline__ BUnit decode_bunit(const BlkGeom& g, int u) {
    BUnit r;
    const int n = u / g.nH;
    r.head = u - n * g.nH;
    const int fi = n % g.f;
    const int b = n / g.f;
    int fa = 2 * fi + g.shift, fb = fa + 1;          // torch.roll(x, -shift) (HGATE.py:186): shifted[t] = x[(t+shift) % F]
    if (fa >= g.F) fa -= g.F;
    if (fb >= g.F) fb -= g.F;
    r.base[0] = ((int64_t)b * g.F + fa) * g.KJ;
    r.base[1] = ((int64_t)b * g.F + fb) * g.KJ;
    r.mrow = (g.shift && fi == g.f - 1) ? 64 : 0;    // the last shifted block straddles the clip ends (HGATE.py:158-172)
    return r;
}

// masks + softmax on one lane's 2 x 16 logits of one query row.
// s[kt][r] = S[q][key slot kt*32 + crow(r,hh)] on entry, p = final probabilities on exit.
// returns bit (kt*16 + r) set where the gradient flows (real key whose logit was not replaced by -10000).
__device__ __forceinline__ uint32_t masked_softmax64(float (&s)[2][16], float (&p)[2][16], uint32_t mb0,
                                                     uint32_t mb1, int hh, int KJ) {
    uint32_t nz = 0;
    float m = -3.0e38f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
        const uint32_t mb = kt ? mb1 : mb0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = crow(r, hh);
            const bool vis = (mb >> j) & 1u;                 // HGATE.py:96-102
            float v = vis ? s[kt][r] : 0.f;
            if (v == 0.f) v = -10000.f; else nz |= 1u << (kt * 16 + r);   // HGATE.py:104
            if (j >= KJ) { v = -3.0e38f; nz &= ~(1u << (kt * 16 + r)); } // pad slot: not a key at all
            s[kt][r] = v;
            m = fmaxf(m, v);
        }
    }
    m = fmaxf(m, partner(m));
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { p[kt][r] = sm_exp(s[kt][r] - m); sum += p[kt][r]; }
    sum += partner(sum);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) p[kt][r] *= inv;        // HGATE.py:105
    return nz;
}

template <typename T, int HD> struct BlkCfg {
    static constexpr int LDW = HD + 4;
    static constexpr int NT = HD / 32;
    static constexpr int EPV = io<T>::EPV;
    static constexpr int CPR = HD / EPV;       // 16-byte chunks per row
    static constexpr int RPI = 64 / CPR;       // rows per wave-wide load
    static constexpr int NLD = 32 / RPI;       // wave-wide loads per 32-row tile
    static constexpr int TILE = 32 * LDW;      // floats per LDS tile
};

// =============================================================== forward
template <typename T, int HD>
__global__ __launch_bounds__(64) void blk_attn_fwd_k(const T* __restrict__ qkv, T* __restrict__ o,
                                                     const uint32_t* __restrict__ maskbits, BlkGeom g,
                                                     int n_units) {
    using C = BlkCfg<T, HD>;
    constexpr int LDW = C::LDW, NT = C::NT, NLD = C::NLD, RPI = C::RPI, TILE = C::TILE;
    __shared__ __attribute__((aligned(16))) float smem[4 * TILE];      // Q0 Q1 K0 K1
    float* Qs = smem;
    float* Ks = smem + 2 * TILE;

    const int lane = threadIdx.x, lq = lane & 31, hh = lane >> 5;
    const int crow_l = lane / C::CPR, ccol = (lane % C::CPR) * C::EPV;
    const int64_t row3d = 3 * (int64_t)g.d;
    const u32x4 zero4 = {0u, 0u, 0u, 0u};

    int u = blockIdx.x;
    if (u >= n_units) return;

    u32x4 qr[2][NLD], kr[2][NLD];
    auto issue_qk = [&](const BUnit& un) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const int j = i * RPI + crow_l;                 // pad rows re-read row KJ-1 and are zeroed
                const bool ok = j < g.KJ;
                const T* p = qkv + (un.base[t] + (ok ? j : g.KJ - 1)) * row3d + un.head * HD + ccol;
                const u32x4 a = *reinterpret_cast<const u32x4*>(p);
                const u32x4 b = *reinterpret_cast<const u32x4*>(p + g.d);
                qr[t][i] = ok ? a : zero4;
                kr[t][i] = ok ? b : zero4;
            }
    };
    BUnit cur = decode_bunit(g, u);
    issue_qk(cur);

    for (; u < n_units; u += gridDim.x) {
        // -- stage Q (pre-scaled, HGATE.py:91) and K
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const int off = (t * 32 + i * RPI + crow_l) * LDW + ccol;
                chunk<T>::to_lds(Qs + off, qr[t][i], qk_scale<HD>());
                chunk<T>::to_lds(Ks + off, kr[t][i], 1.0f);
            }
        // -- V straight into the MFMA B-operand layout (pad rows = 0)
        float v[2][16][NT];
        {
            const T* vb = qkv + 2 * g.d + cur.head * HD + lq * NT;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) v[kt][r][nt] = 0.f;
                    if (crow(r, hh) < g.KJ) load_nt<T, NT>(vb + (cur.base[kt] + crow(r, hh)) * row3d, v[kt][r]);
                }
        }
        uint32_t mb[2][2];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            mb[qt][0] = maskbits[(cur.mrow + qt * 32 + lq) * 2];
            mb[qt][1] = maskbits[(cur.mrow + qt * 32 + lq) * 2 + 1];
        }
        const int un = u + gridDim.x;
        BUnit nxt = cur;
        if (un < n_units) { nxt = decode_bunit(g, un); issue_qk(nxt); }
        lds_fence();

#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            float s[2][16], p[2][16];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                f32x16 st = tile_xyT<HD, LDW>(Ks + kt * TILE, Qs + qt * TILE, lq, hh);
#pragma unroll
                for (int r = 0; r < 16; ++r) s[kt][r] = st[r];
            }
            masked_softmax64(s, p, mb[qt][0], mb[qt][1], hh, g.KJ);

            f32x16 oacc[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int i = 0; i < 16; ++i) oacc[nt][i] = 0.f;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        oacc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(p[kt][r], v[kt][r][nt], oacc[nt], 0, 0, 0);

            // lane (c=lq, hh), reg r -> O[q = crow(r,hh)][c*NT + nt]
            T* ob = o + cur.head * HD + lq * NT;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float ov[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) ov[nt] = oacc[nt][r];
                if (crow(r, hh) < g.KJ) store_nt<T, NT>(ob + (cur.base[qt] + crow(r, hh)) * (int64_t)g.d, ov);
            }
        }
        lds_fence();
        cur = nxt;
    }
}

// =============================================================== backward
template <typename T, int HD>
__global__ __launch_bounds__(64) void blk_attn_bwd_k(const T* __restrict__ qkv, const T* __restrict__ dO,
                                                     T* __restrict__ dqkv,
                                                     const uint32_t* __restrict__ maskbits, BlkGeom g,
                                                     int n_units) {
    using C = BlkCfg<T, HD>;
    constexpr int LDW = C::LDW, NT = C::NT, NLD = C::NLD, RPI = C::RPI, TILE = C::TILE;
    constexpr int TW = 66;                                   // scratch row stride: [32 q][64 key slots]
    __shared__ __attribute__((aligned(16))) float smem[8 * TILE + 32 * TW];   // Q0 Q1 K0 K1 V0 V1 G0 G1 | scratch
    float* Qs = smem;
    float* Ks = smem + 2 * TILE;
    float* Vs = smem + 4 * TILE;
    float* Gs = smem + 6 * TILE;
    float* Sc = smem + 8 * TILE;

    const int lane = threadIdx.x, lq = lane & 31, hh = lane >> 5;
    const int crow_l = lane / C::CPR, ccol = (lane % C::CPR) * C::EPV;
    const int64_t row3d = 3 * (int64_t)g.d;
    const u32x4 zero4 = {0u, 0u, 0u, 0u};

    int u = blockIdx.x;
    if (u >= n_units) return;

    u32x4 qr[2][NLD], kr[2][NLD], vr[2][NLD], gr[2][NLD];
    auto issue_qk = [&](const BUnit& un) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const int j = i * RPI + crow_l;                 // pad rows re-read row KJ-1 and are zeroed
                const bool ok = j < g.KJ;
                const T* p = qkv + (un.base[t] + (ok ? j : g.KJ - 1)) * row3d + un.head * HD + ccol;
                const u32x4 a = *reinterpret_cast<const u32x4*>(p);
                const u32x4 b = *reinterpret_cast<const u32x4*>(p + g.d);
                qr[t][i] = ok ? a : zero4;
                kr[t][i] = ok ? b : zero4;
            }
    };
    auto issue_vg = [&](const BUnit& un) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const int j = i * RPI + crow_l;
                const bool ok = j < g.KJ;
                const int64_t tk = un.base[t] + (ok ? j : g.KJ - 1);
                const u32x4 a = *reinterpret_cast<const u32x4*>(qkv + tk * row3d + 2 * g.d + un.head * HD + ccol);
                const u32x4 b = *reinterpret_cast<const u32x4*>(dO + tk * (int64_t)g.d + un.head * HD + ccol);
                vr[t][i] = ok ? a : zero4;
                gr[t][i] = ok ? b : zero4;
            }
    };
    BUnit cur = decode_bunit(g, u);
    issue_qk(cur);

    for (; u < n_units; u += gridDim.x) {
        // -- q, k of this unit -> LDS; v, dO requested now and landed behind the first S product
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const int off = (t * 32 + i * RPI + crow_l) * LDW + ccol;
                chunk<T>::to_lds(Qs + off, qr[t][i], qk_scale<HD>());
                chunk<T>::to_lds(Ks + off, kr[t][i], 1.0f);
            }
        issue_vg(cur);
        uint32_t mb[2][2];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            mb[qt][0] = maskbits[(cur.mrow + qt * 32 + lq) * 2];
            mb[qt][1] = maskbits[(cur.mrow + qt * 32 + lq) * 2 + 1];
        }
        lds_fence();

        f32x16 dk[2][NT], dv[2][NT];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int i = 0; i < 16; ++i) { dk[kt][nt][i] = 0.f; dv[kt][nt][i] = 0.f; }

        BUnit nxt = cur;
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            // recompute P for query tile qt (lane = query, regs = key slots)
            float s[2][16], p[2][16], ds[2][16];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                f32x16 st = tile_xyT<HD, LDW>(Ks + kt * TILE, Qs + qt * TILE, lq, hh);
#pragma unroll
                for (int r = 0; r < 16; ++r) s[kt][r] = st[r];
            }
            const uint32_t nz = masked_softmax64(s, p, mb[qt][0], mb[qt][1], hh, g.KJ);
            if (qt == 0) {
                // v, dO have landed: stage them
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int i = 0; i < NLD; ++i) {
                        const int off = (t * 32 + i * RPI + crow_l) * LDW + ccol;
                        chunk<T>::to_lds(Vs + off, vr[t][i], 1.0f);
                        chunk<T>::to_lds(Gs + off, gr[t][i], 1.0f);
                    }
                lds_fence();
            }
            // dP^T[key][q] = V dO^T ; dS = P (dP - delta) where the logit was kept
            {
                float delta = 0.f;
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) {
                    f32x16 dp = tile_xyT<HD, LDW>(Vs + kt * TILE, Gs + qt * TILE, lq, hh);
#pragma unroll
                    for (int r = 0; r < 16; ++r) { ds[kt][r] = dp[r]; delta += p[kt][r] * dp[r]; }
                }
                delta += partner(delta);
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        ds[kt][r] = ((nz >> (kt * 16 + r)) & 1u) ? p[kt][r] * (ds[kt][r] - delta) : 0.f;
            }
            f32x16 acc[NT];
            // dQ_qt = scale * sum_kt dS[kt] K_kt   (A = dS in registers: lane = q)
            tile_ay<HD, LDW, true>(ds[0], Ks, lq, hh, acc);
            tile_ay<HD, LDW, false>(ds[1], Ks + TILE, lq, hh, acc);
            {
                T* base = dqkv + cur.head * HD + lq * NT;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float ov[NT];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) ov[nt] = acc[nt][r] * qk_scale<HD>();
                    if (crow(r, hh) < g.KJ) store_nt<T, NT>(base + (cur.base[qt] + crow(r, hh)) * row3d, ov);
                }
            }
            if (qt == 1) {                                       // next unit's q, k land behind the dK / dV products
                const int un = u + gridDim.x;
                if (un < n_units) { nxt = decode_bunit(g, un); issue_qk(nxt); }
            }
            // transpose through the scratch tile: write [q][key slot], read [.][key = lane]
            auto put = [&](const float (&x)[2][16]) {
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        float* pp = Sc + lq * TW + kt * 32 + 8 * gq + 4 * hh;
                        f32x2 a0 = {x[kt][4 * gq], x[kt][4 * gq + 1]}, a1 = {x[kt][4 * gq + 2], x[kt][4 * gq + 3]};
                        reinterpret_cast<f32x2*>(pp)[0] = a0;
                        reinterpret_cast<f32x2*>(pp)[1] = a1;
                    }
            };
            float a[16];
            // dK_kt += dS[kt]^T (scale*Q_qt)   (Qs already holds scale*Q)
            put(ds);
            lds_fence();
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) a[r] = Sc[crow(r, hh) * TW + kt * 32 + lq];
                tile_ay<HD, LDW, false>(a, Qs + qt * TILE, lq, hh, dk[kt]);
            }
            lds_fence();
            // dV_kt += P[kt]^T dO_qt
            put(p);
            lds_fence();
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) a[r] = Sc[crow(r, hh) * TW + kt * 32 + lq];
                tile_ay<HD, LDW, false>(a, Gs + qt * TILE, lq, hh, dv[kt]);
            }
            lds_fence();
        }
        // lane (c=lq, hh), reg r -> row key = crow(r,hh) of tile kt
        {
            T* base = dqkv + cur.head * HD + lq * NT;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float kv[NT], vv[NT];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) { kv[nt] = dk[kt][nt][r]; vv[nt] = dv[kt][nt][r]; }
                    if (crow(r, hh) < g.KJ) {
                        T* row = base + (cur.base[kt] + crow(r, hh)) * row3d;
                        store_nt<T, NT>(row + g.d, kv);
                        store_nt<T, NT>(row + 2 * g.d, vv);
                    }
                }
        }
        cur = nxt;
    }
}

bool bgeom_ok(int B, int F, int KJ, int nH, int hd) {
    return B > 0 && F > 0 && (F % 2) == 0 && KJ > 0 && KJ <= 32 && nH > 0 && (hd == 32 || hd == 64);
}

constexpr int LDS_PER_CU = 160 * 1024;

template <typename T, int HD>
int launch_bfwd(const void* qkv, void* o, const uint32_t* mb, BlkGeom g, int n_units, hipStream_t st) {
    constexpr int per_cu = LDS_PER_CU / (4 * BlkCfg<T, HD>::TILE * 4);
    const int blocks = min(n_units, 256 * (per_cu > 8 ? 8 : per_cu));
    blk_attn_fwd_k<T, HD><<<blocks, 64, 0, st>>>((const T*)qkv, (T*)o, mb, g, n_units);
    HWGAT_LAUNCH_CHECK();
}
template <typename T, int HD>
int launch_bbwd(const void* qkv, const void* dO, void* dqkv, const uint32_t* mb, BlkGeom g, int n_units,
                hipStream_t st) {
    constexpr int per_cu = LDS_PER_CU / ((8 * BlkCfg<T, HD>::TILE + 32 * 66) * 4);
    const int blocks = min(n_units, 256 * (per_cu > 4 ? 4 : per_cu));
    blk_attn_bwd_k<T, HD><<<blocks, 64, 0, st>>>((const T*)qkv, (const T*)dO, (T*)dqkv, mb, g, n_units);
    HWGAT_LAUNCH_CHECK();
}

}  // namespace

extern "C" int hwgat_blk_attn_fwd(const void* qkv, void* o, const uint32_t* maskbits, int B, int F, int KJ,
                                  int nH, int hd, int shifted, int dtype, void* stream) {
    if (!qkv || !o || !maskbits) return HWGAT_EINVAL;
    if (!bgeom_ok(B, F, KJ, nH, hd)) return HWGAT_ESHAPE;
    BlkGeom g{F, KJ, nH, F / 2, nH * hd, shifted ? 1 : 0};
    const int64_t units = (int64_t)B * g.f * nH;
    if (units > 0x7fffffff) return HWGAT_ESHAPE;
    hipStream_t st = (hipStream_t)stream;
#define FWD(T)                                                                              \
    switch (hd) {                                                                           \
        case 32: return launch_bfwd<T, 32>(qkv, o, maskbits, g, (int)units, st);            \
        default: return launch_bfwd<T, 64>(qkv, o, maskbits, g, (int)units, st);            \
    }
    if (dtype == HWGAT_F32) { FWD(float) }
    if (dtype == HWGAT_BF16) { FWD(bf16_t) }
#undef FWD
    return HWGAT_EDTYPE;
}

extern "C" int hwgat_blk_attn_bwd(const void* qkv, const void* dO, void* dqkv, const uint32_t* maskbits,
                                  int B, int F, int KJ, int nH, int hd, int shifted, int dtype,
                                  void* stream) {
    if (!qkv || !dO || !dqkv || !maskbits) return HWGAT_EINVAL;
    if (!bgeom_ok(B, F, KJ, nH, hd)) return HWGAT_ESHAPE;
    BlkGeom g{F, KJ, nH, F / 2, nH * hd, shifted ? 1 : 0};
    const int64_t units = (int64_t)B * g.f * nH;
    if (units > 0x7fffffff) return HWGAT_ESHAPE;
    hipStream_t st = (hipStream_t)stream;
#define BWD(T)                                                                                    \
    switch (hd) {                                                                                 \
        case 32: return launch_bbwd<T, 32>(qkv, dO, dqkv, maskbits, g, (int)units, st);           \
        default: return launch_bbwd<T, 64>(qkv, dO, dqkv, maskbits, g, (int)units, st);           \
    }
    if (dtype == HWGAT_F32) { BWD(float) }
    if (dtype == HWGAT_BF16) { BWD(bf16_t) }
#undef BWD
    return HWGAT_EDTYPE;
}
