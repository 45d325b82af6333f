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
// A workgroup of two wavefronts owns a unit: wave w loads frame tile w and owns query tile w (and, in
// the second half of the backward pass, key tile w), so the per-wave register and LDS footprints are
// half a unit's and 4 (bwd) / 8 (fwd) waves fit a CU.  The backward pass keeps Q, K, dO in LDS (V only
// feeds dP and goes straight to registers); P^T and dS^T meet in one shared scratch tile.
//
// HBM traffic is the algorithmic 4*E*s (fwd) / 7*E*s (bwd) like the window kernel; the MFMA work per
// byte is 2 x (64/58)^2 higher (64-slot tiles for 58 tokens), so at head_dim 64 the backward pass sits
// at the fp32 MFMA / HBM ridge rather than clearly HBM-bound (HGATE section).
#include "attn_common.h"
#include "blk_common.h"
#include "fused_ops.h"            // the dropout hash (attention dropout, HGATE.py:78,106)

namespace {
using namespace blk;

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

// keep[kt][r] = 1/(1-p) or 0 for P[q][key slot kt*32 + crow(r,hh)] of unit u: element ((u * N2 + q) * N2 + key) of the
// reference's (B f, nH, N2, N2) attention tensor, N2 = 2 KJ tokens per block, token = frame * KJ + joint (HGATE.py:87-106)
__device__ __forceinline__ void blk_keep(float (&k)[2][16], const AttnDrop& ad, int u, int q_tok, int hh, int KJ) {
    const uint32_t thresh = drop_thresh(ad.p);
    const float scale = 1.0f / (1.0f - ad.p);
    const uint64_t row = ((uint64_t)u * (2 * KJ) + q_tok) * (2 * KJ);
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = crow(r, hh);
            k[kt][r] = j < KJ ? drop_keep(ad.seed, row + kt * KJ + j, thresh, scale) : 0.f;
        }
}

template <typename T, int HD> struct BlkCfg {
    using E = typename tile_of<T>::E;          // LDS tile element: fp32 tiles + fp32 MFMAs, or raw bf16 tiles + bf16 MFMAs (attn_common.h)
    static constexpr int LDW = HD + tile_of<T>::PAD;
    static constexpr int NT = HD / 32;
    static constexpr int EPV = io<T>::EPV;
    static constexpr int CPR = HD / EPV;       // 16-byte chunks per row
    static constexpr int RPI = 64 / CPR;       // rows per wave-wide load
    static constexpr int NLD = 32 / RPI;       // wave-wide loads per 32-row tile
    static constexpr int TILE = 32 * LDW;      // elements per LDS tile
};

// =============================================================== forward
// Two wavefronts share a unit: wave w loads frame tile w of Q and K and owns query tile w.
template <typename T, int HD, bool ADROP = false>
__global__ __launch_bounds__(128, 2) void blk_attn_fwd_k(const T* __restrict__ qkv, T* __restrict__ o,
                                                      const uint32_t* __restrict__ maskbits, BlkGeom g,
                                                      int n_units, AttnDrop ad) {
    if constexpr (ADROP) ad.seed += seed_base_of(ad.base);
    using C = BlkCfg<T, HD>;
    constexpr int LDW = C::LDW, NT = C::NT, NLD = C::NLD, RPI = C::RPI, TILE = C::TILE;
    using E = typename C::E;
    constexpr bool QS = tile_of<T>::QSCALED;
    __shared__ __attribute__((aligned(16))) E smem[6 * TILE];          // Q0 Q1 K0 K1 V0 V1
    E* Qs = smem;
    E* Ks = smem + 2 * TILE;
    E* Vs = smem + 4 * TILE;

    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, lq = lane & 31, hh = lane >> 5;
    const int crow_l = lane / C::CPR, ccol = (lane % C::CPR) * C::EPV;
    const int64_t row3d = 3 * (int64_t)g.d;
    const u32x4 zero4 = {0u, 0u, 0u, 0u};

    int u = blockIdx.x;
    if (u >= n_units) return;

    u32x4 qr[NLD], kr[NLD], vr[NLD];
    auto issue_qk = [&](const BUnit& un) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int j = i * RPI + crow_l;                     // pad rows re-read row KJ-1 and are zeroed
            const bool ok = j < g.KJ;
            const T* p = qkv + ((w ? un.base[1] : un.base[0]) + (ok ? j : g.KJ - 1)) * row3d + un.head * HD + ccol;
            const u32x4 a = *reinterpret_cast<const u32x4*>(p);
            const u32x4 b = *reinterpret_cast<const u32x4*>(p + g.d);
            const u32x4 c = *reinterpret_cast<const u32x4*>(p + 2 * g.d);
            qr[i] = ok ? a : zero4;
            kr[i] = ok ? b : zero4;
            vr[i] = ok ? c : zero4;
        }
    };
    BUnit cur = decode_bunit(g, u);
    issue_qk(cur);

    for (; u < n_units; u += gridDim.x) {
        // -- stage Q (pre-scaled, HGATE.py:91), K and V: every element is fetched by exactly one wave
        //    (both waves pulling V into registers re-read 19 % of the bytes from HBM: PMC)
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int off = (w * 32 + i * RPI + crow_l) * LDW + ccol;
            raw_to_lds(Qs + off, qr[i], qk_scale<HD>(), T());
            raw_to_lds(Ks + off, kr[i], 1.0f, T());
            raw_to_lds(Vs + off, vr[i], 1.0f, T());
        }
        const uint32_t mb0 = maskbits[(cur.mrow + w * 32 + lq) * 2];
        const uint32_t mb1 = maskbits[(cur.mrow + w * 32 + lq) * 2 + 1];
        const int64_t obase = w ? cur.base[1] : cur.base[0];
        const int ohead = cur.head;
        const int un = u + gridDim.x;
        if (un < n_units) { cur = decode_bunit(g, un); issue_qk(cur); }
        __syncthreads();

        float s[2][16], p[2][16];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            f32x16 st = tile_xyT<HD, LDW>(Ks + kt * TILE, Qs + w * TILE, lq, hh);
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kt][r] = QS ? st[r] : st[r] * qk_scale<HD>();
        }
        masked_softmax64(s, p, mb0, mb1, hh, g.KJ);
        if constexpr (ADROP) {                                  // HGATE.py:106
            float keep[2][16];
            blk_keep(keep, ad, u, w * g.KJ + lq, hh, g.KJ);
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) p[kt][r] *= keep[kt][r];
        }

        f32x16 oacc[NT];
        tile_ay<HD, LDW, true>(p[0], Vs, lq, hh, oacc);           // P fed back as the A operand, V in the B layout from LDS
        tile_ay<HD, LDW, false>(p[1], Vs + TILE, lq, hh, oacc);

        // lane (c=lq, hh), reg r -> O[q = crow(r,hh)][c*NT + nt]
        T* ob = o + ohead * HD + lq * NT;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float ov[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) ov[nt] = oacc[nt][r];
            if (crow(r, hh) < g.KJ) store_nt<T, NT>(ob + (obase + crow(r, hh)) * (int64_t)g.d, ov);
        }
        __syncthreads();                                         // both waves are done with this unit's tiles
    }
}

// =============================================================== backward
// Two wavefronts share a unit.  Phase A: wave w owns QUERY tile w (recompute P, dP, dS, dQ_w).  The
// transposed dS and P tiles meet in a shared scratch tile.  Phase B: wave w owns KEY tile w
// (dK_w = sum_qt dS^T Q_qt, dV_w = sum_qt P^T dO_qt).  Q, K, dO live in LDS; V is only ever the
// row-per-lane operand of dP, so it goes from HBM straight into registers.
template <typename T, int HD, bool ADROP = false>
__global__ __launch_bounds__(128, sizeof(T) == 2 ? 2 : 1) void blk_attn_bwd_k(const T* __restrict__ qkv, const T* __restrict__ dO,
                                                      T* __restrict__ dqkv,
                                                      const uint32_t* __restrict__ maskbits, BlkGeom g,
                                                      int n_units, AttnDrop ad) {
    if constexpr (ADROP) ad.seed += seed_base_of(ad.base);
    using C = BlkCfg<T, HD>;
    constexpr int LDW = C::LDW, NT = C::NT, NLD = C::NLD, RPI = C::RPI, TILE = C::TILE;
    using E = typename C::E;
    constexpr bool QS = tile_of<T>::QSCALED, B16 = sizeof(T) == 2;
    constexpr int TW = 66;                                   // scratch row stride: [64 q slots][64 key slots]
    __shared__ __attribute__((aligned(16))) E smem[6 * TILE + 64 * TW];   // Q0 Q1 K0 K1 G0 G1 | scratch
    E* Qs = smem;
    E* Ks = smem + 2 * TILE;
    E* Gs = smem + 4 * TILE;
    E* Sc = smem + 6 * TILE;

    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, lq = lane & 31, hh = lane >> 5;
    const int crow_l = lane / C::CPR, ccol = (lane % C::CPR) * C::EPV;
    const int64_t row3d = 3 * (int64_t)g.d;
    const u32x4 zero4 = {0u, 0u, 0u, 0u};

    int u = blockIdx.x;
    if (u >= n_units) return;

    u32x4 qr[NLD], kr[NLD], gr[NLD];
    auto issue_qk = [&](const BUnit& un) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int j = i * RPI + crow_l;                     // pad rows re-read row KJ-1 and are zeroed
            const bool ok = j < g.KJ;
            const T* p = qkv + ((w ? un.base[1] : un.base[0]) + (ok ? j : g.KJ - 1)) * row3d + un.head * HD + ccol;
            const u32x4 a = *reinterpret_cast<const u32x4*>(p);
            const u32x4 b = *reinterpret_cast<const u32x4*>(p + g.d);
            qr[i] = ok ? a : zero4;
            kr[i] = ok ? b : zero4;
        }
    };
    BUnit cur = decode_bunit(g, u);
    issue_qk(cur);

    for (; u < n_units; u += gridDim.x) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int off = (w * 32 + i * RPI + crow_l) * LDW + ccol;
            raw_to_lds(Qs + off, qr[i], qk_scale<HD>(), T());
            raw_to_lds(Ks + off, kr[i], 1.0f, T());
        }
        // dO tile w (to LDS behind the S product) and V rows in the row-per-lane operand layout
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int j = i * RPI + crow_l;
            const bool ok = j < g.KJ;
            const u32x4 a = *reinterpret_cast<const u32x4*>(dO + ((w ? cur.base[1] : cur.base[0]) + (ok ? j : g.KJ - 1)) * (int64_t)g.d
                                                            + cur.head * HD + ccol);
            gr[i] = ok ? a : zero4;
        }
        float vx[B16 ? 1 : 2][B16 ? 1 : HD / 8][4];              // fp32: V[key = kt*32 + lq][8m + 4hh + 0..3]
        u32x4 vxb[B16 ? 2 : 1][B16 ? HD / 16 : 1];               // bf16: V[key][16m + 8hh + 0..7], raw = one MFMA operand
        {
            const bool ok = lq < g.KJ;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                const T* vp = qkv + (cur.base[kt] + (ok ? lq : g.KJ - 1)) * row3d + 2 * g.d + cur.head * HD;
                if constexpr (B16) {
#pragma unroll
                    for (int m = 0; m < HD / 16; ++m) {
                        const u32x4 t = *reinterpret_cast<const u32x4*>(vp + 16 * m + 8 * hh);
                        vxb[kt][m] = ok ? t : zero4;
                    }
                } else {
#pragma unroll
                    for (int m = 0; m < HD / 8; ++m) {
                        load_nt<T, 4>(vp + 4 * hh + 8 * m, vx[kt][m]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) vx[kt][m][e] = ok ? vx[kt][m][e] : 0.f;
                    }
                }
            }
        }
        const uint32_t mb0 = maskbits[(cur.mrow + w * 32 + lq) * 2];
        const uint32_t mb1 = maskbits[(cur.mrow + w * 32 + lq) * 2 + 1];
        const int64_t base_w = w ? cur.base[1] : cur.base[0];
        const int head = cur.head;
        __syncthreads();                                         // (1) Q, K of both frames staged

        // ---------------- phase A: query tile w
        float s[2][16], p[2][16], ds[2][16];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            f32x16 st = tile_xyT<HD, LDW>(Ks + kt * TILE, Qs + w * TILE, lq, hh);
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kt][r] = QS ? st[r] : st[r] * qk_scale<HD>();
        }
        const uint32_t nz = masked_softmax64(s, p, mb0, mb1, hh, g.KJ);
        // attention dropout: A = D o P went into O = A V, so dP = D o dA (dA = dO V^T) and dV = A^T dO; the mask is recomputed
        float keep[ADROP ? 2 : 1][ADROP ? 16 : 1];
        if constexpr (ADROP) blk_keep(keep, ad, u, w * g.KJ + lq, hh, g.KJ);
#pragma unroll
        for (int i = 0; i < NLD; ++i)
            raw_to_lds(Gs + (w * 32 + i * RPI + crow_l) * LDW + ccol, gr[i], 1.0f, T());
        lds_fence();                                             // own dO tile visible to this wave
        // dP^T[key][q] = V dO_w^T ; dS = P (dP - delta) where the logit was kept
        {
            float delta = 0.f;
            const E* yr = Gs + (w * 32 + lq) * LDW + (B16 ? 8 : 4) * hh;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                f32x16 dp;
#pragma unroll
                for (int i = 0; i < 16; ++i) dp[i] = 0.f;
                if constexpr (B16) {
#pragma unroll
                    for (int m = 0; m < HD / 16; ++m) {
                        const bf16x8 yf = *reinterpret_cast<const bf16x8*>(yr + 16 * m);
                        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vxb[kt][m]), yf, dp, 0, 0, 0);
                    }
                } else {
#pragma unroll
                    for (int m = 0; m < HD / 8; ++m) {
                        const f32x4 yf = *reinterpret_cast<const f32x4*>(yr + 8 * m);
                        dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vx[kt][m][0], yf.x, dp, 0, 0, 0);
                        dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vx[kt][m][1], yf.y, dp, 0, 0, 0);
                        dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vx[kt][m][2], yf.z, dp, 0, 0, 0);
                        dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vx[kt][m][3], yf.w, dp, 0, 0, 0);
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float dpr = dp[r];
                    if constexpr (ADROP) dpr *= keep[kt][r];
                    ds[kt][r] = dpr;
                    delta += p[kt][r] * dpr;
                }
            }
            delta += partner(delta);
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    ds[kt][r] = ((nz >> (kt * 16 + r)) & 1u) ? p[kt][r] * (ds[kt][r] - delta) : 0.f;
        }
        // next unit's q, k land behind the remaining products
        {
            const int un = u + gridDim.x;
            if (un < n_units) { cur = decode_bunit(g, un); issue_qk(cur); }
        }
        f32x16 acc[NT];
        // dQ_w = scale * sum_kt dS[kt] K_kt   (A = dS in registers: lane = q)
        tile_ay<HD, LDW, true>(ds[0], Ks, lq, hh, acc);
        tile_ay<HD, LDW, false>(ds[1], Ks + TILE, lq, hh, acc);
        T* gbase = dqkv + head * HD + lq * NT;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float ov[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) ov[nt] = acc[nt][r] * qk_scale<HD>();
            if (crow(r, hh) < g.KJ) store_nt<T, NT>(gbase + (base_w + crow(r, hh)) * row3d, ov);
        }
        // scratch [q slot][key slot]: this wave writes its 32 query rows
        auto put = [&](const float (&x)[2][16]) {
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) put4(Sc + (w * 32 + lq) * TW + kt * 32 + 8 * gq + 4 * hh, x[kt] + 4 * gq);
        };
        put(ds);
        __syncthreads();                                         // (2) dS of both query tiles + both dO tiles visible

        // ---------------- phase B: key tile w
        float a[16];
        // dK_w = sum_qt dS[qt][kt = w]^T (scale*Q_qt)   (Qs already holds scale*Q)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) a[r] = (float)Sc[(qt * 32 + crow(r, hh)) * TW + w * 32 + lq];
            if (qt == 0) tile_ay<HD, LDW, true>(a, Qs, lq, hh, acc);
            else tile_ay<HD, LDW, false>(a, Qs + TILE, lq, hh, acc);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float ov[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) ov[nt] = QS ? acc[nt][r] : acc[nt][r] * qk_scale<HD>();
            if (crow(r, hh) < g.KJ) store_nt<T, NT>(gbase + (base_w + crow(r, hh)) * row3d + g.d, ov);
        }
        __syncthreads();                                         // (3) dS scratch consumed
        if constexpr (ADROP) {
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) p[kt][r] *= keep[kt][r];
        }
        put(p);
        __syncthreads();                                         // (4) P of both query tiles visible
        // dV_w = sum_qt P[qt][kt = w]^T dO_qt
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) a[r] = (float)Sc[(qt * 32 + crow(r, hh)) * TW + w * 32 + lq];
            if (qt == 0) tile_ay<HD, LDW, true>(a, Gs, lq, hh, acc);
            else tile_ay<HD, LDW, false>(a, Gs + TILE, lq, hh, acc);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float ov[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) ov[nt] = acc[nt][r];
            if (crow(r, hh) < g.KJ) store_nt<T, NT>(gbase + (base_w + crow(r, hh)) * row3d + 2 * g.d, ov);
        }
        __syncthreads();                                         // (5) tiles + scratch free for the next unit
    }
}

bool bgeom_ok(int B, int F, int KJ, int nH, int hd) {
    return B > 0 && F > 0 && (F % 2) == 0 && KJ > 0 && KJ <= 32 && nH > 0 && (hd == 32 || hd == 64);
}

constexpr int LDS_PER_CU = 160 * 1024;

template <typename T, int HD>
int launch_bfwd(const void* qkv, void* o, const uint32_t* mb, BlkGeom g, int n_units, AttnDrop ad, hipStream_t st) {
    constexpr int per_cu = LDS_PER_CU / (6 * BlkCfg<T, HD>::TILE * (int)sizeof(typename BlkCfg<T, HD>::E));
    const int blocks = min(n_units, 256 * (per_cu > 8 ? 8 : per_cu));
    if (ad.p > 0.f) blk_attn_fwd_k<T, HD, true><<<blocks, 128, 0, st>>>((const T*)qkv, (T*)o, mb, g, n_units, ad);
    else blk_attn_fwd_k<T, HD><<<blocks, 128, 0, st>>>((const T*)qkv, (T*)o, mb, g, n_units, ad);
    HWGAT_LAUNCH_CHECK();
}
template <typename T, int HD>
int launch_bbwd(const void* qkv, const void* dO, void* dqkv, const uint32_t* mb, BlkGeom g, int n_units, AttnDrop ad,
                hipStream_t st) {
    constexpr int per_cu = LDS_PER_CU / ((6 * BlkCfg<T, HD>::TILE + 64 * 66) * (int)sizeof(typename BlkCfg<T, HD>::E));
    const int blocks = min(n_units, 256 * (per_cu > 4 ? 4 : per_cu));
    if (ad.p > 0.f) blk_attn_bwd_k<T, HD, true><<<blocks, 128, 0, st>>>((const T*)qkv, (const T*)dO, (T*)dqkv, mb, g, n_units, ad);
    else blk_attn_bwd_k<T, HD><<<blocks, 128, 0, st>>>((const T*)qkv, (const T*)dO, (T*)dqkv, mb, g, n_units, ad);
    HWGAT_LAUNCH_CHECK();
}

}  // namespace

extern "C" int hwgat_blk_attn_fwd_drop(const void* qkv, void* o, const uint32_t* maskbits, int B, int F, int KJ,
                                       int nH, int hd, int shifted, int dtype, uint32_t drop_seed, float drop_p,
                                       const uint32_t* seed_base, void* stream) {
    if (!qkv || !o || !maskbits || drop_p < 0.f || drop_p >= 1.f) return HWGAT_EINVAL;
    const AttnDrop ad = make_drop(drop_seed, drop_p, seed_base);
    if (!bgeom_ok(B, F, KJ, nH, hd)) return HWGAT_ESHAPE;
    BlkGeom g{F, KJ, nH, F / 2, nH * hd, shifted ? 1 : 0};
    const int64_t units = (int64_t)B * g.f * nH;
    if (units > 0x7fffffff) return HWGAT_ESHAPE;
    hipStream_t st = (hipStream_t)stream;
#define FWD(T)                                                                              \
    switch (hd) {                                                                           \
        case 32: return launch_bfwd<T, 32>(qkv, o, maskbits, g, (int)units, ad, st);        \
        default: return launch_bfwd<T, 64>(qkv, o, maskbits, g, (int)units, ad, st);        \
    }
    if (dtype == HWGAT_F32) {
        // head_dim 64 (every HGATE stage): the 16x16-tile, four-waves-per-unit kernels of blk_attn_f32.hip / blk_attn_bf16.hip;
        // the 32x32-tile form stays for head_dim 32 and as the lab A/B (HWGAT_BLK_F32=0, HWGAT_BLK_B16=0)
        static const bool old_f32 = lab_env("HWGAT_BLK_F32") && lab_env("HWGAT_BLK_F32")[0] == '0';
        if (hd == 64 && !old_f32) return hwgat_launch_blk_fwd_f32(qkv, o, maskbits, B, F, KJ, nH, shifted, ad.seed, ad.p, ad.base, st);
        FWD(float)
    }
    if (dtype == HWGAT_BF16) {
        static const bool old_b16 = lab_env("HWGAT_BLK_B16") && lab_env("HWGAT_BLK_B16")[0] == '0';     // see hwgat_blk_attn_bwd
        if (hd == 64 && !old_b16) return hwgat_launch_blk_fwd_b16(qkv, o, maskbits, B, F, KJ, nH, shifted, ad.seed, ad.p, ad.base, st);
        FWD(bf16_t)
    }
#undef FWD
    return HWGAT_EDTYPE;
}

extern "C" int hwgat_blk_attn_fwd(const void* qkv, void* o, const uint32_t* maskbits, int B, int F, int KJ,
                                  int nH, int hd, int shifted, int dtype, void* stream) {
    return hwgat_blk_attn_fwd_drop(qkv, o, maskbits, B, F, KJ, nH, hd, shifted, dtype, 0u, 0.f, nullptr, stream);
}

extern "C" int hwgat_blk_attn_bwd_drop(const void* qkv, const void* dO, void* dqkv, const uint32_t* maskbits,
                                       int B, int F, int KJ, int nH, int hd, int shifted, int dtype,
                                       uint32_t drop_seed, float drop_p, const uint32_t* seed_base, void* stream) {
    if (!qkv || !dO || !dqkv || !maskbits || drop_p < 0.f || drop_p >= 1.f) return HWGAT_EINVAL;
    const AttnDrop ad = make_drop(drop_seed, drop_p, seed_base);
    if (!bgeom_ok(B, F, KJ, nH, hd)) return HWGAT_ESHAPE;
    BlkGeom g{F, KJ, nH, F / 2, nH * hd, shifted ? 1 : 0};
    const int64_t units = (int64_t)B * g.f * nH;
    if (units > 0x7fffffff) return HWGAT_ESHAPE;
    hipStream_t st = (hipStream_t)stream;
#define BWD(T)                                                                                    \
    switch (hd) {                                                                                 \
        case 32: return launch_bbwd<T, 32>(qkv, dO, dqkv, maskbits, g, (int)units, ad, st);       \
        default: return launch_bbwd<T, 64>(qkv, dO, dqkv, maskbits, g, (int)units, ad, st);       \
    }
    if (dtype == HWGAT_F32) {
        static const bool old_f32 = lab_env("HWGAT_BLK_F32") && lab_env("HWGAT_BLK_F32")[0] == '0';
        if (hd == 64 && !old_f32) return hwgat_launch_blk_bwd_f32(qkv, dO, dqkv, maskbits, B, F, KJ, nH, shifted, ad.seed, ad.p, ad.base, st);
        BWD(float)
    }
    if (dtype == HWGAT_BF16) {
        // head_dim 64 (every HGATE stage): the 16x16-tile, four-waves-per-unit kernel of blk_attn_bf16.hip; the 32x32-tile
        // form stays for head_dim 32 and as the lab A/B (HWGAT_BLK_B16=0)
        static const bool old_b16 = lab_env("HWGAT_BLK_B16") && lab_env("HWGAT_BLK_B16")[0] == '0';
        if (hd == 64 && !old_b16) return hwgat_launch_blk_bwd_b16(qkv, dO, dqkv, maskbits, B, F, KJ, nH, shifted, ad.seed, ad.p, ad.base, st);
        BWD(bf16_t)
    }
#undef BWD
    return HWGAT_EDTYPE;
}

extern "C" int hwgat_blk_attn_bwd(const void* qkv, const void* dO, void* dqkv, const uint32_t* maskbits,
                                  int B, int F, int KJ, int nH, int hd, int shifted, int dtype,
                                  void* stream) {
    return hwgat_blk_attn_bwd_drop(qkv, dO, dqkv, maskbits, B, F, KJ, nH, hd, shifted, dtype, 0u, 0.f, nullptr, stream);
}
