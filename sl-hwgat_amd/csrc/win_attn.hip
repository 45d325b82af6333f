// Fused window graph-attention for HWGAT on gfx950 (MI355X).
//
// Replaces MSA.forward's attention core (reference hwgat/models/HWGATE.py:89-114)
// together with window_partition / window_reverse / torch.roll
// (HWGATE.py:30-47,197-215), which become index arithmetic here.
//
// Work unit = one (window, head): 32 tokens (2 frames x 16 joints of one body
// part window) x head_dim.  One 64-lane wavefront owns a unit end to end, with a
// wave-private LDS region, so the kernel has no workgroup barriers at all.
//
//   S^T = K Q^T            32 x v_mfma_f32_32x32x2_f32 (exact fp32 fma chains);
//                          computed "swapped" so that a lane holds 16 keys of
//                          ONE query row (partner lane^32 holds the other 16):
//                          row max / row sum are 15 in-lane ops + 1 shuffle.
//   masks + softmax        in registers (adjacency / shift bit rows, the train
//                          mode probability-threshold drop, the "== 0 -> -10000"
//                          fill, exactly as SURVEY.md 8a lists them)
//   O = P V                P stays in the accumulator layout and is fed straight
//                          back as the MFMA A operand (k-step r pairs keys
//                          crow(r,0), crow(r,1)); V streams from HBM directly
//                          into the B-operand layout, never touching LDS.
//
// HBM traffic is exactly the algorithmic 4*E*s (fwd) / 7*E*s (bwd): every q,k,v
// (and dO) element is read once, every o (dq,dk,dv) element written once; all
// rows are 128-byte-line aligned segments of hd*s bytes.
#include <stdlib.h>
#include <type_traits>
#include "attn_common.h"
#include "fused_ops.h"            // the dropout hash (attention dropout)

namespace {

struct WinGeom {
    int F, K, nW, nH, f, d, shift;
};

// ---- unit decoding ---------------------------------------------------------
struct Unit {
    int64_t base0, base1;   // token index of slot 0 of frame A / frame B
    int head, mrow;         // head index, row offset into maskbits
};
__device__ __forceinline__ Unit decode_unit(const WinGeom& g, int u) {
    Unit r;
    const int n = u / g.nH;
    r.head = u - n * g.nH;
    const int wi = n % g.nW;
    const int t2 = n / g.nW;
    const int fi = t2 % g.f;
    const int b = t2 / g.f;
    int fa = 2 * fi + g.shift, fb = fa + 1;          // torch.roll(x, -shift): shifted[t] = x[(t+shift) % F]
    if (fa >= g.F) fa -= g.F;
    if (fb >= g.F) fb -= g.F;
    r.base0 = ((int64_t)b * g.F + fa) * g.K + wi * 16;
    r.base1 = ((int64_t)b * g.F + fb) * g.K + wi * 16;
    r.mrow = ((g.shift && fi == g.f - 1) ? g.nW : 0) * 32 + wi * 32;
    return r;
}
__device__ __forceinline__ int64_t tok_of(const Unit& u, int t) {
    return (t < 16 ? u.base0 : u.base1) + (t & 15);
}


// masks + softmax on one lane's 16 logits of query row (lane & 31).
// s[r] = S[q][crow(r, hh)] on entry; p[r] = final probability on exit.
// returns bit r set where the gradient flows (entry was not replaced by -10000).
template <bool TRAIN>
__device__ __forceinline__ uint32_t masked_softmax(float (&s)[16], float (&p)[16], uint32_t mrow_bits,
                                                   int hh, float thr) {
    if constexpr (TRAIN) {                                  // HWGATE.py:94-100
        float m0 = s[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) m0 = fmaxf(m0, s[r]);
        m0 = fmaxf(m0, partner(m0));
        float e[16], sum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { e[r] = sm_exp(s[r] - m0); sum += e[r]; }
        sum += partner(sum);
        const float cut = thr * sum;                        // e / sum > thr  <=>  e > thr * sum (sum > 0): no 16 divisions
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (e[r] > cut) s[r] = 0.f;
    }
    uint32_t nz = 0;
    float m = -3.0e38f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const bool vis = (mrow_bits >> crow(r, hh)) & 1u;   // HWGATE.py:102-108
        float v = vis ? s[r] : 0.f;
        if (v == 0.f) v = -10000.f; else nz |= 1u << r;     // HWGATE.py:110
        s[r] = v;
        m = fmaxf(m, v);
    }
    m = fmaxf(m, partner(m));
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { p[r] = sm_exp(s[r] - m); sum += p[r]; }
    sum += partner(sum);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int r = 0; r < 16; ++r) p[r] *= inv;               // HWGATE.py:111
    return nz;
}


// Attention dropout (reference HWGATE.py:78,112: nn.Dropout on the softmax output, train mode).  keep[r] = 1/(1-p) or 0
// for this lane's 16 probabilities P[q = lq][key = crow(r, hh)] of unit u.  The element index is the one of the
// reference's (B_ = B f nW, nH, 32, 32) attention tensor, (u * 32 + q) * 32 + key, hashed like every other dropout
// site (fused_ops.h): hwgat_dropout_mask_f32(out, units * 1024, seed, p) IS this mask (tests feed it to the oracle).
__device__ __forceinline__ void attn_keep(float (&k)[16], const AttnDrop& ad, int u, int lq, int hh) {
    const uint64_t base = ((uint64_t)u * 32 + lq) * 32 + 4 * hh;
    const uint32_t thresh = drop_thresh(ad.p);
    const float scale = 1.0f / (1.0f - ad.p);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f32x4 t = drop_keep4(ad.seed, base + 8 * j, thresh, scale);
        k[4 * j] = t.x; k[4 * j + 1] = t.y; k[4 * j + 2] = t.z; k[4 * j + 3] = t.w;
    }
}

// =============================================================== forward
template <typename T, int HD, bool TRAIN, bool ADROP = false>
__global__ __launch_bounds__(256, 2) void win_attn_fwd_k(const T* __restrict__ qkv, T* __restrict__ o,
                                                         const uint32_t* __restrict__ maskbits,
                                                         const float* __restrict__ thr_p, WinGeom g,
                                                         int n_units, AttnDrop ad) {
    if constexpr (ADROP) ad.seed += seed_base_of(ad.base);
    using TL = tile_of<T>;             // fp32: fp32 tiles and MFMAs; bf16: raw bf16 tiles, v_mfma_f32_32x32x16_bf16 (attn_common.h)
    using E = typename TL::E;
    constexpr bool B16 = sizeof(T) == 2;
    constexpr int LDW = HD + TL::PAD;
    constexpr int NT = HD / 32;
    constexpr int EPV = io<T>::EPV;
    constexpr int CPR = HD / EPV;      // 16-byte chunks per row
    constexpr int RPI = 64 / CPR;      // rows per wave-wide load
    constexpr int NLD = 32 / RPI;      // wave-wide loads per tile
    __shared__ __attribute__((aligned(16))) E smem[4 * 2 * 32 * LDW];

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;   // uniform: unit decoding in SGPRs
    const int lq = lane & 31, hh = lane >> 5;
    E* Qs = smem + wave * (2 * 32 * LDW);
    E* Ks = Qs + 32 * LDW;
    const int crow_l = lane / CPR, ccol = (lane % CPR) * EPV;
    const int64_t row3d = 3 * (int64_t)g.d;
    const float thr = TRAIN ? *thr_p : 0.f;

    const int nwaves = gridDim.x * 4;
    int u = blockIdx.x * 4 + wave;
    if (u >= n_units) return;

    u32x4 qr[NLD], kr[NLD];
    // V of the NEXT unit, already in the MFMA B-operand layout (fp32: converted floats; bf16: the raw pairs)
    typedef typename std::conditional<B16, brow<NT>, float[NT]>::type vrow_t;
    vrow_t vn[16];
    // the whole next unit is requested while this one computes: Q and K as 16-byte row chunks for the LDS tiles, V
    // straight into operand registers.  (V used to be requested at the top of its own unit and was needed ~2 700
    // cycles later -- less than an HBM round trip under load, so every unit stalled on it.)
    auto issue_qkv = [&](const Unit& un) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const T* p = qkv + tok_of(un, i * RPI + crow_l) * row3d + un.head * HD + ccol;
            qr[i] = load16_s<T>(p);
            kr[i] = load16_s<T>(p + g.d);
        }
        const T* vb = qkv + 2 * g.d + un.head * HD + lq * NT;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if constexpr (B16) vn[r].load_s(vb + tok_of(un, crow(r, hh)) * row3d);
            else load_nt_s<T, NT>(vb + tok_of(un, crow(r, hh)) * row3d, vn[r]);
        }
    };
    Unit cur = decode_unit(g, u);
    issue_qkv(cur);

    for (; u < n_units; u += nwaves) {
        // -- stage Q (pre-scaled, HWGATE.py:89) and K into the wave's LDS tiles
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            E* dq = Qs + (i * RPI + crow_l) * LDW + ccol;
            raw_to_lds(dq, qr[i], qk_scale<HD>(), T());
            raw_to_lds(dq + 32 * LDW, kr[i], 1.0f, T());
        }
        // -- V of this unit arrived with the previous prefetch
        vrow_t v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if constexpr (B16) v[r] = vn[r];
            else {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) v[r][nt] = vn[r][nt];
            }
        }
        const uint32_t mbits = maskbits[cur.mrow + lq];
        // -- prefetch the next unit's q, k, v while this one computes
        const int un = u + nwaves;
        Unit nxt = cur;
        if (un < n_units) { nxt = decode_unit(g, un); issue_qkv(nxt); }
        lds_fence();

        f32x16 st = tile_xyT<HD, LDW>(Ks, Qs, lq, hh);     // st[r] = S[q=lq][key=crow(r,hh)]
        float s[16], p[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = TL::QSCALED ? st[r] : st[r] * qk_scale<HD>();
        masked_softmax<TRAIN>(s, p, mbits, hh, thr);
        if constexpr (ADROP) {                                  // HWGATE.py:112
            float keep[16];
            attn_keep(keep, ad, u, lq, hh);
#pragma unroll
            for (int r = 0; r < 16; ++r) p[r] *= keep[r];
        }

        f32x16 oacc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) oacc[nt][i] = 0.f;
        if constexpr (B16) {
            mfma_ab<NT>(p, v, oacc);
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    oacc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(p[r], v[r][nt], oacc[nt], 0, 0, 0);
        }

        // lane (c=lq, hh), reg r -> O[q = crow(r,hh)][c*NT + nt]
        T* ob = o + cur.head * HD + lq * NT;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float ov[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) ov[nt] = oacc[nt][r];
            store_nt_s<T, NT>(ob + tok_of(cur, crow(r, hh)) * (int64_t)g.d, ov);
        }
        lds_fence();
        cur = nxt;
    }
}

// =============================================================== backward

// (bf16 tiles: 18 KiB of LDS per wave instead of 34, so two 4-wave workgroups share a CU)
template <typename T, int HD, bool TRAIN, int WAVES, bool ADROP = false>
__global__ __launch_bounds__(WAVES * 64, (sizeof(T) == 2 && HD <= 64) ? 2 : 1) void win_attn_bwd_k(const T* __restrict__ qkv,
                                                                const T* __restrict__ dO,
                                                                T* __restrict__ dqkv,
                                                                const uint32_t* __restrict__ maskbits,
                                                                const float* __restrict__ thr_p,
                                                                WinGeom g, int n_units, AttnDrop ad) {
    if constexpr (ADROP) ad.seed += seed_base_of(ad.base);
    using TL = tile_of<T>;                                   // see win_attn_fwd_k
    using E = typename TL::E;
    constexpr int LDW = HD + TL::PAD;
    constexpr int NT = HD / 32;
    constexpr int EPV = io<T>::EPV;
    constexpr int CPR = HD / EPV;
    constexpr int RPI = 64 / CPR;
    constexpr int NLD = 32 / RPI;
    constexpr int TW = 34;                                   // transpose scratch row stride
    constexpr int SCR = 2 * 32 * TW;                         // P^T and dS^T scratch, in tile elements (bf16 tiles: P and dS
                                                             // are rounded here instead of at the MFMA operand -- the same values)
    constexpr int EXTRA = (32 * LDW >= SCR) ? 0 : SCR;       // scratch aliases the V tile when it fits
    constexpr int PER_WAVE = 4 * 32 * LDW + EXTRA;
    __shared__ __attribute__((aligned(16))) E smem[WAVES * PER_WAVE];

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;   // uniform: unit decoding in SGPRs
    const int lq = lane & 31, hh = lane >> 5;
    E* Qs = smem + wave * PER_WAVE;
    E* Ks = Qs + 32 * LDW;
    E* Gs = Ks + 32 * LDW;
    E* Vs = Gs + 32 * LDW;
    E* Pt = EXTRA ? Vs + 32 * LDW : Vs;                       // [q][key], stride TW
    E* Dt = Pt + 32 * TW;
    const int crow_l = lane / CPR, ccol = (lane % CPR) * EPV;
    const int64_t row3d = 3 * (int64_t)g.d;
    const float thr = TRAIN ? *thr_p : 0.f;

    const int nwaves = gridDim.x * WAVES;
    int u = blockIdx.x * WAVES + wave;
    if (u >= n_units) return;

    u32x4 qr[NLD], kr[NLD], vr[NLD], gr[NLD];
    auto issue = [&](const Unit& un) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int64_t tk = tok_of(un, i * RPI + crow_l);
            const T* p = qkv + tk * row3d + un.head * HD + ccol;
            qr[i] = load16_s<T>(p);
            kr[i] = load16_s<T>(p + g.d);
            vr[i] = load16_s<T>(p + 2 * g.d);
            gr[i] = load16_s<T>(dO + tk * (int64_t)g.d + un.head * HD + ccol);
        }
    };
    Unit cur = decode_unit(g, u);
    issue(cur);

    for (; u < n_units; u += nwaves) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int off = (i * RPI + crow_l) * LDW + ccol;
            raw_to_lds(Qs + off, qr[i], qk_scale<HD>(), T());
            raw_to_lds(Ks + off, kr[i], 1.0f, T());
            raw_to_lds(Vs + off, vr[i], 1.0f, T());
            raw_to_lds(Gs + off, gr[i], 1.0f, T());
        }
        const uint32_t mbits = maskbits[cur.mrow + lq];
        const int un = u + nwaves;
        Unit nxt = cur;
        if (un < n_units) { nxt = decode_unit(g, un); issue(nxt); }
        lds_fence();

        // recompute P (lane = query, regs = keys)
        float s[16], p[16], ds[16];
        {
            f32x16 st = tile_xyT<HD, LDW>(Ks, Qs, lq, hh);
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = TL::QSCALED ? st[r] : st[r] * qk_scale<HD>();
        }
        const uint32_t nz = masked_softmax<TRAIN>(s, p, mbits, hh, thr);
        // dP^T[key][q] = V dO^T
        {
            f32x16 dp = tile_xyT<HD, LDW>(Vs, Gs, lq, hh);  // with attention dropout this is dA, A = D o P: dP = D o dA
            float keep[ADROP ? 16 : 1];
            if constexpr (ADROP) {
                attn_keep(keep, ad, u, lq, hh);
#pragma unroll
                for (int r = 0; r < 16; ++r) dp[r] *= keep[r];
            }
            float delta = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) delta += p[r] * dp[r];
            delta += partner(delta);
#pragma unroll
            for (int r = 0; r < 16; ++r) ds[r] = ((nz >> r) & 1u) ? p[r] * (dp[r] - delta) : 0.f;
            if constexpr (ADROP) {                            // dV = A^T dO
#pragma unroll
                for (int r = 0; r < 16; ++r) p[r] *= keep[r];
            }
        }
        lds_fence();                                          // V tile is dead from here on
        // transpose P and dS through LDS: write [q][key], later read [.][key = lane]
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) put4(Pt + lq * TW + 8 * gq + 4 * hh, p + 4 * gq), put4(Dt + lq * TW + 8 * gq + 4 * hh, ds + 4 * gq);
        lds_fence();

        f32x16 acc[NT];
        T* gq_base = dqkv + cur.head * HD + lq * NT;
        auto store_acc = [&](T* base, float mul) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float ov[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) ov[nt] = acc[nt][r] * mul;
                store_nt_s<T, NT>(base + tok_of(cur, crow(r, hh)) * row3d, ov);
            }
        };
        // dQ = scale * dS K        (A = dS in registers: lane = q)
        tile_ay<HD, LDW>(ds, Ks, lq, hh, acc);
        store_acc(gq_base, qk_scale<HD>());
        // dK = dS^T (scale*Q)      (A = dS^T from scratch: lane = key; the fp32 Qs already holds scale*Q)
        float a[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) a[r] = (float)Dt[crow(r, hh) * TW + lq];
        tile_ay<HD, LDW>(a, Qs, lq, hh, acc);
        store_acc(gq_base + g.d, TL::QSCALED ? 1.0f : qk_scale<HD>());
        // dV = P^T dO
#pragma unroll
        for (int r = 0; r < 16; ++r) a[r] = (float)Pt[crow(r, hh) * TW + lq];
        tile_ay<HD, LDW>(a, Gs, lq, hh, acc);
        store_acc(gq_base + 2 * g.d, 1.0f);
        lds_fence();
        cur = nxt;
    }
}

// =============================================================== backward, head_dim 128: two waves per unit
// With hd = 128 one wave needs 4 x 32 x 132 floats = 66 KiB of LDS for its Q, K, dO, V tiles: two waves per CU, i.e.
// two of the four matrix pipes idle and nothing to hide a wave's load / softmax / store phases behind (measured
// 0.52 of the HBM roof in fp32 and 0.26 with bf16 storage, where the 320 MFMAs of a unit are the whole cost).
// Here the two waves of a 128-thread workgroup SHARE a unit by head-dim halves: wave w owns columns
// [64 w, 64 w + 64) of Q, K, V, dO (34 KiB of tiles, as in the hd = 64 kernel -> four waves per CU).  The two
// products that contract over head_dim, S = (scale Q) K^T and dP = dO V^T, are formed as per-half partial sums and
// exchanged through an 8 KiB LDS mailbox (a + b on one side, b + a on the other: bit-identical, so both waves
// run the same softmax); dQ, dK, dV split by columns and need no reduction.  160 MFMAs per wave and unit.
template <typename T, bool TRAIN, bool ADROP = false>
__global__ __launch_bounds__(128, sizeof(T) == 2 ? 3 : 2) void win_attn_bwd_split_k(const T* __restrict__ qkv, const T* __restrict__ dO,
                                                               T* __restrict__ dqkv,
                                                               const uint32_t* __restrict__ maskbits,
                                                               const float* __restrict__ thr_p, WinGeom g,
                                                               int n_units, AttnDrop ad) {
    if constexpr (ADROP) ad.seed += seed_base_of(ad.base);
    using TL = tile_of<T>;                                     // see win_attn_fwd_k
    using E = typename TL::E;
    constexpr int HD = 128, HW = 64, LDW = HW + TL::PAD, NT = HW / 32;
    constexpr int EPV = io<T>::EPV;
    constexpr int CPR = HW / EPV, RPI = 64 / CPR, NLD = 32 / RPI;
    constexpr int TW = 34;
    static_assert(2 * 32 * TW <= 32 * LDW, "transpose scratch must fit the dead V tile");
    constexpr int PER_WAVE = 4 * 32 * LDW;
    constexpr int MBOX = 1024 * (int)(sizeof(float) / sizeof(E));         // 4 KiB mailbox per wave, in tile elements
    __shared__ __attribute__((aligned(16))) E smem[2 * PER_WAVE + 2 * MBOX];            // fp32: 77 824 B -> two workgroups per CU

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int lq = lane & 31, hh = lane >> 5;
    E* Qs = smem + wave * PER_WAVE;
    E* Ks = Qs + 32 * LDW;
    E* Gs = Ks + 32 * LDW;
    E* Vs = Gs + 32 * LDW;
    E* Pt = Vs;                                                // [q][key], stride TW (V is dead by then)
    E* Dt = Pt + 32 * TW;
    float* mine = reinterpret_cast<float*>(smem + 2 * PER_WAVE + wave * MBOX);   // mailbox: register quad r4 of lane l at [r4][l][4]
    const float* theirs = reinterpret_cast<const float*>(smem + 2 * PER_WAVE + (wave ^ 1) * MBOX);
    const int crow_l = lane / CPR, ccol = (lane % CPR) * EPV;
    const int col0 = wave * HW;
    const int64_t row3d = 3 * (int64_t)g.d;
    const float thr = TRAIN ? *thr_p : 0.f;

    int u = blockIdx.x;
    if (u >= n_units) return;                                  // whole workgroup: both waves share u

    u32x4 qr[NLD], kr[NLD], vr[NLD], gr[NLD];
    auto issue = [&](const Unit& un) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int64_t tk = tok_of(un, i * RPI + crow_l);
            const T* p = qkv + tk * row3d + un.head * HD + col0 + ccol;
            qr[i] = load16_s<T>(p);
            kr[i] = load16_s<T>(p + g.d);
            vr[i] = load16_s<T>(p + 2 * g.d);
            gr[i] = load16_s<T>(dO + tk * (int64_t)g.d + un.head * HD + col0 + ccol);
        }
    };
    // partial 32x32 product of this wave + the other wave's, through the mailbox (callers place the barriers)
    auto post = [&](const f32x16& t) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 v = {t[4 * q], t[4 * q + 1], t[4 * q + 2], t[4 * q + 3]};
            *reinterpret_cast<f32x4*>(mine + q * 256 + lane * 4) = v;
        }
    };
    auto collect = [&](const f32x16& t, float (&out)[16]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(theirs + q * 256 + lane * 4);
            out[4 * q] = t[4 * q] + v.x; out[4 * q + 1] = t[4 * q + 1] + v.y;
            out[4 * q + 2] = t[4 * q + 2] + v.z; out[4 * q + 3] = t[4 * q + 3] + v.w;
        }
    };
    Unit cur = decode_unit(g, u);
    issue(cur);

    for (; u < n_units; u += gridDim.x) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int off = (i * RPI + crow_l) * LDW + ccol;
            raw_to_lds(Qs + off, qr[i], qk_scale<HD>(), T());
            raw_to_lds(Ks + off, kr[i], 1.0f, T());
            raw_to_lds(Vs + off, vr[i], 1.0f, T());
            raw_to_lds(Gs + off, gr[i], 1.0f, T());
        }
        const uint32_t mbits = maskbits[cur.mrow + lq];
        const int un = u + gridDim.x;
        Unit nxt = cur;
        if (un < n_units) { nxt = decode_unit(g, un); issue(nxt); }
        lds_fence();

        // S over the full head_dim = this half + the other half
        float s[16], p[16], ds[16], dp[16];
        {
            const f32x16 st = tile_xyT<HW, LDW>(Ks, Qs, lq, hh);
            post(st);
            __syncthreads();
            collect(st, s);
            if constexpr (!TL::QSCALED) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s[r] *= qk_scale<HD>();
            }
        }
        const uint32_t nz = masked_softmax<TRAIN>(s, p, mbits, hh, thr);
        // dP^T[key][q] = V dO^T, likewise
        {
            const f32x16 dt = tile_xyT<HW, LDW>(Vs, Gs, lq, hh);
            __syncthreads();                                  // the other wave has read S from the mailbox
            post(dt);
            __syncthreads();
            collect(dt, dp);
            float keep[ADROP ? 16 : 1];
            if constexpr (ADROP) {                            // dP = D o dA (see win_attn_bwd_k)
                attn_keep(keep, ad, u, lq, hh);
#pragma unroll
                for (int r = 0; r < 16; ++r) dp[r] *= keep[r];
            }
            float delta = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) delta += p[r] * dp[r];
            delta += partner(delta);
#pragma unroll
            for (int r = 0; r < 16; ++r) ds[r] = ((nz >> r) & 1u) ? p[r] * (dp[r] - delta) : 0.f;
            if constexpr (ADROP) {
#pragma unroll
                for (int r = 0; r < 16; ++r) p[r] *= keep[r];
            }
        }
        lds_fence();                                          // this wave's V tile is dead from here on
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) put4(Pt + lq * TW + 8 * gq + 4 * hh, p + 4 * gq), put4(Dt + lq * TW + 8 * gq + 4 * hh, ds + 4 * gq);
        lds_fence();

        f32x16 acc[NT];
        T* gq_base = dqkv + cur.head * HD + col0 + lq * NT;
        auto store_acc = [&](T* base, float mul) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float ov[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) ov[nt] = acc[nt][r] * mul;
                store_nt_s<T, NT>(base + tok_of(cur, crow(r, hh)) * row3d, ov);
            }
        };
        tile_ay<HW, LDW>(ds, Ks, lq, hh, acc);                 // dQ[:, half] = scale * dS K[:, half]
        store_acc(gq_base, qk_scale<HD>());
        float a[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) a[r] = (float)Dt[crow(r, hh) * TW + lq];
        tile_ay<HW, LDW>(a, Qs, lq, hh, acc);                  // dK[:, half] = dS^T (scale Q)[:, half]
        store_acc(gq_base + g.d, TL::QSCALED ? 1.0f : qk_scale<HD>());
#pragma unroll
        for (int r = 0; r < 16; ++r) a[r] = (float)Pt[crow(r, hh) * TW + lq];
        tile_ay<HW, LDW>(a, Gs, lq, hh, acc);                  // dV[:, half] = P^T dO[:, half]
        store_acc(gq_base + 2 * g.d, 1.0f);
        __syncthreads();                                      // mailbox (dP) read by both before the next unit posts S
        cur = nxt;
    }
}

__global__ void mfma_probe_k(const float* a, const float* b, float* out) {
    const int lane = threadIdx.x;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    // A[i][k] supplied by lane (i = lane&31, k = lane>>5); B[k][j] by lane (j = lane&31, k = lane>>5)
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(lane & 31) * 2 + (lane >> 5)], b[(lane >> 5) * 32 + (lane & 31)], acc, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 16; ++i) out[lane * 16 + i] = acc[i];
}

// register-only MFMA loop: the attainable v_mfma_f32_32x32x2_f32 rate on this part / clock
template <int NACC>
__global__ __launch_bounds__(256) void mfma_peak_k(float* out, int iters, float seed) {
    f32x16 acc[NACC];
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;
    float x = seed + threadIdx.x * 1e-3f, y = seed * 0.5f + threadIdx.x * 2e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[a], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int i = 0; i < 16; ++i) s += acc[a][i];
    if (s == 12345.678f) out[threadIdx.x] = s;                  // keep the chain alive
}

// the same with v_mfma_f32_16x16x4_f32 (4 accumulator registers per tile): 4*NACC independent tiles
typedef float f32x4v __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void mfma_peak16_k(float* out, int iters, float seed) {
    f32x4v acc[4 * NACC];
#pragma unroll
    for (int a = 0; a < 4 * NACC; ++a)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[a][i] = 0.f;
    float x = seed + threadIdx.x * 1e-3f, y = seed * 0.5f + threadIdx.x * 2e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 2; ++r)                              // 2 x (4*NACC) x 2048 flop = 4 x NACC x 4096 flop
#pragma unroll
            for (int a = 0; a < 4 * NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc[a], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < 4 * NACC; ++a)
#pragma unroll
        for (int i = 0; i < 4; ++i) s += acc[a][i];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

// the same with v_mfma_f32_32x32x16_bf16: NACC independent 32x32 tiles, 32768 flop per MFMA
template <int NACC>
__global__ __launch_bounds__(256) void mfma_peak_bf16_k(float* out, int iters, float seed) {
    f32x16 acc[NACC];
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;
    bf16x8 x, y;
#pragma unroll
    for (int i = 0; i < 8; ++i) { x[i] = (bf16_t)(seed + threadIdx.x * 1e-3f + i); y[i] = (bf16_t)(seed * 0.5f + i * 0.25f); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[a], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int i = 0; i < 16; ++i) s += acc[a][i];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

bool geom_ok(int B, int F, int nW, int nH, int hd) {
    return B > 0 && F > 0 && (F % 2) == 0 && nW > 0 && nH > 0 && (hd == 32 || hd == 64 || hd == 128);
}

template <typename T, int HD>
int launch_fwd(const void* qkv, void* o, const uint32_t* mb, const float* thr, WinGeom g, int n_units, AttnDrop ad,
               hipStream_t st) {
    const int blocks = min((n_units + 3) / 4, 256 * 2);         // (three resident workgroups with bf16 tiles: 130 vs 123 us, no gain)
    if (thr && ad.p > 0.f)
        win_attn_fwd_k<T, HD, true, true><<<blocks, 256, 0, st>>>((const T*)qkv, (T*)o, mb, thr, g, n_units, ad);
    else if (thr)
        win_attn_fwd_k<T, HD, true><<<blocks, 256, 0, st>>>((const T*)qkv, (T*)o, mb, thr, g, n_units, ad);
    else
        win_attn_fwd_k<T, HD, false><<<blocks, 256, 0, st>>>((const T*)qkv, (T*)o, mb, thr, g, n_units, ad);
    HWGAT_LAUNCH_CHECK();
}
template <typename T, int HD>
int launch_bwd(const void* qkv, const void* dO, void* dqkv, const uint32_t* mb, const float* thr,
               WinGeom g, int n_units, AttnDrop ad, hipStream_t st) {
#define BWD_ARGS (const T*)qkv, (const T*)dO, (T*)dqkv, mb, thr, g, n_units, ad
    if constexpr (HD == 128) {                               // two waves per unit, four waves per CU (see win_attn_bwd_split_k)
        static const bool whole = [] { const char* e = lab_env("HWGAT_ATTN_SPLIT"); return e && e[0] == '0'; }();
        if (!whole) {
            const int blocks = min(n_units, 256 * (sizeof(T) == 2 ? 3 : 2));   // bf16 tiles: 44 KiB per workgroup
            if (thr && ad.p > 0.f) win_attn_bwd_split_k<T, true, true><<<blocks, 128, 0, st>>>(BWD_ARGS);
            else if (thr) win_attn_bwd_split_k<T, true><<<blocks, 128, 0, st>>>(BWD_ARGS);
            else win_attn_bwd_split_k<T, false><<<blocks, 128, 0, st>>>(BWD_ARGS);
            HWGAT_LAUNCH_CHECK();
        }
    }
    constexpr int WAVES = HD <= 64 ? 4 : 2;                  // 4 x 34 KiB or 2 x 66 KiB of LDS per CU
    const int blocks = min((n_units + WAVES - 1) / WAVES, (sizeof(T) == 2 && HD <= 64) ? 512 : 256);
    if (thr && ad.p > 0.f) win_attn_bwd_k<T, HD, true, WAVES, true><<<blocks, WAVES * 64, 0, st>>>(BWD_ARGS);
    else if (thr) win_attn_bwd_k<T, HD, true, WAVES><<<blocks, WAVES * 64, 0, st>>>(BWD_ARGS);
    else win_attn_bwd_k<T, HD, false, WAVES><<<blocks, WAVES * 64, 0, st>>>(BWD_ARGS);
#undef BWD_ARGS
    HWGAT_LAUNCH_CHECK();
}

// attention dropout only exists in train mode (thr given); p in [0, 1)
bool drop_ok(const float* thr, float p) { return p >= 0.f && p < 1.f && (p == 0.f || thr); }

}  // namespace

extern "C" int hwgat_debug_mfma32x32x2(const float* a, const float* b, float* out, void* stream) {
    if (!a || !b || !out) return HWGAT_EINVAL;
    mfma_probe_k<<<1, 64, 0, (hipStream_t)stream>>>(a, b, out);
    HWGAT_LAUNCH_CHECK();
}

extern "C" int hwgat_debug_mfma_peak(float* out, int blocks, int iters, int n_acc, void* stream) {
    if (!out || blocks <= 0 || iters <= 0) return HWGAT_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (n_acc == 4) mfma_peak_k<4><<<blocks, 256, 0, st>>>(out, iters, 1.0f);
    else if (n_acc == 16) mfma_peak_k<16><<<blocks, 256, 0, st>>>(out, iters, 1.0f);
    else if (n_acc == -4) mfma_peak16_k<4><<<blocks, 256, 0, st>>>(out, iters, 1.0f);      // 16x16x4 form
    else if (n_acc == 100 + 4) mfma_peak_bf16_k<4><<<blocks, 256, 0, st>>>(out, iters, 1.0f);   // bf16 32x32x16, 4 x NACC x 32768 flop per iteration per wave
    else if (n_acc == 100 + 16) mfma_peak_bf16_k<16><<<blocks, 256, 0, st>>>(out, iters, 1.0f);
    else return HWGAT_ESHAPE;
    HWGAT_LAUNCH_CHECK();
}

extern "C" int hwgat_win_attn_fwd_drop(const void* qkv, void* o, const uint32_t* maskbits, const float* thr,
                                       int B, int F, int nW, int nH, int hd, int shifted, int dtype,
                                       uint32_t drop_seed, float drop_p, const uint32_t* seed_base, void* stream) {
    if (!qkv || !o || !maskbits || !drop_ok(thr, drop_p)) return HWGAT_EINVAL;
    if (!geom_ok(B, F, nW, nH, hd)) return HWGAT_ESHAPE;
    WinGeom g{F, nW * 16, nW, nH, F / 2, nH * hd, shifted ? 1 : 0};
    const int64_t units = (int64_t)B * g.f * nW * nH;
    if (units > 0x7fffffff) return HWGAT_ESHAPE;
    hipStream_t st = (hipStream_t)stream;
    const AttnDrop ad = make_drop(drop_seed, drop_p, seed_base);
#define FWD(T)                                                                                      \
    switch (hd) {                                                                                   \
        case 32: return launch_fwd<T, 32>(qkv, o, maskbits, thr, g, (int)units, ad, st);            \
        case 64: return launch_fwd<T, 64>(qkv, o, maskbits, thr, g, (int)units, ad, st);            \
        default: return launch_fwd<T, 128>(qkv, o, maskbits, thr, g, (int)units, ad, st);           \
    }
    if (dtype == HWGAT_F32) { FWD(float) }
    if (dtype == HWGAT_BF16) { FWD(bf16_t) }
#undef FWD
    return HWGAT_EDTYPE;
}

extern "C" int hwgat_win_attn_fwd(const void* qkv, void* o, const uint32_t* maskbits, const float* thr,
                                  int B, int F, int nW, int nH, int hd, int shifted, int dtype,
                                  void* stream) {
    return hwgat_win_attn_fwd_drop(qkv, o, maskbits, thr, B, F, nW, nH, hd, shifted, dtype, 0u, 0.f, nullptr, stream);
}

extern "C" int hwgat_win_attn_bwd_drop(const void* qkv, const void* dO, void* dqkv, const uint32_t* maskbits,
                                       const float* thr, int B, int F, int nW, int nH, int hd, int shifted,
                                       int dtype, uint32_t drop_seed, float drop_p, const uint32_t* seed_base, void* stream) {
    if (!qkv || !dO || !dqkv || !maskbits || !drop_ok(thr, drop_p)) return HWGAT_EINVAL;
    if (!geom_ok(B, F, nW, nH, hd)) return HWGAT_ESHAPE;
    WinGeom g{F, nW * 16, nW, nH, F / 2, nH * hd, shifted ? 1 : 0};
    const int64_t units = (int64_t)B * g.f * nW * nH;
    if (units > 0x7fffffff) return HWGAT_ESHAPE;
    hipStream_t st = (hipStream_t)stream;
    const AttnDrop ad = make_drop(drop_seed, drop_p, seed_base);
#define BWD(T)                                                                                            \
    switch (hd) {                                                                                         \
        case 32: return launch_bwd<T, 32>(qkv, dO, dqkv, maskbits, thr, g, (int)units, ad, st);           \
        case 64: return launch_bwd<T, 64>(qkv, dO, dqkv, maskbits, thr, g, (int)units, ad, st);           \
        default: return launch_bwd<T, 128>(qkv, dO, dqkv, maskbits, thr, g, (int)units, ad, st);          \
    }
    if (dtype == HWGAT_F32) { BWD(float) }
    if (dtype == HWGAT_BF16) { BWD(bf16_t) }
#undef BWD
    return HWGAT_EDTYPE;
}

extern "C" int hwgat_win_attn_bwd(const void* qkv, const void* dO, void* dqkv, const uint32_t* maskbits,
                                  const float* thr, int B, int F, int nW, int nH, int hd, int shifted,
                                  int dtype, void* stream) {
    return hwgat_win_attn_bwd_drop(qkv, dO, dqkv, maskbits, thr, B, F, nW, nH, hd, shifted, dtype, 0u, 0.f, nullptr, stream);
}
