// HGATE block graph-attention, fp32 storage and fp32 MFMA arithmetic, head_dim 64, on gfx950 (MI355X): forward and backward.
//
// Same contract as blk_attn_fwd_k / blk_attn_bwd_k (blk_attn.hip: MSA.forward's attention core of the reference's
// hwgat/models/HGATE.py:84-108 differentiated, with block_partition / block_reverse / torch.roll, HGATE.py:30-47,184-207,
// as index arithmetic).  The 32x32-tile kernel there stages through registers and fits 4 (backward) to 6 (forward) waves
// on a CU, so nothing hides its loads and its softmax: 0.37 / 0.36 of the HBM roof where the fp32 matrix pipe allows
// about 0.9 / 0.7 (the unit's 64 x 64 x 64 products put the forward at the fp32 MFMA / HBM ridge and the backward above it).
// Here a unit (2 frames x KJ <= 32 joints = 64 token slots, one head) is shared by a workgroup of FOUR waves on
// v_mfma_f32_16x16x4_f32 tiles, the structure of blk_attn_bf16.hip:
//
//   stage   Q, K, V, dO of the unit: 4 images of 64 slot rows x 256 bytes, fetched by LDS-DMA (16 bytes per lane, whole
//           256-byte rows, no registers); pad slots (joint >= KJ) re-read joint KJ-1 and are masked out below.
//   phase A wave w owns QUERY slots 16w .. 16w+15.  S^T = K Q^T and dP^T = V dO^T: both sides as row operands (lane (l, g)
//           reads the 16-byte chunk 4 m + g of its row: one ds_read_b128 feeds four MFMAs; the k index of an MFMA is any
//           four channels as long as both operands agree).  Result lane = query, 4 x 4 registers = the lane's 16 of the 64
//           key slots.  Masks, the "== 0 -> -10000" fill (HGATE.py:104) and the softmax in registers; dS = P (dP - delta)
//           where the logit was kept.  O^T = V^T P^T / dQ^T = K^T dS^T: the A operand of row i = channel 4 i + ct, so ONE
//           ds_read_b128 of chunk l of a key row feeds the four channel tiles, and a lane ends up with 16 consecutive
//           channels of its query = 64 bytes of stores.
//   phase B wave w owns KEY slots 16w .. 16w+15: dK^T = Q^T dS, dV^T = dO^T P, with P and dS exchanged through two
//           [query][key] images that overlay the K and V images once every wave is through phase A.
//
// Every operand read is conflict-free under one XOR swizzle of the sixteen 16-byte chunks of a row, applied on the DMA's
// source side (xrf below).  64 KB of LDS per workgroup in the backward pass (two workgroups = 8 waves per CU), 48 KB in the
// forward pass (three workgroups).  HBM traffic is the algorithmic 4 E s / 7 E s.
#include <stdlib.h>
#include "blk16_common.h"

namespace {
using namespace blk;

typedef __attribute__((address_space(3))) f32x4v lds_f32x4;
typedef __attribute__((address_space(3))) float lds_f32;

constexpr int RBF = 256;                                 // bytes per slot row of an image
constexpr int IMGF = 64 * RBF;                           // one image: 64 slot rows

__device__ __forceinline__ f32x4v mfma4(float a, float b, f32x4v c) {
    // D(16x16) += A(16x4) B(4x16): lane l supplies A[i = l&15][k = l>>4] and B[k = l>>4][j = l&15];
    // register r of lane l is D[i = 4 (l>>4) + r][j = l&15]
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// chunk c (0..15) of row `row` sits in slot c ^ xrf(row).  xrf is a bijection of row & 15 whose values for rows r and r ^ 4
// differ by 12: the ds_read_b128 lane groups ({0-3, 12-15, 20-27}, ...) then never meet on a slot, neither when the 16 lanes
// of a group read one chunk column of 16 rows (row operands) nor when they read 16 chunks of two rows 4 apart (column
// operands); the low three bits are the row's, which keeps the b32 reads and b128 writes of the exchange images apart.
__device__ __forceinline__ int xrf(int row) { return (row & 7) | ((((row >> 2) ^ (row >> 3)) & 1) << 3); }
__device__ __forceinline__ uint32_t chunk_off_f(int row, int c) { return row * RBF + ((c ^ xrf(row)) << 4); }

// row operand: channels 16 m + 4 g .. + 3 of slot row `row`
__device__ __forceinline__ f32x4v row_op_f(const char* img, int row, int m, int gq) {
    return *(const lds_f32x4*)(img + chunk_off_f(row, 4 * m + gq));
}
// column operand: channels 4 l .. 4 l + 3 of slot row `row` (the A operands of the four channel tiles ct = 0..3, where row i
// of tile ct is channel 4 i + ct)
__device__ __forceinline__ f32x4v col_op_f(const char* img, int row, int lr) {
    return *(const lds_f32x4*)(img + chunk_off_f(row, lr));
}

// ---- stage: wave w issues DMA instructions 4w .. 4w+3 (4 slot rows each) of each image: Q, K, V (and dO with NIMG = 4)
template <int NIMG>
__device__ __forceinline__ void stage_unit_f(char* sm, const float* qkv, const float* dO, const BlkGeom& g, const BUnit& un,
                                             int64_t qkv_bytes, int64_t do_bytes, int lane, int w) {
    const int64_t rs = 3 * (int64_t)g.d;
    const uint32_t rs4 = (uint32_t)rs * 4, d4 = (uint32_t)g.d * 4;
    const int64_t t0 = min(un.base[0], un.base[1]);              // (a shifted block that wraps has frame B in front of frame A)
    const float* qb = qkv + t0 * rs + un.head * HD;
    const int span_q = (int)min(qkv_bytes - ((const char*)qb - (const char*)qkv), (int64_t)0x7fffffff);
    const auto rq = __builtin_amdgcn_make_buffer_rsrc((void*)qb, 0, span_q, 0x00020000);
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const int ins = 4 * w + jj;                              // slot rows 4 ins .. 4 ins + 3
        const int row = 4 * ins + (lane >> 4), cp = lane & 15;
        const int joint = min(row & 31, g.KJ - 1);               // pad slots re-read the last joint (finite data, masked later)
        const uint32_t frame_rel = (uint32_t)((ins >> 3 ? un.base[1] : un.base[0]) - t0);
        const uint32_t src = (uint32_t)(cp ^ xrf(row)) << 4;
        const int vq = (int)((frame_rel + joint) * rs4 + src);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rq, (lds_void)(sm + ins * 1024), 16, vq, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rq, (lds_void)(sm + IMGF + ins * 1024), 16, vq, (int)d4, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rq, (lds_void)(sm + 2 * IMGF + ins * 1024), 16, vq, (int)(2 * d4), 0, 0);
        if constexpr (NIMG == 4) {
            const float* gb = dO + t0 * (int64_t)g.d + un.head * HD;
            const int span_g = (int)min(do_bytes - ((const char*)gb - (const char*)dO), (int64_t)0x7fffffff);
            const auto rg = __builtin_amdgcn_make_buffer_rsrc((void*)gb, 0, span_g, 0x00020000);
            const int vg = (int)((frame_rel + joint) * d4 + src);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rg, (lds_void)(sm + 3 * IMGF + ins * 1024), 16, vg, 0, 0, 0);
        }
    }
}

// X^T (+)= sum over the 64 key (query) slots of  img[slot][channel]^T  b[slot]:  acc[ct][r] = channel 16 g + 4 r + ct of the
// lane's query (key), where b[kt][r] is the lane's value for slot 16 kt + 4 g + r (the result layout of phase A)
__device__ __forceinline__ void colT_times_regs(const char* img, const f32x4v (&b)[4], int lr, int gq, f32x4v (&acc)[4]) {
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const f32x4v a = col_op_f(img, 16 * kt + 4 * gq + r, lr);
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) acc[ct] = mfma4(a[ct], b[kt][r], acc[ct]);
        }
}
// the lane's 16 consecutive channels 16 g .. 16 g + 15 (scaled) to p[0..15]
__device__ __forceinline__ void store16(float* p, const f32x4v (&acc)[4], float scale) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
        *reinterpret_cast<f32x4v*>(p + 4 * j) = f32x4v{acc[0][j], acc[1][j], acc[2][j], acc[3][j]} * scale;
}

// =============================================================== forward
template <bool ADROP>
__global__ __launch_bounds__(256, 3) void blk_fwd_f32_k(const float* __restrict__ qkv, float* __restrict__ o,
                                                        const uint32_t* __restrict__ maskbits, BlkGeom g, int64_t qkv_bytes,
                                                        AttnDrop ad) {
    if constexpr (ADROP) ad.seed += seed_base_of(ad.base);
    __shared__ __attribute__((aligned(1024))) char sm[3 * IMGF];  // Q | K | V
    const char* Qt = sm;
    const char* Kt = sm + IMGF;
    const char* Vt = sm + 2 * IMGF;
    const int lane = threadIdx.x & 63, lr = lane & 15, gq = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const BUnit un = decode_bunit(g, blockIdx.x);
    stage_unit_f<3>(sm, qkv, nullptr, g, un, qkv_bytes, 0, lane, w);
    const int slot = 16 * w + lr;
    const bool real = (slot & 31) < g.KJ;
    const int64_t tok = (w >> 1 ? un.base[1] : un.base[0]) + min(slot & 31, g.KJ - 1);
    const uint32_t mb0 = maskbits[(un.mrow + slot) * 2], mb1 = maskbits[(un.mrow + slot) * 2 + 1];
    wait_vm0();
    wg_barrier();

    f32x4v s[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) s[kt] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const f32x4v q = row_op_f(Qt, slot, m, gq);
        f32x4v k[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) k[kt] = row_op_f(Kt, 16 * kt + lr, m, gq);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) s[kt] = mfma4(k[kt][e], q[e], s[kt]);      // S[q][key 16 kt + 4g + r]
    }
    uint32_t nz;
    const float inv = 1.0f / masked_exp64(s, mb0, mb1, gq, g.KJ, nz);
    if constexpr (ADROP) {                                       // HGATE.py:106 (on the numerators: 1 / row sum is applied to O)
        f32x4v keep[4];
        blk_keep16(keep, ad, blockIdx.x, slot, gq, g.KJ);
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) s[kt] *= keep[kt];
    }
    f32x4v acc[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) acc[ct] = f32x4v{0.f, 0.f, 0.f, 0.f};
    colT_times_regs(Vt, s, lr, gq, acc);                        // (products outside any lane-dependent branch: MFMAs want all lanes)
    if (real) store16(o + tok * (int64_t)g.d + un.head * HD + 16 * gq, acc, inv);
}

// =============================================================== backward
template <bool ADROP>
__global__ __launch_bounds__(256, 2) void blk_bwd_f32_k(const float* __restrict__ qkv, const float* __restrict__ dO,
                                                        float* __restrict__ dqkv, const uint32_t* __restrict__ maskbits,
                                                        BlkGeom g, int64_t qkv_bytes, int64_t do_bytes, AttnDrop ad) {
    if constexpr (ADROP) ad.seed += seed_base_of(ad.base);
    __shared__ __attribute__((aligned(1024))) char sm[4 * IMGF];  // Q | K (then dS) | V (then P) | dO
    char* Qt = sm;
    char* Kt = sm + IMGF;
    char* Vt = sm + 2 * IMGF;
    char* Gt = sm + 3 * IMGF;
    char* Dt = Kt;                                               // [query][key] images of phase B
    char* Pt = Vt;
    const int lane = threadIdx.x & 63, lr = lane & 15, gq = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const BUnit un = decode_bunit(g, blockIdx.x);
    const int64_t rs = 3 * (int64_t)g.d;                         // qkv row stride (elements)

    stage_unit_f<4>(sm, qkv, dO, g, un, qkv_bytes, do_bytes, lane, w);
    // this lane's query slot in phase A / key slot in phase B, and the mask words of the query
    const int slot = 16 * w + lr;
    const bool real = (slot & 31) < g.KJ;
    const int64_t tok = (w >> 1 ? un.base[1] : un.base[0]) + min(slot & 31, g.KJ - 1);
    const uint32_t mb0 = maskbits[(un.mrow + slot) * 2], mb1 = maskbits[(un.mrow + slot) * 2 + 1];
    wait_vm0();
    wg_barrier();

    // ================================================= phase A: query slot `slot`
    f32x4v s[4], dp[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) { s[kt] = f32x4v{0.f, 0.f, 0.f, 0.f}; dp[kt] = f32x4v{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const f32x4v q = row_op_f(Qt, slot, m, gq), gd = row_op_f(Gt, slot, m, gq);
        f32x4v k[4], v[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) { k[kt] = row_op_f(Kt, 16 * kt + lr, m, gq); v[kt] = row_op_f(Vt, 16 * kt + lr, m, gq); }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                s[kt] = mfma4(k[kt][e], q[e], s[kt]);                            // S[q][key 16 kt + 4g + r]
                dp[kt] = mfma4(v[kt][e], gd[e], dp[kt]);                         // dP[q][key]
            }
    }
    uint32_t nz;
    const float sum = masked_exp64(s, mb0, mb1, gq, g.KJ, nz);
    const float inv = real ? 1.0f / sum : 0.f;                                   // pad query slots: P = dS = 0
    // attention dropout: A = D o P went into O = A V, so dP = D o dA (dA = dO V^T, in `dp`) and dV = A^T dO; mask recomputed
    f32x4v keep[ADROP ? 4 : 1];
    if constexpr (ADROP) blk_keep16(keep, ad, blockIdx.x, slot, gq, g.KJ);
    float delta = 0.f;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
        s[kt] *= inv;                                                            // P (HGATE.py:105)
        if constexpr (ADROP) dp[kt] *= keep[kt];
#pragma unroll
        for (int r = 0; r < 4; ++r) delta = __builtin_fmaf(s[kt][r], dp[kt][r], delta);
    }
    delta = xg_sum(delta);
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) dp[kt][r] = ((nz >> (4 * kt + r)) & 1u) ? s[kt][r] * (dp[kt][r] - delta) : 0.f;   // dS
        if constexpr (ADROP) s[kt] *= keep[kt];                                  // the P image feeds dV only: A = D o P
    }
    // dQ^T[c][q] = scale * sum_key K[key][c] dS[q][key]
    {
        f32x4v acc[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[ct] = f32x4v{0.f, 0.f, 0.f, 0.f};
        colT_times_regs(Kt, dp, lr, gq, acc);
        if (real) store16(dqkv + tok * rs + un.head * HD + 16 * gq, acc, SCALE);
    }
    wait_lds_barrier();                                          // every wave is through with the K and V images
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {                             // [query][key] images: keys 16 kt + 4g .. + 3 = chunk 4 kt + g
        *(lds_f32x4*)(Pt + chunk_off_f(slot, 4 * kt + gq)) = s[kt];
        *(lds_f32x4*)(Dt + chunk_off_f(slot, 4 * kt + gq)) = dp[kt];
    }
    wait_lds_barrier();                                          // P, dS of all four query groups are in the images

    // ================================================= phase B: key slot `slot`
    {
        f32x4v dk[4], dv[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) { dk[ct] = f32x4v{0.f, 0.f, 0.f, 0.f}; dv[ct] = f32x4v{0.f, 0.f, 0.f, 0.f}; }
        const int kc = 4 * w + (lr >> 2), kb = (lr & 3) * 4;    // this lane's key column of the [query][key] images
#pragma unroll
        for (int qt = 0; qt < 4; ++qt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qrow = 16 * qt + 4 * gq + r;
                const f32x4v qa = col_op_f(Qt, qrow, lr), ga = col_op_f(Gt, qrow, lr);
                const float dsb = *(const lds_f32*)(Dt + chunk_off_f(qrow, kc) + kb);
                const float pb = *(const lds_f32*)(Pt + chunk_off_f(qrow, kc) + kb);
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) {
                    dk[ct] = mfma4(qa[ct], dsb, dk[ct]);                         // dK[key][c] += sum_q dS[q][key] Q[q][c]
                    dv[ct] = mfma4(ga[ct], pb, dv[ct]);                          // dV[key][c] += sum_q P[q][key] dO[q][c]
                }
            }
        if (real) {
            float* row = dqkv + tok * rs + un.head * HD + 16 * gq;
            store16(row + g.d, dk, SCALE);
            store16(row + 2 * g.d, dv, 1.0f);
        }
    }
}

}  // namespace

int hwgat_launch_blk_fwd_f32(const void* qkv, void* o, const uint32_t* maskbits, int B, int F, int KJ, int nH, int shifted,
                             uint32_t drop_seed, float drop_p, const uint32_t* seed_base, hipStream_t st) {
    const AttnDrop ad = make_drop(drop_seed, drop_p, seed_base);
    BlkGeom g{F, KJ, nH, F / 2, nH * HD, shifted ? 1 : 0};
    const int64_t units = (int64_t)B * g.f * nH;
    const int64_t clip_bytes = (int64_t)F * KJ * 3 * g.d * 4;
    if (units > 0x7fffffff || clip_bytes > 0x7fffffff) return HWGAT_ESHAPE;
    if (ad.p > 0.f) blk_fwd_f32_k<true><<<(int)units, 256, 0, st>>>((const float*)qkv, (float*)o, maskbits, g, clip_bytes * B, ad);
    else blk_fwd_f32_k<false><<<(int)units, 256, 0, st>>>((const float*)qkv, (float*)o, maskbits, g, clip_bytes * B, ad);
    HWGAT_LAUNCH_CHECK();
}

int hwgat_launch_blk_bwd_f32(const void* qkv, const void* dO, void* dqkv, const uint32_t* maskbits, int B, int F, int KJ,
                             int nH, int shifted, uint32_t drop_seed, float drop_p, const uint32_t* seed_base, hipStream_t st) {
    const AttnDrop ad = make_drop(drop_seed, drop_p, seed_base);
    BlkGeom g{F, KJ, nH, F / 2, nH * HD, shifted ? 1 : 0};
    const int64_t units = (int64_t)B * g.f * nH;
    const int64_t clip_bytes = (int64_t)F * KJ * 3 * g.d * 4;
    if (units > 0x7fffffff || clip_bytes > 0x7fffffff) return HWGAT_ESHAPE;
    if (ad.p > 0.f)
        blk_bwd_f32_k<true><<<(int)units, 256, 0, st>>>((const float*)qkv, (const float*)dO, (float*)dqkv, maskbits, g,
                                                        clip_bytes * B, clip_bytes * B / 3, ad);
    else
        blk_bwd_f32_k<false><<<(int)units, 256, 0, st>>>((const float*)qkv, (const float*)dO, (float*)dqkv, maskbits, g,
                                                         clip_bytes * B, clip_bytes * B / 3, ad);
    HWGAT_LAUNCH_CHECK();
}
