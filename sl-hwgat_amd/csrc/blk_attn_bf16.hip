// HGATE block graph-attention, bf16 storage, head_dim 64, on gfx950 (MI355X): forward and backward.
//
// Same contract as blk_attn_fwd_k / blk_attn_bwd_k (blk_attn.hip: MSA.forward's attention core of the reference's
// hwgat/models/HGATE.py:84-108 differentiated, with block_partition / block_reverse / torch.roll, HGATE.py:30-47,184-207,
// as index arithmetic).  The 32x32-tile kernel there needs 288 registers per wave for a unit (its 2 x 16 scores, their
// probabilities and gradients per lane, 32-row B-side fragments assembled with v_perm) and reaches 0.29 - 0.37 of the HBM
// roof at one or two waves per SIMD.  Here a unit (2 frames x KJ <= 32 joints = 64 token slots, one head) is shared by a
// workgroup of FOUR waves on 16x16x32 bf16 MFMA tiles, the layout of band_attn_bf16.hip:
//
//   stage   Q, K, V, dO of the unit: 4 images of 64 slot rows x 128 bytes, fetched by LDS-DMA (16 bytes per lane, whole
//           128-byte lines, no registers); pad slots (joint >= KJ) re-read joint KJ-1 and are masked out below.
//   phase A wave w owns QUERY slots 16w .. 16w+15.  S^T = K Q^T and dP^T = V dO^T: row operands (ds_read_b128) of both
//           sides, result lane = query, 4 x 4 registers = the lane's 16 of the 64 key slots.  Masks, the "== 0 -> -10000"
//           fill (HGATE.py:104) and the softmax in fp32 registers; dS = P (dP - delta) where the logit was kept.
//           dQ^T = K^T dS^T with K as a COLUMN operand (ds_read_b64_tr_b16 from the same image): the result is four
//           consecutive channels of one query per lane = 8-byte stores.  P and dS go to two [query][key] images.
//   phase B wave w owns KEY slots 16w .. 16w+15: dK^T = Q^T dS, dV^T = dO^T P with Q / dO as column operands and the
//           exchanged images read transposed (lane = key).
//
// Every operand read is conflict-free under one XOR swizzle of the 16-byte chunks of a 128-byte row (chunk c of row r sits
// in slot c ^ rotl1((r >> 1) & 7), applied on the DMA's source side).  ~100 registers per wave, 48 KB of LDS per
// workgroup: three workgroups = 12 waves per CU.
#include <stdlib.h>
#include "blk16_common.h"

namespace {
using namespace blk;

constexpr int IMG = 64 * RB;                             // one image: 64 slot rows x 128 bytes

// ---- stage: wave w issues DMA instructions 2w, 2w+1 (8 slot rows each) of each image: Q, K, V (and dO with NIMG = 4)
template <int NIMG>
__device__ __forceinline__ void stage_unit(char* sm, const bf16_t* qkv, const bf16_t* dO, const BlkGeom& g, const BUnit& un,
                                           int64_t qkv_bytes, int64_t do_bytes, int lane, int w) {
    const int64_t rs = 3 * (int64_t)g.d;
    const uint32_t rs2 = (uint32_t)rs * 2, d2 = (uint32_t)g.d * 2;
    const int64_t t0 = min(un.base[0], un.base[1]);              // (a shifted block that wraps has frame B in front of frame A)
    const bf16_t* qb = qkv + t0 * rs + un.head * HD;
    const int span_q = (int)min(qkv_bytes - ((const char*)qb - (const char*)qkv), (int64_t)0x7fffffff);
    const auto rq = __builtin_amdgcn_make_buffer_rsrc((void*)qb, 0, span_q, 0x00020000);
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
        const int ins = 2 * w + jj;                              // slot rows 8 ins .. 8 ins + 7
        const int row = 8 * ins + (lane >> 3), cp = lane & 7;
        const int joint = min(row & 31, g.KJ - 1);               // pad slots re-read the last joint (finite data, masked later)
        const uint32_t frame_rel = (uint32_t)((ins >> 2 ? un.base[1] : un.base[0]) - t0);
        const uint32_t src = (uint32_t)(cp ^ xr(row)) << 4;
        const int vq = (int)((frame_rel + joint) * rs2 + src);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rq, (lds_void)(sm + ins * 1024), 16, vq, 0, 0, 2);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rq, (lds_void)(sm + IMG + ins * 1024), 16, vq, (int)d2, 0, 2);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rq, (lds_void)(sm + 2 * IMG + ins * 1024), 16, vq, (int)(2 * d2), 0, 2);
        if constexpr (NIMG == 4) {
            const bf16_t* gb = dO + t0 * (int64_t)g.d + un.head * HD;
            const int span_g = (int)min(do_bytes - ((const char*)gb - (const char*)dO), (int64_t)0x7fffffff);
            const auto rg = __builtin_amdgcn_make_buffer_rsrc((void*)gb, 0, span_g, 0x00020000);
            const int vg = (int)((frame_rel + joint) * d2 + src);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rg, (lds_void)(sm + 3 * IMG + ins * 1024), 16, vg, 0, 0, 2);
        }
    }
}

// =============================================================== forward
// wave w owns query slots 16w .. 16w+15: S^T = K Q^T, masks + softmax, O^T = V^T P^T (V as a column operand), O rows scaled
// by 1 / row sum and stored as four consecutive channels per lane.  24 KB of LDS: six workgroups per CU.
template <bool ADROP>
__global__ __launch_bounds__(256, 4) void blk_fwd_b16_k(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ o,
                                                        const uint32_t* __restrict__ maskbits, BlkGeom g, int64_t qkv_bytes,
                                                        AttnDrop ad) {
    if constexpr (ADROP) ad.seed += seed_base_of(ad.base);
    __shared__ __attribute__((aligned(1024))) char sm[3 * IMG];  // Q | K | V
    const char* Qt = sm;
    const char* Kt = sm + IMG;
    const char* Vt = sm + 2 * IMG;
    const int lane = threadIdx.x & 63, lr = lane & 15, gq = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const BUnit un = decode_bunit(g, blockIdx.x);
    stage_unit<3>(sm, qkv, nullptr, g, un, qkv_bytes, 0, lane, w);
    const int slot = 16 * w + lr;
    const bool real = (slot & 31) < g.KJ;
    const int64_t tok = (w >> 1 ? un.base[1] : un.base[0]) + min(slot & 31, g.KJ - 1);
    const uint32_t mb0 = maskbits[(un.mrow + slot) * 2], mb1 = maskbits[(un.mrow + slot) * 2 + 1];
    wait_vm0();
    wg_barrier();

    f32x4v s[4];
    {
        const u32x4v q0 = row_op(Qt, slot, 0, gq), q1 = row_op(Qt, slot, 1, gq);
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            const f32x4v z = {0.f, 0.f, 0.f, 0.f};
            s[kt] = mfma32(row_op(Kt, 16 * kt + lr, 1, gq), q1, mfma32(row_op(Kt, 16 * kt + lr, 0, gq), q0, z));
        }
    }
    uint32_t nz;
    const float inv = __builtin_amdgcn_rcpf(masked_exp64(s, mb0, mb1, gq, g.KJ, nz));
    if constexpr (ADROP) {                                       // HGATE.py:106 (on the numerators: 1 / row sum is applied to O)
        f32x4v keep[4];
        blk_keep16(keep, ad, blockIdx.x, slot, gq, g.KJ);
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) s[kt] *= keep[kt];
    }
    u32x2v pb[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) pb[kt] = to_bf(s[kt]);       // numerators, rounded to bf16 as MFMA operands
    bf16_t* row = o + tok * (int64_t)g.d + un.head * HD + 4 * gq;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {                             // (products outside any lane-dependent branch: transposed reads
        f32x4v acc = {0.f, 0.f, 0.f, 0.f};                       //  and MFMAs want all lanes)
#pragma unroll
        for (int h = 0; h < 2; ++h)
            acc = mfma32(col_op(Vt, h, ct, lr, gq), u32x4v{pb[2 * h].x, pb[2 * h].y, pb[2 * h + 1].x, pb[2 * h + 1].y}, acc);
        if (real) *reinterpret_cast<u32x2v*>(row + 16 * ct) = to_bf(acc * inv);  // lane (q, g), register r -> channel 16 ct + 4g + r
    }
}

// =============================================================== backward
template <bool ADROP>
__global__ __launch_bounds__(256, 3) void blk_bwd_b16_k(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dO,
                                                        bf16_t* __restrict__ dqkv, const uint32_t* __restrict__ maskbits,
                                                        BlkGeom g, int64_t qkv_bytes, int64_t do_bytes, AttnDrop ad) {
    if constexpr (ADROP) ad.seed += seed_base_of(ad.base);
    __shared__ __attribute__((aligned(1024))) char sm[6 * IMG];  // Q | K | V | dO | P | dS
    char* Qt = sm;
    char* Kt = sm + IMG;
    char* Vt = sm + 2 * IMG;
    char* Gt = sm + 3 * IMG;
    char* Pt = sm + 4 * IMG;
    char* Dt = sm + 5 * IMG;
    const int lane = threadIdx.x & 63, lr = lane & 15, gq = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const BUnit un = decode_bunit(g, blockIdx.x);
    const int64_t rs = 3 * (int64_t)g.d;                         // qkv row stride (elements)
    const uint32_t rs2 = (uint32_t)rs * 2, d2 = (uint32_t)g.d * 2;

    stage_unit<4>(sm, qkv, dO, g, un, qkv_bytes, do_bytes, lane, w);
    // this lane's query slot in phase A / key slot in phase B, and the mask words of the query
    const int slot = 16 * w + lr;
    const bool real = (slot & 31) < g.KJ;
    const int64_t tok = (w >> 1 ? un.base[1] : un.base[0]) + min(slot & 31, g.KJ - 1);
    const uint32_t mb0 = maskbits[(un.mrow + slot) * 2], mb1 = maskbits[(un.mrow + slot) * 2 + 1];
    wait_vm0();
    wg_barrier();

    // ================================================= phase A: query slot `slot`
    f32x4v s[4], dp[4];
    {
        const u32x4v q0 = row_op(Qt, slot, 0, gq), q1 = row_op(Qt, slot, 1, gq);
        const u32x4v g0 = row_op(Gt, slot, 0, gq), g1 = row_op(Gt, slot, 1, gq);
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            const f32x4v z = {0.f, 0.f, 0.f, 0.f};
            s[kt] = mfma32(row_op(Kt, 16 * kt + lr, 1, gq), q1, mfma32(row_op(Kt, 16 * kt + lr, 0, gq), q0, z));     // S[q][key 16 kt + 4g + r]
            dp[kt] = mfma32(row_op(Vt, 16 * kt + lr, 1, gq), g1, mfma32(row_op(Vt, 16 * kt + lr, 0, gq), g0, z));    // dP[q][key]
        }
    }
    uint32_t nz;
    const float sum = masked_exp64(s, mb0, mb1, gq, g.KJ, nz);
    const float inv = real ? __builtin_amdgcn_rcpf(sum) : 0.f;          // pad query slots: P = dS = 0
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
    u32x2v pb[4], db[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) dp[kt][r] = ((nz >> (4 * kt + r)) & 1u) ? s[kt][r] * (dp[kt][r] - delta) : 0.f;   // dS
        if constexpr (ADROP) pb[kt] = to_bf(s[kt] * keep[kt]);                   // the P image feeds dV only: A = D o P
        else pb[kt] = to_bf(s[kt]);
        db[kt] = to_bf(dp[kt]);
        // [query][key] images: keys 16 kt + 4g .. + 3 = 8 bytes at byte column 32 kt + 8 g
        const uint32_t off = chunk_off(slot, 2 * kt + (gq >> 1)) + (gq & 1) * 8;
        *(lds_u32x2*)(Pt + off) = pb[kt];
        *(lds_u32x2*)(Dt + off) = db[kt];
    }
    // dQ^T[c][q] = scale * sum_key K[key][c] dS[q][key]: lane (q, g), register r -> channel 16 ct + 4g + r
    {
        f32x4v acc[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            acc[ct] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int h = 0; h < 2; ++h)
                acc[ct] = mfma32(col_op(Kt, h, ct, lr, gq), u32x4v{db[2 * h].x, db[2 * h].y, db[2 * h + 1].x, db[2 * h + 1].y}, acc[ct]);
        }
        if (real) {
            bf16_t* row = dqkv + tok * rs + un.head * HD + 4 * gq;
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) *reinterpret_cast<u32x2v*>(row + 16 * ct) = to_bf(acc[ct] * SCALE);
        }
    }
    wait_lds_barrier();                                          // P, dS of all four query groups are in the images

    // ================================================= phase B: key slot `slot`
    {
        f32x4v dk[4], dv[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) { dk[ct] = f32x4v{0.f, 0.f, 0.f, 0.f}; dv[ct] = f32x4v{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            // B operand: element e of lane (key = 16 w + lr, g) = X[query 32 h + 16 (e >> 2) + 4g + (e & 3)][key]: the column
            // operand pattern on the [query][key] images, "channel chunk" = key group w
            const u32x4v d2 = col_op(Dt, h, w, lr, gq), p2 = col_op(Pt, h, w, lr, gq);
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                dk[ct] = mfma32(col_op(Qt, h, ct, lr, gq), d2, dk[ct]);                  // dK[key][c] += sum_q dS[q][key] Q[q][c]
                dv[ct] = mfma32(col_op(Gt, h, ct, lr, gq), p2, dv[ct]);                  // dV[key][c] += sum_q P[q][key] dO[q][c]
            }
        }
        if (real) {
            bf16_t* row = dqkv + tok * rs + un.head * HD + 4 * gq;               // lane (key, g), register r -> channel 16 ct + 4g + r
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                *reinterpret_cast<u32x2v*>(row + g.d + 16 * ct) = to_bf(dk[ct] * SCALE);
                *reinterpret_cast<u32x2v*>(row + 2 * g.d + 16 * ct) = to_bf(dv[ct]);
            }
        }
    }
}

}  // namespace

int hwgat_launch_blk_fwd_b16(const void* qkv, void* o, const uint32_t* maskbits, int B, int F, int KJ, int nH, int shifted,
                             uint32_t drop_seed, float drop_p, const uint32_t* seed_base, hipStream_t st) {
    const AttnDrop ad = make_drop(drop_seed, drop_p, seed_base);
    BlkGeom g{F, KJ, nH, F / 2, nH * HD, shifted ? 1 : 0};
    const int64_t units = (int64_t)B * g.f * nH;
    const int64_t clip_bytes = (int64_t)F * KJ * 3 * g.d * 2;
    if (units > 0x7fffffff || clip_bytes > 0x7fffffff) return HWGAT_ESHAPE;
    if (ad.p > 0.f) blk_fwd_b16_k<true><<<(int)units, 256, 0, st>>>((const bf16_t*)qkv, (bf16_t*)o, maskbits, g, clip_bytes * B, ad);
    else blk_fwd_b16_k<false><<<(int)units, 256, 0, st>>>((const bf16_t*)qkv, (bf16_t*)o, maskbits, g, clip_bytes * B, ad);
    HWGAT_LAUNCH_CHECK();
}

int hwgat_launch_blk_bwd_b16(const void* qkv, const void* dO, void* dqkv, const uint32_t* maskbits, int B, int F, int KJ,
                             int nH, int shifted, uint32_t drop_seed, float drop_p, const uint32_t* seed_base, hipStream_t st) {
    const AttnDrop ad = make_drop(drop_seed, drop_p, seed_base);
    BlkGeom g{F, KJ, nH, F / 2, nH * HD, shifted ? 1 : 0};
    const int64_t units = (int64_t)B * g.f * nH;
    const int64_t clip_bytes = (int64_t)F * KJ * 3 * g.d * 2;
    if (units > 0x7fffffff || clip_bytes > 0x7fffffff) return HWGAT_ESHAPE;
    if (ad.p > 0.f)
        blk_bwd_b16_k<true><<<(int)units, 256, 0, st>>>((const bf16_t*)qkv, (const bf16_t*)dO, (bf16_t*)dqkv, maskbits, g,
                                                        clip_bytes * B, clip_bytes * B / 3, ad);
    else
        blk_bwd_b16_k<false><<<(int)units, 256, 0, st>>>((const bf16_t*)qkv, (const bf16_t*)dO, (bf16_t*)dqkv, maskbits, g,
                                                         clip_bytes * B, clip_bytes * B / 3, ad);
    HWGAT_LAUNCH_CHECK();
}
