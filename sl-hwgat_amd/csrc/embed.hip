// Part gather + Fourier feature mapping + positional encoding, and TemporalMerging,
// for HWGAT on gfx950.
//
// embed: replaces WindowCreate's host-side numpy gather (reference
// hwgat/dataTransform.py:445-455), the Fourier mapping (hwgat/models/HWGATE.py:343-345)
// and PositionalEncoding's add (HWGATE.py:25-27) with one streaming kernel: each
// lane computes one projection p_m, one sincosf, and writes the sin half and the cos
// half of the token row (two fully coalesced row segments per wave).  Arguments reach
// hundreds of radians, so the accurate (range-reducing) sincosf is used on purpose.
//
// merge: TemporalMerging (HWGATE.py:55-63) as a 16-byte-vector permutation copy.
#include "common.h"
#include "fused_ops.h"

namespace {

template <typename T, int C>
__global__ __launch_bounds__(256) void embed_fwd_k(const float* __restrict__ x, const int32_t* __restrict__ idx,
                                                   const float* __restrict__ bmat, const float* __restrict__ pe,
                                                   T* __restrict__ out, int64_t n_tok, int T_, int J, int K,
                                                   int d0, uint32_t seed, float drop_p,
                                                   const uint32_t* __restrict__ sbase) {
    seed += seed_base_of(sbase);
    const int half = d0 >> 1;
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwave = (int64_t)gridDim.x * 4;
    const int64_t n_bt = n_tok / K;                           // (clip, frame) rows of K tokens
    constexpr float TWO_PI = 6.283185307179586f;            // float(2.*torch.pi), HWGATE.py:343
    const uint32_t dth = drop_thresh(drop_p);               // PositionalEncoding's Dropout (HWGATE.py:28)
    const float dsc = 1.0f / (1.0f - drop_p);
    for (int m0 = 0; m0 < half; m0 += 64) {
        const int m = m0 + lane;
        const bool act = m < half;
        float bw[C];
#pragma unroll
        for (int c = 0; c < C; ++c) bw[c] = act ? bmat[m * C + c] : 0.f;
        // A wave walks whole (clip, frame) rows of K tokens: no division per token (the 64-bit tok / K, tok % K, bt % T of a
        // flat token loop were most of this kernel's time: ~700 vector-ALU cycles per token, 200 us for 168 / 335 MB), four
        // tokens per trip so that their gather-index and coordinate loads are in flight together.
        for (int64_t bt = wave; bt < n_bt; bt += nwave) {
            const int t = (int)(bt % T_);
            const float* xrow = x + bt * J * C;
            T* orow = out + bt * K * (int64_t)d0;
            float pes = 0.f, pec = 0.f;
            if (pe) { pes = pe[t * d0 + m]; pec = pe[t * d0 + half + m]; }
#pragma unroll 4
            for (int k = 0; k < K; ++k) {
                const int j = idx ? idx[k] : k;
                const float* xp = xrow + j * C;
                float sn, cs;
                if constexpr (sizeof(T) == 2) {
                    // bf16 output (8 significand bits): the hardware sine / cosine of the angle in REVOLUTIONS, v_sin_f32 /
                    // v_cos_f32 on fract(x . b) -- absolute error ~1e-6 plus ~2e-5 rad from summing before the 2 pi,
                    // against 4e-3 of output rounding
                    float rev = xp[0] * bw[0];
#pragma unroll
                    for (int c = 1; c < C; ++c) rev = fmaf(xp[c], bw[c], rev);
                    rev = __builtin_amdgcn_fractf(rev);
                    sn = __builtin_amdgcn_sinf(rev);
                    cs = __builtin_amdgcn_cosf(rev);
                } else {
                    float p = (TWO_PI * xp[0]) * bw[0];
#pragma unroll
                    for (int c = 1; c < C; ++c) p = fmaf(TWO_PI * xp[c], bw[c], p);
                    sincosf(p, &sn, &cs);
                }
                sn += pes;
                cs += pec;
                if (dth) {
                    const uint64_t e0 = (uint64_t)(bt * K + k) * d0;
                    sn *= drop_keep(seed, e0 + m, dth, dsc);
                    cs *= drop_keep(seed, e0 + half + m, dth, dsc);
                }
                if (act) {
                    io<T>::st(orow + k * d0 + m, sn);
                    io<T>::st(orow + k * d0 + half + m, cs);
                }
            }
        }
    }
}

// 16-byte chunks; forward: out[b,fi,k,tp*d+c] = in[b,2fi+tp,k,c]
// out_m (inverse only): a second copy multiplied by the dropout mask of (seed, un-merged element index) -- the gradient of
// a stage's last block arrives merged; its fc2-dropout backward needs the masked un-merged gradient (see ln_bwd_k<MASK>)
template <typename T>
__global__ __launch_bounds__(256) void merge_k(const T* __restrict__ in, T* __restrict__ out, int64_t n_chunks,
                                               int F, int K, int d, int inverse, T* __restrict__ out_m = nullptr,
                                               uint32_t mseed = 0, float mp = 0.f,
                                               const uint32_t* __restrict__ sbase = nullptr) {
    mseed += seed_base_of(sbase);
    constexpr int EPV = io<T>::EPV;
    const int cpr = d / EPV;                                   // chunks per input row
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_chunks; i += (int64_t)gridDim.x * 256) {
        // i indexes the un-merged tensor (b, F, K, cpr)
        const int c = (int)(i % cpr);
        int64_t r = i / cpr;
        const int k = (int)(r % K); r /= K;
        const int fr = (int)(r % F);
        const int64_t b = r / F;
        const int64_t j = (((b * (F / 2) + (fr >> 1)) * K + k) * 2 + (fr & 1)) * cpr + c;
        const u32x4* src = reinterpret_cast<const u32x4*>(in);
        u32x4* dst = reinterpret_cast<u32x4*>(out);
        if (inverse) {
            const u32x4 v = src[j];
            dst[i] = v;
            if (out_m != nullptr) {
                const uint32_t th = drop_thresh(mp);
                const float sc = 1.0f / (1.0f - mp);
                u32x4 w = v;
                if constexpr (EPV == 4) {
                    const f32x4 k = drop_keep4(mseed, (uint64_t)i * 4, th, sc);
                    w.x = __float_as_uint(__uint_as_float(v.x) * k.x); w.y = __float_as_uint(__uint_as_float(v.y) * k.y);
                    w.z = __float_as_uint(__uint_as_float(v.z) * k.z); w.w = __float_as_uint(__uint_as_float(v.w) * k.w);
                } else {
                    const f32x4 k0 = drop_keep4(mseed, (uint64_t)i * 8, th, sc), k1 = drop_keep4(mseed, (uint64_t)i * 8 + 4, th, sc);
                    const uint32_t r4[4] = {v.x, v.y, v.z, v.w};
                    const float kk[8] = {k0.x, k0.y, k0.z, k0.w, k1.x, k1.y, k1.z, k1.w};
                    uint32_t o4[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const bf16_t lo = (bf16_t)(__uint_as_float(r4[q] << 16) * kk[2 * q]);
                        const bf16_t hi = (bf16_t)(__uint_as_float(r4[q] & 0xffff0000u) * kk[2 * q + 1]);
                        o4[q] = (uint32_t)__builtin_bit_cast(uint16_t, lo) | ((uint32_t)__builtin_bit_cast(uint16_t, hi) << 16);
                    }
                    w.x = o4[0]; w.y = o4[1]; w.z = o4[2]; w.w = o4[3];
                }
                reinterpret_cast<u32x4*>(out_m)[i] = w;
            }
        } else {
            dst[j] = src[i];
        }
    }
}

}  // namespace

extern "C" int hwgat_embed_fwd(const float* x, const int32_t* idx, const float* bmat, const float* pe, void* out,
                               int B, int T, int J, int K, int C, int d0, int dtype, uint32_t seed, float drop_p,
                               const uint32_t* seed_base, void* stream) {
    if (!x || !bmat || !out || B <= 0 || T <= 0 || J <= 0 || K <= 0 || drop_p < 0.f || drop_p >= 1.f) return HWGAT_EINVAL;
    if (d0 <= 0 || (d0 & 1) || (!idx && J != K) || (C != 2 && C != 3)) return HWGAT_ESHAPE;
    hipStream_t st = (hipStream_t)stream;
    const int64_t n_tok = (int64_t)B * T * K;
    const int64_t n_bt = (int64_t)B * T;                      // one wave per (clip, frame) row of K tokens, grid-stride beyond
    const int grid = (int)(n_bt / 4 < 4096 ? (n_bt + 3) / 4 : 4096);
#define GO(TT, CC) embed_fwd_k<TT, CC><<<grid, 256, 0, st>>>(x, idx, bmat, pe, (TT*)out, n_tok, T, J, K, d0, seed, drop_p, seed_base)
    if (dtype == HWGAT_F32) { if (C == 2) GO(float, 2); else GO(float, 3); }
    else if (dtype == HWGAT_BF16) { if (C == 2) GO(bf16_t, 2); else GO(bf16_t, 3); }
    else return HWGAT_EDTYPE;
#undef GO
    HWGAT_LAUNCH_CHECK();
}

extern "C" int hwgat_merge(const void* in, void* out, int B, int F, int K, int d, int inverse, int dtype,
                           void* stream) {
    if (!in || !out || B <= 0 || F <= 0 || K <= 0 || d <= 0) return HWGAT_EINVAL;
    if (F & 1) return HWGAT_ESHAPE;
    hipStream_t st = (hipStream_t)stream;
    if (dtype != HWGAT_F32 && dtype != HWGAT_BF16) return HWGAT_EDTYPE;
    const int epv = dtype == HWGAT_F32 ? 4 : 8;
    if (d % epv) return HWGAT_ESHAPE;
    const int64_t n = (int64_t)B * F * K * (d / epv);
    const int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    if (dtype == HWGAT_F32) merge_k<float><<<grid, 256, 0, st>>>((const float*)in, (float*)out, n, F, K, d, inverse);
    else merge_k<bf16_t><<<grid, 256, 0, st>>>((const bf16_t*)in, (bf16_t*)out, n, F, K, d, inverse);
    HWGAT_LAUNCH_CHECK();
}

extern "C" int hwgat_unmerge_masked(const void* in, void* out, void* out_masked, int B, int F, int K, int d, int dtype,
                                    uint32_t mask_seed, float mask_p, const uint32_t* seed_base, void* stream) {
    if (!in || !out || !out_masked || B <= 0 || F <= 0 || K <= 0 || d <= 0) return HWGAT_EINVAL;
    if (mask_p <= 0.f || mask_p >= 1.f) return HWGAT_EINVAL;
    if (F & 1) return HWGAT_ESHAPE;
    hipStream_t st = (hipStream_t)stream;
    if (dtype != HWGAT_F32 && dtype != HWGAT_BF16) return HWGAT_EDTYPE;
    const int epv = dtype == HWGAT_F32 ? 4 : 8;
    if (d % epv) return HWGAT_ESHAPE;
    const int64_t n = (int64_t)B * F * K * (d / epv);
    const int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    if (dtype == HWGAT_F32) merge_k<float><<<grid, 256, 0, st>>>((const float*)in, (float*)out, n, F, K, d, 1, (float*)out_masked, mask_seed, mask_p, seed_base);
    else merge_k<bf16_t><<<grid, 256, 0, st>>>((const bf16_t*)in, (bf16_t*)out, n, F, K, d, 1, (bf16_t*)out_masked, mask_seed, mask_p, seed_base);
    HWGAT_LAUNCH_CHECK();
}

// The device-resident dropout seed of a model: state[0] = step counter, state[1] = base seed of the current step (what the
// kernels read through `seed_base`), state[2] = the model's initial seed, state[3] = rank salt.  One thread; as a node of
// a captured train step it makes every replay draw fresh masks without any host-side value in the graph.
namespace {
__global__ void seed_advance_k(uint32_t* __restrict__ state) {
    const uint32_t n = state[0] + 1u;
    state[0] = n;
    state[1] = state[2] * 0x9E3779B1u + n * 0x85EBCA77u + state[3] * 0x27D4EB2Fu;
}
__global__ void seed_set_k(uint32_t* __restrict__ state, uint32_t counter, uint32_t initial, uint32_t salt) {
    state[0] = counter;
    state[1] = initial * 0x9E3779B1u + counter * 0x85EBCA77u + salt * 0x27D4EB2Fu;
    state[2] = initial;
    state[3] = salt;
}
}  // namespace
extern "C" int hwgat_seed_set(uint32_t* state, uint32_t counter, uint32_t initial, uint32_t salt, void* stream) {
    if (!state) return HWGAT_EINVAL;
    seed_set_k<<<1, 1, 0, (hipStream_t)stream>>>(state, counter, initial, salt);
    HWGAT_LAUNCH_CHECK();
}
extern "C" int hwgat_seed_advance(uint32_t* state, void* stream) {
    if (!state) return HWGAT_EINVAL;
    seed_advance_k<<<1, 1, 0, (hipStream_t)stream>>>(state);
    HWGAT_LAUNCH_CHECK();
}

extern "C" int hwgat_abi_version(void) { return HWGAT_ABI_VERSION; }
extern "C" int hwgat_is_lab_build(void) {
#ifdef HWGAT_LAB
    return 1;
#else
    return 0;
#endif
}
