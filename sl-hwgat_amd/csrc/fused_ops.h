// Elementwise helpers shared by the fused linear kernels (fp32 and bf16 families).
#pragma once
#include "common.h"

// PRO_LN_FOLD: LayerNorm of A folded into the weights and the epilogue (hwgat_ln_fold): the loaders see plain A, W is
// W o gamma, and the epilogue turns the accumulator into rstd_m (acc - mean_m s_n) + c_n with s_n = sum_k W'[n,k] and
// c_n = bias_n + sum_k beta_k W[n,k] (passed in the gamma / beta slots).  Same value as PRO_LN up to rounding; the
// per-element normalisation leaves the load path, where it cost the qkv launch 14 % (fp32) to 33 % (bf16).
enum { PRO_NONE = 0, PRO_LN = 1, PRO_DROP = 2, PRO_LN_FOLD = 3 };
// extra work of an NT epilogue (template parameter STAT of the NT kernels)
enum { X_NONE = 0, X_STAT = 1, X_STAT_MERGE = 2, X_LNFOLD = 3 };
// EPI_BIAS_GELU_DROP_G / EPI_MUL_AUX: the training pair of the MLP's first linear.  The forward epilogue has
// Phi(x) and phi(x) in hand anyway, so it stores  gelu'(x) * mask  (C2) instead of the pre-activation, and the backward
// dX launch only multiplies by it -- no erf / exp / mask hash in the backward epilogue, where nothing hides them
// (EPI_GELU_BWD cost its launch +130 us at stage 2, fp32 and bf16 alike).
enum { EPI_BIAS = 0, EPI_BIAS_DROP_RES = 1, EPI_BIAS_GELU_DROP = 2, EPI_GELU_BWD = 3, EPI_NONE = 4,
       EPI_BIAS_GELU_DROP_G = 5, EPI_MUL_AUX = 6 };
constexpr bool epi_has_bias(int e) { return e == EPI_BIAS || e == EPI_BIAS_DROP_RES || e == EPI_BIAS_GELU_DROP || e == EPI_BIAS_GELU_DROP_G; }
constexpr bool epi_reads_extra(int e) { return e == EPI_BIAS_DROP_RES || e == EPI_GELU_BWD || e == EPI_MUL_AUX; }   // res, else aux
constexpr bool epi_drops(int e) { return e == EPI_BIAS_DROP_RES || e == EPI_BIAS_GELU_DROP || e == EPI_BIAS_GELU_DROP_G || e == EPI_GELU_BWD; }

// Dropout mask: a counter-based hash of (seed, element index).  One 32-bit hash serves an
// aligned PAIR of elements (16 bits each): keep iff its 16 bits >= p * 65536 (p is thereby
// quantised to 1/65536); survivors are scaled by 1/(1-p).
// (A mixer built from v_mul_u32_u24 -- 15 instructions instead of 12, no v_mul_lo_u32 -- was measured on the same box:
//  every masked launch got SLOWER, d_h1 1502 -> 1527 us fp32, 484 -> 519 bf16: the instruction count is what costs.)
__device__ __forceinline__ uint32_t mix32(uint32_t seed, uint64_t pair) {
    uint32_t x = ((uint32_t)pair ^ seed) * 0x9E3779B1u + (uint32_t)(pair >> 32) * 0x85EBCA77u;
    x ^= x >> 15;
    x *= 0x2C1B3C6Du;
    x ^= x >> 12;
    x *= 0x297A2D39u;
    return x ^ (x >> 15);
}
__device__ __forceinline__ float drop_keep(uint32_t seed, uint64_t idx, uint32_t thresh, float scale) {
    const uint32_t h = mix32(seed, idx >> 1);
    return ((idx & 1 ? h >> 16 : h & 0xffffu) >= thresh) ? scale : 0.f;
}
// four consecutive elements starting at an index that is a multiple of 4
__device__ __forceinline__ f32x4 drop_keep4(uint32_t seed, uint64_t idx, uint32_t thresh, float scale) {
    const uint32_t h0 = mix32(seed, idx >> 1), h1 = mix32(seed, (idx >> 1) + 1);
    f32x4 k;
    k.x = (h0 & 0xffffu) >= thresh ? scale : 0.f;
    k.y = (h0 >> 16) >= thresh ? scale : 0.f;
    k.z = (h1 & 0xffffu) >= thresh ? scale : 0.f;
    k.w = (h1 >> 16) >= thresh ? scale : 0.f;
    return k;
}
__device__ __forceinline__ uint32_t drop_thresh(float p) {
    return p <= 0.f ? 0u : (uint32_t)fminf(p * 65536.0f + 0.5f, 65535.0f);
}

// Attention dropout (reference HWGATE.py:78,112 / HGATE.py:78,106 / WGATE.py:81,103: nn.Dropout on the softmax output, train
// mode): the mask is the common hash over the element index of the reference's attention tensor, so
// hwgat_dropout_mask_f32(out, numel, seed, p) reproduces it (tests feed it to the oracles).  `base`: see common.h, seeds.
struct AttnDrop {
    uint32_t seed;
    float p;                    // 0: no attention dropout
    const uint32_t* base;       // device word added to `seed` at kernel entry (NULL = 0)
};
inline AttnDrop make_drop(uint32_t seed, float p, const uint32_t* base) {      // p is quantised to 1/65536
    return AttnDrop{seed, p < 0.5f / 65536.0f ? 0.f : p, base};
}

// Exact-erf GELU (nn.GELU default, HWGATE.py:132) and its derivative.  erf is evaluated branch-free
// with Abramowitz-Stegun 7.1.26: erf(z) = 1 - (a1 t + ... + a5 t^5) exp(-z^2), t = 1/(1 + p z), z >= 0,
// |error| <= 1.5e-7 (fp32-rounding level; libm's erff costs ~4x the instructions and made the GELU
// epilogues VALU-bound).  With z = |x|/sqrt(2) the exponential exp(-x^2/2) is shared with the
// Gaussian pdf term of the derivative.
__device__ __forceinline__ void gelu_parts(float x, float& cdf, float& pdf) {
    const float ax = fabsf(x);
    const float e = __expf(-0.5f * x * x);                             // exp(-z^2), z = |x|/sqrt(2)
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.7071067811865476f, ax, 1.0f));
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float erfc_half = 0.5f * poly * t * e;                        // 0.5 * erfc(z)
    cdf = x >= 0.f ? 1.0f - erfc_half : erfc_half;                      // Phi(x) = 0.5 (1 + erf(x/sqrt2))
    pdf = 0.3989422804014327f * e;
}
__device__ __forceinline__ float gelu_f(float x) {
    float cdf, pdf;
    gelu_parts(x, cdf, pdf);
    return x * cdf;
}
// u = gelu(x) * keep, g = gelu'(x) * keep from one evaluation of (Phi, phi)
__device__ __forceinline__ void gelu_fwd_grad(float x, float keep, float& u, float& g) {
    float cdf, pdf;
    gelu_parts(x, cdf, pdf);
    u = x * cdf * keep;
    g = fmaf(x, pdf, cdf) * keep;
}
__device__ __forceinline__ float gelu_grad(float x) {
    float cdf, pdf;
    gelu_parts(x, cdf, pdf);
    return fmaf(x, pdf, cdf);
}

// Tiling configuration of the linear kernels: WM x WN waves, each owning TM x TN MFMA tiles of
// 32x32; BK = K-slab depth in elements; OCC = resident blocks (of THREADS threads) per CU;
// EPV = elements per 16-byte staging vector (4 fp32, 8 bf16); PAD = LDS row padding in elements.
template <int WM_, int WN_, int TM_, int TN_, int BK_ = 32, int OCC_ = 2, int EPV_ = 4, int PAD_ = 4> struct TileCfg {
    static constexpr int WM = WM_, WN = WN_, TMW = TM_, TNW = TN_, BK = BK_, OCC = OCC_, EPV = EPV_;
    static constexpr int LDT = BK + PAD_;                      // LDS tile row stride (elements)
    static constexpr int THREADS = WM * WN * 64;
    static constexpr int BM = WM * TMW * 32, BN = WN * TNW * 32;
    static constexpr int TPR = BK / EPV;                       // threads per staged row (16 B each)
    static constexpr int RPP = THREADS / TPR;                  // rows staged per pass
    static constexpr int PA = BM / RPP, PW = BN / RPP;
    static constexpr int SLOTS = 256 * OCC;                    // resident blocks on the chip (256 CUs)
};
