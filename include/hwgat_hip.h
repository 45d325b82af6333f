/*
 * hwgat_hip.h -- C ABI of libhwgat_hip.so, the MI355X (gfx950) backend for the
 * HWGAT hot path (reference: hwgat/models/HWGATE.py).
 *
 * The reference has no native interface of its own (it is 100 % Python on
 * ATen), so every entry point below cites the reference *Python* lines whose
 * arithmetic it replaces.  Conventions for all entry points:
 *
 *   - plain device pointers + sizes, no torch types; the caller owns every
 *     buffer, nothing is allocated or freed here; no global mutable state and no
 *     environment variables are read (the kernel lab's A/B switches exist only in
 *     -DHWGAT_LAB builds, csrc/common.h);
 *   - `dtype` selects activation storage: HWGAT_F32 (0) or HWGAT_BF16 (1);
 *     parameters, statistics and all arithmetic are fp32;
 *   - `stream` is a hipStream_t passed as void*; launches are asynchronous and
 *     stream-ordered; entry points are re-entrant;
 *   - return 0 on success, a negative HWGAT_E* code on bad arguments (nothing
 *     is launched then), or a positive hipError_t if the launch failed.
 *
 * Activations are kept in the natural token order (B, F, K, d) everywhere:
 * window partition / reverse / cyclic roll (HWGATE.py:30-47,197-215) are pure
 * index arithmetic inside the kernels and never materialised.
 */
#ifndef HWGAT_HIP_H
#define HWGAT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HWGAT_F32 0
#define HWGAT_BF16 1

#define HWGAT_EINVAL (-1)   /* null pointer / non-positive size            */
#define HWGAT_ESHAPE (-2)   /* unsupported shape (head_dim, width, ...)    */
#define HWGAT_EDTYPE (-3)   /* unknown dtype code                          */

/* library / ABI version: major*1000 + minor.  The major number changes whenever an entry point is removed or its
 * argument list / code tables (pro, epi, dtype) change meaning; a binding must compare hwgat_abi_version() with the
 * HWGAT_ABI_VERSION of the header it was written against and refuse a mismatch (sl-hwgat_amd/_lib.py does).
 *   1000  round 1 (incl. hwgat_split3_bf16, hwgat_linear_nt_f32x9 -- removed in 2000)
 *   2000  round 2: *_ex linears, ln_fold / ln_finalize, epilogues 5 / 6, masked-gradient producers
 *   3000  round 3: see INTEGRATION.md section 3
 *   4000  round 4: every entry point with a dropout seed takes `const uint32_t* seed_base` in front of `stream`;
 *         hwgat_seed_set / hwgat_seed_advance; hwgat_is_lab_build; hwgat_blk_attn_*_drop, hwgat_band_attn_*_drop;
 *         hwgat_ln_bwd_det, hwgat_linear_tn_*_det (bit-reproducible parameter gradients) */
#define HWGAT_ABI_VERSION 4000
int hwgat_abi_version(void);

/* ---- dropout seeds (round 4).  Every `*_seed` argument below is a SITE seed, a host integer that identifies one dropout
 * site of one block.  The seed a kernel hashes with is  site_seed + *seed_base  (32-bit wrap), where `seed_base` is a
 * DEVICE word read when the kernel runs; NULL means 0 (the site seed is then the whole seed, as before ABI 4000).
 * hwgat_seed_advance(state): state = 4 device words { step counter, base seed of the current step, initial seed, rank
 * salt }; one thread does  counter += 1; base = initial * 0x9E3779B1 + counter * 0x85EBCA77 + salt * 0x27D4EB2F.  A model
 * passes &state[1] as `seed_base` to every launch of a train step and runs hwgat_seed_advance once per step: no host
 * integer of the step depends on the step number, so the whole step can be captured in a HIP graph and replayed with
 * fresh masks (reference: nn.Dropout draws from the device generator, hwgat/models/HWGATE.py:27,116,133,135). */
int hwgat_seed_advance(uint32_t* state, void* stream);
/* the same state written from host integers (kernel arguments, no copy): state = { counter, base(counter), initial, salt }.
 * The eager path of a model calls it once per train-mode forward; after it, hwgat_seed_advance continues from `counter`. */
int hwgat_seed_set(uint32_t* state, uint32_t counter, uint32_t initial, uint32_t salt, void* stream);
/* 1 for the kernel-lab build (`python sl-hwgat_amd/build.py --lab`, libhwgat_hip_lab.so: environment A/B switches compiled
 * in), 0 for the product library (reads no environment variables).  The lab tools assert 1 on the library they load. */
int hwgat_is_lab_build(void);

/* ---- debug: dump the lane->element maps of v_mfma_f32_32x32x2_f32 so the
 * host can verify the operand layouts the kernels assume.  out: 64*16 floats
 * = D tile of A(32x2) . B(2x32) with A[i][k] = a[i*2+k], B[k][j] = b[k*32+j]. */
int hwgat_debug_mfma32x32x2(const float* a, const float* b, float* out, void* stream);

/* ---- debug: register-only loop of `iters` x 4 x n_acc v_mfma_f32_32x32x2_f32 per wave (n_acc in
 * {4,16} independent accumulators, `blocks` x 4 waves): the attainable fp32 MFMA rate of the part. */
int hwgat_debug_mfma_peak(float* out, int blocks, int iters, int n_acc, void* stream);

/* ---- a-2/a-3/a-12: part gather + Fourier features + positional encoding.
 * Replaces WindowCreate (dataTransform.py:445-455), the Fourier mapping
 * (HWGATE.py:343-345) and PositionalEncoding's add (HWGATE.py:25-27).
 *   x    (B, T, J, C) fp32 raw keypoints
 *   idx  (K) int32 joint index per model slot, or NULL for identity (J == K)
 *   bmat (d0/2, C) fp32 frozen Gaussian matrix (state_dict key "B")
 *   pe   (T, d0) fp32 sinusoid table or NULL (pe=False)
 *   out  (B, T, K, d0) `dtype`
 *   out[..., m] = sin(p_m) + pe, out[..., d0/2+m] = cos(p_m) + pe,
 *   p_m = sum_c (2*pi*x_c) * bmat[m][c]  (fp32, accurate range reduction).
 *   drop_p > 0: PositionalEncoding's Dropout (HWGATE.py:28) applied in the same pass with the
 *   hash mask of the fused linears (element index = flat index of `out`, seed `seed`). */
int hwgat_embed_fwd(const float* x, const int32_t* idx, const float* bmat, const float* pe,
                    void* out, int B, int T, int J, int K, int C, int d0, int dtype,
                    uint32_t seed, float drop_p, const uint32_t* seed_base, void* stream);

/* ---- LayerNorm over the last axis (HWGATE.py:203, 219, 353), eps 1e-5.
 *   x, y (N, d) `dtype`; gamma, beta (d) fp32; mean, rstd (N) fp32 (saved
 *   for backward).  d in {128, 256, 512, 1024}.  y may be NULL: statistics only
 *   (the fused linears normalise on the fly from mean/rstd). */
int hwgat_ln_fwd(const void* x, const float* gamma, const float* beta, void* y,
                 float* mean, float* rstd, int64_t N, int d, int dtype, void* stream);

/* backward of hwgat_ln_fwd.  dx = dLN/dx (+ dres if dres != NULL, the
 * shortcut gradient of HWGATE.py:217/219); dgamma, dbeta (d) fp32 are
 * ACCUMULATED into (caller zeroes them once per step). */
int hwgat_ln_bwd(const void* dy, const void* x, const float* mean, const float* rstd,
                 const float* gamma, const void* dres, void* dx, float* dgamma, float* dbeta,
                 int64_t N, int d, int dtype, void* stream);

/* The same, plus dx_masked = dx * dropout-mask(mask_seed, element index) with keep-scale 1/(1-mask_p): the masked
 * gradient that the PRODUCER of this tensor needs in front of its Dropout (HWGATE.py:116,135 backward), written once
 * here instead of being re-hashed in the GEMM loaders that consume it.  dres is required (the block's shortcut). */
int hwgat_ln_bwd_masked(const void* dy, const void* x, const float* mean, const float* rstd,
                        const float* gamma, const void* dres, void* dx, float* dgamma, float* dbeta,
                        int64_t N, int d, int dtype, void* dx_masked, uint32_t mask_seed, float mask_p,
                        const uint32_t* seed_base, void* stream);

/* The same once more, and xn = LN(x) = xhat * gamma + beta written to `xn` (N, d) `dtype`: the layer input of the Linear
 * that follows this LayerNorm (HWGATE.py:203 -> :86, :219 -> :131), for that Linear's weight-gradient launch
 * (hwgat_linear_tn_*), which then needs no LayerNorm in its loaders.  dres required; dx_masked may be NULL. */
int hwgat_ln_bwd_xn(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                    const float* beta, const void* dres, void* dx, float* dgamma, float* dbeta, int64_t N, int d,
                    int dtype, void* dx_masked, uint32_t mask_seed, float mask_p, void* xn,
                    const uint32_t* seed_base, void* stream);

/* Bit-reproducible form of the three LayerNorm-backward entry points above (round 4, "deterministic training"): every
 * option in one call -- beta and xn both given or both NULL; dres optional unless dx_masked or xn is given; dx_masked
 * optional -- and dgamma / dbeta summed in a FIXED order: each block stores its column sums into its own image of `ws`
 * (hwgat_ln_bwd_det_bytes(d) bytes, need not be zeroed) and a second pass adds the images in block order (no atomics). */
int64_t hwgat_ln_bwd_det_bytes(int d);
int hwgat_ln_bwd_det(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                     const float* beta, const void* dres, void* dx, float* dgamma, float* dbeta, int64_t N, int d,
                     int dtype, void* dx_masked, uint32_t mask_seed, float mask_p, void* xn,
                     const uint32_t* seed_base, float* ws, int64_t ws_bytes, void* stream);

/* ---- a-4/a-5/a-6/a-10: fused window attention (MSA.forward, HWGATE.py:89-114)
 * over the body-part joint graph, with partition/roll/reverse as index math.
 *   qkv      (B, F, K, 3, nH, hd) `dtype` -- the qkv Linear output in natural
 *            token order (HWGATE.py:86 column order [q|k|v][head][hd])
 *   o        (B, F, K, nH, hd) `dtype`   -- heads concatenated (HWGATE.py:114)
 *   maskbits (2, nW, 32) uint32: bit j of word [s][w][i] = key j visible to
 *            query i in part window w; s=0 adjacency only (HWGATE.py:106-108),
 *            s=1 adjacency AND last-slot shift mask (HWGATE.py:102-104,169-187)
 *   thr      device pointer to ONE fp32 probability threshold (train mode,
 *            HWGATE.py:94-100) or NULL for eval mode
 *   shifted  1 for odd blocks (roll by -1 frame before, +1 after; HWGATE.py:197-211)
 * K = nW*16, F even, hd in {32, 64, 128}.  Semantics incl. the "== 0 -> -10000"
 * fill and uniform all-masked rows are exactly SURVEY.md 8a "MSA exact semantics". */
int hwgat_win_attn_fwd(const void* qkv, void* o, const uint32_t* maskbits, const float* thr,
                       int B, int F, int nW, int nH, int hd, int shifted, int dtype,
                       void* stream);

/* backward: do (B,F,K,nH,hd) -> dqkv (B,F,K,3,nH,hd); probabilities are
 * recomputed from qkv (+ the same thr), nothing is saved by the forward. */
int hwgat_win_attn_bwd(const void* qkv, const void* dO, void* dqkv, const uint32_t* maskbits,
                       const float* thr, int B, int F, int nW, int nH, int hd, int shifted,
                       int dtype, void* stream);

/* the same pair with ATTENTION DROPOUT (reference HWGATE.py:78,112: nn.Dropout(attn_drop) on the softmax output,
 * `attn_drop_rate` of HWGATE.py:273): the probabilities are multiplied by mask / (1 - drop_p) before the P V product.
 * The mask is a hash of (drop_seed, element index of the reference's (B f nW, nH, 32, 32) attention tensor) and is
 * recomputed by the backward: hwgat_dropout_mask_f32(out, B (F/2) nW nH 1024, drop_seed, drop_p) returns exactly it.
 * drop_p in [0, 1); drop_p > 0 needs thr (dropout exists only in train mode), else HWGAT_EINVAL.
 * hwgat_win_attn_fwd / _bwd are these with drop_p = 0. */
int hwgat_win_attn_fwd_drop(const void* qkv, void* o, const uint32_t* maskbits, const float* thr,
                            int B, int F, int nW, int nH, int hd, int shifted, int dtype,
                            uint32_t drop_seed, float drop_p, const uint32_t* seed_base, void* stream);
int hwgat_win_attn_bwd_drop(const void* qkv, const void* dO, void* dqkv, const uint32_t* maskbits,
                            const float* thr, int B, int F, int nW, int nH, int hd, int shifted,
                            int dtype, uint32_t drop_seed, float drop_p, const uint32_t* seed_base, void* stream);

/* ---- (f) rank 3, sibling model HGATE: fused BLOCK attention (MSA.forward of
 * hwgat/models/HGATE.py:84-108) with block_partition / block_reverse / torch.roll
 * (HGATE.py:30-47,184-207) as index math.  A block is 2 frames x KJ joints (all joints
 * of the skeleton: 29 in HGATEParams), no part windows and no train-mode threshold.
 *   qkv      (B, F, KJ, 3, nH, hd) `dtype`, o (B, F, KJ, nH, hd) `dtype`
 *   maskbits (2, 64, 2) uint32: query slot i = tp*32 + joint (tp = frame of the pair);
 *            bit j of word [s][i][kt] = key joint j of frame kt visible to query i;
 *            s=0 adjacency only (HGATE.py:100-102), s=1 adjacency AND the shift mask
 *            of the LAST block of a shifted layer (HGATE.py:96-98,154-172).  Slots
 *            with joint >= KJ are padding: never read, written or counted as keys.
 *   shifted  1 for odd blocks (roll by -1 frame before, +1 after; HGATE.py:185-207)
 * F even, 1 <= KJ <= 32, hd in {32, 64}. */
int hwgat_blk_attn_fwd(const void* qkv, void* o, const uint32_t* maskbits,
                       int B, int F, int KJ, int nH, int hd, int shifted, int dtype,
                       void* stream);

/* backward: do (B,F,KJ,nH,hd) -> dqkv (B,F,KJ,3,nH,hd); probabilities recomputed. */
int hwgat_blk_attn_bwd(const void* qkv, const void* dO, void* dqkv, const uint32_t* maskbits,
                       int B, int F, int KJ, int nH, int hd, int shifted, int dtype,
                       void* stream);

/* The same with attention dropout (reference HGATE.py:78,106: nn.Dropout on the softmax output, train mode): the
 * probabilities are multiplied by mask / (1 - drop_p), mask = the common hash over the element index of the reference's
 * (B F/2, nH, 2 KJ, 2 KJ) attention tensor (token = frame * KJ + joint): hwgat_dropout_mask_f32(out, B (F/2) nH (2 KJ)^2,
 * drop_seed, drop_p) returns exactly it.  The backward recomputes the mask.  drop_p = 0: the plain kernels, bit for bit. */
int hwgat_blk_attn_fwd_drop(const void* qkv, void* o, const uint32_t* maskbits, int B, int F, int KJ, int nH, int hd,
                            int shifted, int dtype, uint32_t drop_seed, float drop_p, const uint32_t* seed_base,
                            void* stream);
int hwgat_blk_attn_bwd_drop(const void* qkv, const void* dO, void* dqkv, const uint32_t* maskbits, int B, int F, int KJ,
                            int nH, int hd, int shifted, int dtype, uint32_t drop_seed, float drop_p,
                            const uint32_t* seed_base, void* stream);

/* ---- (f) rank 3, sibling model WGATE: fused BAND attention (MSA.forward of
 * hwgat/models/WGATE.py:87-108) with window_partition / window_reverse (WGATE.py:32-65)
 * as index math.  A WGATE window is one 16-joint part window over ALL F frames with an
 * additive 0 / -10000 mask (WGATE.py:97-100,190) from a block-tridiagonal adjacency
 * (model_params.py:209-228): only keys of frames f-1, f, f+1 can carry weight.
 *   qkv      (B, F, K, 3, nH, hd) `dtype`, o (B, F, K, nH, hd) `dtype`, K = nW*16
 *   maskrows (nW, 16) uint64: bit (16*t + j) of row [w][i] = key joint j of frame
 *            f-1+t (t = 0,1,2) visible to query joint i of frame f, the same for every f
 *            (frames outside the clip are dropped by the kernel); every row must have at
 *            least one visible key in t = 1 (the reference's adjacency has a unit diagonal)
 * hd in {16, 32}; any F >= 1. */
int hwgat_band_attn_fwd(const void* qkv, void* o, const uint64_t* maskrows,
                        int B, int F, int nW, int nH, int hd, int dtype, void* stream);

/* backward: do (B,F,K,nH,hd) -> dqkv (B,F,K,3,nH,hd); probabilities recomputed. */
int hwgat_band_attn_bwd(const void* qkv, const void* dO, void* dqkv, const uint64_t* maskrows,
                        int B, int F, int nW, int nH, int hd, int dtype, void* stream);

/* The same with attention dropout (reference WGATE.py:81,103): mask over the element index of the reference's DENSE
 * (B nW, nH, F 16, F 16) attention tensor (token = frame * 16 + joint) -- only the band entries are ever evaluated.
 * hwgat_dropout_mask_f32(out, B nW nH (F 16)^2, drop_seed, drop_p) is the whole mask. */
int hwgat_band_attn_fwd_drop(const void* qkv, void* o, const uint64_t* maskrows, int B, int F, int nW, int nH, int hd,
                             int dtype, uint32_t drop_seed, float drop_p, const uint32_t* seed_base, void* stream);
int hwgat_band_attn_bwd_drop(const void* qkv, const void* dO, void* dqkv, const uint64_t* maskrows, int B, int F, int nW,
                             int nH, int hd, int dtype, uint32_t drop_seed, float drop_p, const uint32_t* seed_base,
                             void* stream);

/* debug: one v_mfma_f32_16x16x4_f32 with a (16x4), b (4x16) row-major -> out (64 lanes x 4 regs) */
int hwgat_debug_mfma16x16x4(const float* a, const float* b, float* out, void* stream);

/* ---- a-11: final LayerNorm + mean over all f*K tokens (HWGATE.py:353-354).
 *   x (B, n_tok, d) `dtype`; feat (B, d) fp32 must be ZERO on entry (sums of
 *   normalised values are accumulated, then hwgat_lnpool_finish scales them);
 *   mean, rstd (B*n_tok) fp32 saved. */
int hwgat_lnpool_fwd(const void* x, float* xhat_sum, float* mean, float* rstd,
                     int B, int n_tok, int d, int dtype, void* stream);
/* Bit-reproducible form (the reference's eval() forward is deterministic, HWGATE.py:352-360): every block of the
 * launch stores its partial sum into `partial` (B * hwgat_lnpool_partial_rows(B, n_tok) * d floats, caller-owned) and a
 * second launch adds a clip's rows in index order into xhat_sum (which need NOT be zero on entry).  partial == NULL
 * is hwgat_lnpool_fwd (fp32 atomics, summation order varies from run to run). */
int hwgat_lnpool_partial_rows(int B, int n_tok);
int hwgat_lnpool_fwd_det(const void* x, float* xhat_sum, float* mean, float* rstd,
                         int B, int n_tok, int d, int dtype, float* partial, void* stream);
/* backward: g (B, d) fp32 = dfeat * gamma / n_tok  ->  dx (B, n_tok, d) */
int hwgat_lnpool_bwd(const float* g, const void* x, const float* mean, const float* rstd,
                     void* dx, int B, int n_tok, int d, int dtype, void* stream);
/* ... and dx_masked = dx * dropout-mask(mask_seed, element index), see hwgat_ln_bwd_masked (dx_masked may be NULL) */
int hwgat_lnpool_bwd_masked(const float* g, const void* x, const float* mean, const float* rstd,
                            void* dx, int B, int n_tok, int d, int dtype, void* dx_masked, uint32_t mask_seed,
                            float mask_p, const uint32_t* seed_base, void* stream);

/* ---- a-9: TemporalMerging (HWGATE.py:55-63): (B,F,K,d) -> (B,F/2,K,2d),
 * out[b,fi,k,tp*d+c] = in[b,2fi+tp,k,c]; `inverse` = 1 maps gradients back. */
int hwgat_merge(const void* in, void* out, int B, int F, int K, int d, int inverse,
                int dtype, void* stream);
/* inverse mapping of a gradient plus a second, dropout-masked copy (mask of (mask_seed, un-merged element index)):
 * what the last block of a stage needs in front of its fc2 Dropout (HWGATE.py:135 backward) */
int hwgat_unmerge_masked(const void* in, void* out, void* out_masked, int B, int F, int K, int d, int dtype,
                         uint32_t mask_seed, float mask_p, const uint32_t* seed_base, void* stream);

/* ---- a-6/a-7/a-8: fp32 Linear layers on f32 MFMA with fused elementwise work.
 * Replaces nn.Linear (HWGATE.py:86,115,131,134) + bias + GELU (:132) + Dropout
 * (:116,:133,:135) + residual adds (:217,:219) forward, and their dX backward.
 *   C[M,N] = pro(A)[M,K] . W[N,K]^T, all fp32 row-major; N % 128 == K % 32 == 0, any M >= 1
 *   (a ragged last 128-row block clamps its loads to row M-1 and guards its stores).
 *   pro: 0 none | 1 LayerNorm: (A-mean[m])*rstd[m]*gamma[k]+beta[k] | 2 dropout mask on A
 *        (keep-scale 1/(1-pro_p), element index m*K+k, seed pro_seed)
 *        | 3 folded LayerNorm (hwgat_ln_fold below): A is the un-normalised input, W = W o gamma, gamma = s[N],
 *        beta = c[N], bias ignored; the epilogue forms rstd[m] (acc - mean[m] s[n]) + c[n] -- the value of pro 1 up
 *        to rounding, with the per-element normalisation out of the load path; epi 0 or 2 only, M % 128 == 0
 *   epi: 0  C = acc + bias
 *        1  C = res + dropout(acc + bias)             (mask index m*N+n, seed epi_seed)
 *        2  C2 = acc + bias ; C = dropout(gelu(C2))   (exact-erf GELU)
 *        3  C = acc * dropmask * gelu'(aux)           (backward of epi 2; aux = saved C2)
 *        4  C = acc
 *        5  C2 = gelu'(acc + bias) * dropmask ; C = dropout(gelu(acc + bias))   (training form of epi 2: the factor the
 *           backward needs is stored instead of the pre-activation)
 *        6  C = acc * aux                             (backward of epi 5; aux = its saved C2)
 *   bias may be NULL (treated as 0).  For dX pass W = transposed weight. */
int hwgat_linear_nt_f32(const float* A, const float* W, const float* bias, float* C, int64_t M, int N,
                        int K, int pro, const float* mean, const float* rstd, const float* gamma,
                        const float* beta, uint32_t pro_seed, float pro_p, int epi, const float* res,
                        float* C2, const float* aux, uint32_t epi_seed, float epi_p,
                        const uint32_t* seed_base, void* stream);

/* The same launch for the two linears whose OUTPUT is the input of a LayerNorm (proj -> norm2, HWGATE.py:217-219;
 * fc2 -> the next block's norm1, :219 -> :203, through TemporalMerging :55-63 at a stage end).  Requires pro = 0,
 * epi = 1, M % 256 == 0.  In addition to hwgat_linear_nt_f32:
 *   stat_sum, stat_sq  (rows of the output) fp32, ZEROED by the caller: per-row sum / sum of squares of the stored
 *                      output values are accumulated into them (hwgat_ln_finalize then yields mean / rstd), which
 *                      removes the separate statistics pass over the tensor;
 *   merge_K > 0        the output is stored in the TemporalMerging layout: row (b, f, k) of the (B, merge_F, merge_K, N)
 *                      result goes to row (b, f/2, k), columns (f & 1) N .. (f & 1) N + N - 1 of a (B, merge_F/2,
 *                      merge_K, 2N) tensor C; statistics are then per merged row (2N values).  res stays in the
 *                      natural layout. */
int hwgat_linear_nt_f32_ex(const float* A, const float* W, const float* bias, float* C, int64_t M, int N,
                           int K, int pro, const float* mean, const float* rstd, const float* gamma,
                           const float* beta, uint32_t pro_seed, float pro_p, int epi, const float* res,
                           float* C2, const float* aux, uint32_t epi_seed, float epi_p, float* stat_sum,
                           float* stat_sq, int merge_F, int merge_K, const uint32_t* seed_base, void* stream);

/* Weights of a Linear that follows a LayerNorm (norm1 -> qkv, HWGATE.py:203-205 / :86; norm2 -> fc1, :219 / :131),
 * folded for pro = 3 of the NT launches:  Wf[n,k] = W[n,k] gamma[k] in `dtype` (HWGAT_F32 / HWGAT_BF16),
 * s[n] = sum_k Wf[n,k] (of the stored values), c[n] = bias[n] + sum_k beta[k] W[n,k].  W, bias, gamma, beta fp32
 * (the master weights); bias may be NULL.  One launch per step and linear (N x K elements). */
int hwgat_ln_fold(const float* W, const float* bias, const float* gamma, const float* beta, int N, int K,
                  void* Wf, float* s, float* c, int dtype, void* stream);

/* (row sum, row sum of squares) of a d-wide tensor -> (mean, rstd) in place, nn.LayerNorm's biased variance and
 * eps 1e-5 (HWGATE.py:162,166). */
int hwgat_ln_finalize(float* sum_mean, float* sq_rstd, int64_t n, int d, void* stream);

/* weight/bias gradient: dW[N,K] += dropmask(A)[M,N]^T . ln(B)[M,K] ; db[N] += colsum(dropmask(A))
 * (db may be NULL).  Accumulates with fp32 atomics across M slices: caller provides zeroed
 * (or to-be-accumulated-into) dW/db.  N % 128 == K % 128 == 0, any M >= 1.  pro_p == 0: no mask.
 * mean != NULL: B is LayerNorm-ed on the fly, (B-mean[m])*rstd[m]*gamma[k]+beta[k]. */
int hwgat_linear_tn_f32(const float* A, const float* B, float* dW, float* db, int64_t M, int N, int K,
                        uint32_t pro_seed, float pro_p, const float* mean, const float* rstd,
                        const float* gamma, const float* beta, const uint32_t* seed_base, void* stream);

/* ---- BASELINE config 3: the same two linears with bf16 activations / weights on
 * v_mfma_f32_32x32x16_bf16 (fp32 accumulate).  A, W, C, C2, res, aux are bf16; bias, LN
 * statistics/affine, dW, db stay fp32.  Semantics, prologue/epilogue codes and dropout masks are
 * identical to the fp32 entry points (any M >= 1: a ragged tail gets its own small launch); K % 64 == 0 here. */
int hwgat_linear_nt_bf16(const void* A, const void* W, const float* bias, void* C, int64_t M, int N, int K,
                         int pro, const float* mean, const float* rstd, const float* gamma,
                         const float* beta, uint32_t pro_seed, float pro_p, int epi, const void* res,
                         void* C2, const void* aux, uint32_t epi_seed, float epi_p,
                         const uint32_t* seed_base, void* stream);
/* hwgat_linear_nt_f32_ex for bf16 activations: the statistics are those of the bf16-rounded output values (what the
 * next LayerNorm reads), the merged store writes bf16. */
int hwgat_linear_nt_bf16_ex(const void* A, const void* W, const float* bias, void* C, int64_t M, int N, int K,
                            int pro, const float* mean, const float* rstd, const float* gamma,
                            const float* beta, uint32_t pro_seed, float pro_p, int epi, const void* res,
                            void* C2, const void* aux, uint32_t epi_seed, float epi_p, float* stat_sum,
                            float* stat_sq, int merge_F, int merge_K, const uint32_t* seed_base, void* stream);
int hwgat_linear_tn_bf16(const void* A, const void* B, float* dW, float* db, int64_t M, int N, int K,
                         uint32_t pro_seed, float pro_p, const float* mean, const float* rstd,
                         const float* gamma, const float* beta, const uint32_t* seed_base, void* stream);
/* The same for plain operands (no dropout mask, no LayerNorm) with a caller-owned workspace of
 * hwgat_linear_tn_bf16_ws_bytes(M, N, K) bytes: the partial dW tiles of the M slices are written as slabs and added in a
 * FIXED order by a second launch (no global atomics: ~20 us less per launch at the HWGAT shapes, dW bit-reproducible).
 * ws == NULL, too small, or a shape the slab kernel does not take (the query returns 0): hwgat_linear_tn_bf16. */
int64_t hwgat_linear_tn_bf16_ws_bytes(int64_t M, int N, int K);
int hwgat_linear_tn_bf16_ws(const void* A, const void* B, float* dW, float* db, int64_t M, int N, int K,
                            float* ws, int64_t ws_bytes, void* stream);

/* Bit-reproducible weight / bias gradients (round 4): the same kernels as hwgat_linear_tn_f32 / _bf16 (any prologue), but
 * the block of M split s STORES its partial dW tile and bias gradient into image s of the caller's ZERO-FILLED workspace
 * and one pass adds the images in split order -- no float atomics, the same bits on every run.  M % 32 == 0;
 * ws_bytes >= hwgat_linear_tn_det_bytes(M, N, K) (0: shape not supported).  Slower than the atomic / slab forms by the
 * workspace round trip; the reference's own single-device training is reproducible, this is the mode that matches it. */
int64_t hwgat_linear_tn_det_bytes(int64_t M, int N, int K);
int hwgat_linear_tn_f32_det(const float* A, const float* B, float* dW, float* db, int64_t M, int N, int K,
                            uint32_t pro_seed, float pro_p, const float* mean, const float* rstd,
                            const float* gamma, const float* beta, const uint32_t* seed_base, float* ws,
                            int64_t ws_bytes, void* stream);
int hwgat_linear_tn_bf16_det(const void* A, const void* B, float* dW, float* db, int64_t M, int N, int K,
                             uint32_t pro_seed, float pro_p, const float* mean, const float* rstd,
                             const float* gamma, const float* beta, const uint32_t* seed_base, float* ws,
                             int64_t ws_bytes, void* stream);

/* hwgat_linear_tn_f32 with a caller-owned workspace of hwgat_linear_tn_f32_ws_bytes(M, N, K) bytes: where the 256x256-tile
 * kernel takes the shape (N, K multiples of 256, more than one tile) the partial dW tiles of the M slices go to slabs and a
 * second launch adds them in a FIXED order instead of 64 MB of global float atomics (~25 us less per launch, dW
 * bit-reproducible).  Every argument as in hwgat_linear_tn_f32; ws == NULL, too small, or the query returned 0: identical
 * to hwgat_linear_tn_f32. */
int64_t hwgat_linear_tn_f32_ws_bytes(int64_t M, int N, int K);
int hwgat_linear_tn_f32_ws(const float* A, const float* B, float* dW, float* db, int64_t M, int N, int K,
                           uint32_t pro_seed, float pro_p, const float* mean, const float* rstd,
                           const float* gamma, const float* beta, float* ws, int64_t ws_bytes,
                           const uint32_t* seed_base, void* stream);

/* out[C,R] = in[R,C]^T (used on weights only) */
int hwgat_transpose_f32(const float* in, float* out, int R, int C, void* stream);

/* ---- per-step weight preparation: ONE launch makes every derived copy of the fp32 master weights the fused linears
 * consume, for all blocks of a model (no reference counterpart: operands of the kernels above).  `table` is an array of
 * `n` entries in DEVICE memory, sorted by first_block; entry i owns workgroups [first_block, next entry's first_block):
 *   op 0 copy:       out[N][K] = (dtype) W[N][K]                    ceil(N/32) * ceil(K/32) workgroups
 *   op 1 transpose:  out[K][N] = (dtype) W[N][K]^T                  ceil(N/32) * ceil(K/32) workgroups
 *   op 2 LN fold:    out = W o gamma (dtype), s[N], c[N] exactly as hwgat_ln_fold (bias may be NULL)   ceil(N/4) workgroups
 * total_blocks = the sum of the entries' workgroups; dtype = HWGAT_F32 | HWGAT_BF16 of every `out`. */
typedef struct {
    const float* W; const float* bias; const float* gamma; const float* beta;
    void* out; float* s; float* c;
    int32_t N, K, op, first_block;
} hwgat_prep_entry;
int hwgat_weight_prep(const hwgat_prep_entry* table, int n, int total_blocks, int dtype, void* stream);

/* the dropout mask the fused kernels use: out[i] = keep(seed, i) ? 1/(1-p) : 0 */
int hwgat_dropout_mask_f32(float* out, int64_t n, uint32_t seed, float p, const uint32_t* seed_base, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HWGAT_HIP_H */
