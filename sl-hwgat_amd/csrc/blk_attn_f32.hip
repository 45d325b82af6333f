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
//   stage   Q, K, V, dO of the unit: 4 images of 64 slot rows x 256 bytes.  The workgroups are persistent (two or three per
//           CU) and a wave holds the NEXT unit's rows in registers (16 bytes per lane and load, whole 256-byte rows, 64 / 48
//           registers), fetched while the current unit is computed and written to the images when they are free: with two
//           or three units resident per CU nothing else hides a unit's load latency (one workgroup per unit with LDS-DMA
//           staging left the matrix pipe idle half the time: LABLOG 10.8).  Pad slots (joint >= KJ) re-read joint KJ-1
//           and are masked out below.
//   phase A wave w owns QUERY slots 16w .. 16w+15.  S^T = K Q^T and dP^T = V dO^T: both sides as row operands (lane (l, g)
//           reads the 16-byte chunk 4 m + g of its row: one ds_read_b128 feeds four MFMAs; the k index of an MFMA is any
//           four channels as long as both operands agree).  Result lane = query, 4 x 4 registers = the lane's 16 of the 64
//           key slots.  Masks, the "== 0 -> -10000" fill (HGATE.py:104) and the softmax in registers; dS = P (dP - delta)
//           where the logit was kept.  O = P V / dQ = dS K: P (dS) straight from the registers as the A operand, and column j of
//           the B operand of channel tile ct = channel 4 j + ct, so ONE ds_read_b128 of chunk l of a key row feeds the four
//           channel tiles and the sixteen lanes of a result register hold one whole 256-byte row: 4 rows per store.
//   phase B wave w owns KEY slots 16w .. 16w+15: dK^T = Q^T dS, dV^T = dO^T P, with P and dS exchanged through two
//           [query][key] images that overlay the K and V images once every wave is through phase A.
//
// Every operand read is conflict-free under one XOR swizzle of the sixteen 16-byte chunks of a row (xrf below).  64 KB of LDS per workgroup in the backward pass (two workgroups = 8 waves per CU), 48 KB in the
// forward pass (three workgroups).  HBM traffic is the algorithmic 4 E s / 7 E s.
#include <stdlib.h>
#include "blk16_common.h"

namespace {
using namespace blk;

typedef __attribute__((address_space(3))) f32x4v lds_f32x4;
typedef __attribute__((address_space(3))) float lds_f32;

#ifdef HWGAT_LAB
// lab build only: s_memtime stamps of wave 0 of every workgroup, [workgroup][unit slot < 8][10], through a pointer set with
// hwgat_lab_blk_stamps (tools/blk_stamps.py)
__device__ unsigned long long* g_blk_stamps = nullptr;
#define BLK_STAMP(i)                                                                                              \
    do {                                                                                                          \
        if (g_blk_stamps && threadIdx.x == 0 && stamp_slot < 8)                                                   \
            g_blk_stamps[((size_t)blockIdx.x * 8 + stamp_slot) * 10 + (i)] = __builtin_amdgcn_s_memtime();       \
    } while (0)
#define BLK_STAMP_NEXT() ++stamp_slot
#define BLK_STAMP_DECL() int stamp_slot = 0
#else
#define BLK_STAMP(i) do {} while (0)
#define BLK_STAMP_NEXT() do {} while (0)
#define BLK_STAMP_DECL() do {} while (0)
#endif

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

typedef __attribute__((address_space(3))) u32x4v lds_u32x4f;

// ---- stage: wave w fetches slot rows 16w .. 16w+15 of each image (Q, K, V and, with NIMG = 4, dO): load jj covers rows
// 16w + 4jj .. + 3, lane l the 16-byte chunk l & 15 of row l >> 4 of them
struct LaneOffs { uint32_t q[4], g[4]; };                     // BYTE offsets of the lane's four loads inside a frame of qkv / dO
__device__ __forceinline__ LaneOffs lane_offs(const BlkGeom& g, int lane, int w) {
    LaneOffs lo;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const int row = 16 * w + 4 * jj + (lane >> 4);
        const uint32_t joint = min(row & 31, g.KJ - 1);          // pad slots re-read the last joint (finite data, masked later)
        lo.q[jj] = (joint * 3u * g.d + (lane & 15) * 4) * 4;
        lo.g[jj] = (joint * (uint32_t)g.d + (lane & 15) * 4) * 4;
    }
    return lo;
}
// wave-uniform base + 32-bit lane offset: the scalar-base form of global_load (no 64-bit address registers per load)
// (the empty asm keeps the 32 -> 64 bit extension of the offset in the block of the load: hoisted out of the loop, instruction
//  selection no longer sees it and falls back to 64-bit address registers, which it builds in the destination registers --
//  behind a wait for every outstanding store)
__device__ __forceinline__ u32x4v ld16(const float* base, uint32_t byte_off) {
    asm volatile("" : "+v"(byte_off));
    return __builtin_nontemporal_load(reinterpret_cast<const u32x4v*>(reinterpret_cast<const char*>(base) + byte_off));
}
// load jj of the four (the NIMG images' slot rows 16w + 4jj .. + 3)
template <int NIMG>
__device__ __forceinline__ void fetch_part(u32x4v (&r)[NIMG][4], int jj, const float* qkv, const float* dO, const BlkGeom& g,
                                           const BUnit& un, const LaneOffs& lo, int w) {
    const int64_t base = w >> 1 ? un.base[1] : un.base[0];
    const float* qb = qkv + base * 3 * g.d + un.head * HD;
    r[0][jj] = ld16(qb, lo.q[jj]);
    r[1][jj] = ld16(qb + g.d, lo.q[jj]);
    r[2][jj] = ld16(qb + 2 * g.d, lo.q[jj]);
    if constexpr (NIMG == 4) r[3][jj] = ld16(dO + base * g.d + un.head * HD, lo.g[jj]);
}
template <int NIMG>
__device__ __forceinline__ void fetch_unit(u32x4v (&r)[NIMG][4], const float* qkv, const float* dO, const BlkGeom& g,
                                           const BUnit& un, const LaneOffs& lo, int w) {
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) fetch_part<NIMG>(r, jj, qkv, dO, g, un, lo, w);
}
template <int NIMG>
__device__ __forceinline__ void put_unit(char* sm, const u32x4v (&r)[NIMG][4], int lane, int w) {
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const uint32_t off = chunk_off_f(16 * w + 4 * jj + (lane >> 4), lane & 15);
#pragma unroll
        for (int im = 0; im < NIMG; ++im) *(lds_u32x4f*)(sm + im * IMGF + off) = r[im][jj];
    }
}

// The workgroups that share a CU run the same code on the same schedule: started together they stay in step, want the matrix
// pipe in the same phases and leave it idle in the same phases (stamps: LABLOG 10.8).  The grid is `per_cu` waves of 256
// workgroups; workgroup i of wave k = i / 256 starts k * skew * 64 cycles late, so that the residents of a CU sit in different
// phases of a unit.
__device__ __forceinline__ void start_skew(int skew, int per_cu) {
    const int k = blockIdx.x / 256;
    for (int i = 0; i < k * (skew & 0xffff); ++i) __builtin_amdgcn_s_sleep(1);
}

// ---- products tile by tile (16 key / query slots), skipping the tiles that contribute exactly nothing.  The adjacency of a
// skeleton is sparse (0.075 for the reference's; 4 of its 16 tiles of 16 x 16 slots are empty): a tile of S whose keys no
// query of the wave sees is never looked at (the masks replace it by -10000), and a tile of P (dS) that is zero in every
// lane adds zeros to O, delta, dQ, dK, dV.  The tests are wave-uniform ballots on the registers themselves, so the result is
// the dense one bit for bit whatever the masks and the data (a row without any visible key has P != 0 everywhere and skips
// nothing); the branch keeps EXEC whole for the MFMAs.
// bit kt: some lane's (query's) mask words show a key of tile kt (keys 16 (kt & 1) .. + 15 of frame kt >> 1)
__device__ __forceinline__ uint32_t vis_tiles(uint32_t mb0, uint32_t mb1) {
    uint32_t v = 0;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
        v |= __builtin_amdgcn_ballot_w64((((kt >> 1 ? mb1 : mb0) >> (16 * (kt & 1))) & 0xffffu) != 0) != 0 ? 1u << kt : 0u;
    return v;
}
__device__ __forceinline__ bool tile_any(const f32x4v& x) {
    return __builtin_amdgcn_ballot_w64(x.x != 0.f || x.y != 0.f || x.z != 0.f || x.w != 0.f) != 0;
}
// the four row operands (channel chunks 4 m + g, m = 0..3) of slot row `row`
__device__ __forceinline__ void row_ops4(f32x4v (&x)[4], const char* img, int row, int gq) {
#pragma unroll
    for (int m = 0; m < 4; ++m) x[m] = row_op_f(img, row, m, gq);
}
// one 16 x 16 tile of X Y^T over the 64 channels: lane (l, g), register r = X row 4 g + r against Y row l.  Two accumulation
// chains: back-to-back MFMAs into one accumulator wait 40 cycles for each other where independent ones issue every 32.
__device__ __forceinline__ f32x4v xyT16(const f32x4v (&x)[4], const f32x4v (&y)[4]) {
    f32x4v a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        a = mfma4(x[m][0], y[m][0], a);
        b = mfma4(x[m][1], y[m][1], b);
        a = mfma4(x[m][2], y[m][2], a);
        b = mfma4(x[m][3], y[m][3], b);
    }
    return a + b;
}
// S^T-shaped product of the wave's 16 rows of `y` (row operands in registers) with the 64 slot rows of `img`, tile kt only
// where need(kt); the operands of tile kt + 1 are read while tile kt multiplies (read or not: a skipped tile costs 4 reads)
template <typename Need>
__device__ __forceinline__ void rows_xyT(f32x4v (&out)[4], const char* img, const f32x4v (&y)[4], int lr, int gq, Need need) {
    f32x4v x[2][4];
    row_ops4(x[0], img, lr, gq);
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
        if (kt < 3) row_ops4(x[(kt + 1) & 1], img, 16 * (kt + 1) + lr, gq);
        if (need(kt)) out[kt] = xyT16(x[kt & 1], y);
        else out[kt] = f32x4v{0.f, 0.f, 0.f, 0.f};
    }
}
// the four column operands (chunk l) of the slot rows 16 t + 4 g + r, r = 0..3, of `img`
__device__ __forceinline__ void col_ops4(f32x4v (&v)[4], const char* img, int t, int lr, int gq) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = col_op_f(img, 16 * t + 4 * gq + r, lr);
}
// acc[ct][r] += sum over the 16 slots of tile t of  b[r'] v[r'][ct]:  channel 4 l + ct of row 4 g + r of the wave's 16 queries
// (keys); b[r'] is the lane's value for slot 16 t + 4 g + r' (the A operand: row i of the product = the lane's l) and v[r'] one
// ds_read_b128 of chunk l of that slot row (the B operand: column j of channel tile ct = channel 4 j + ct).  The sixteen lanes
// l of a result register then hold one whole 256-byte row.
__device__ __forceinline__ void tile_times_col(const f32x4v& b, const f32x4v (&v)[4], f32x4v (&acc)[4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[ct] = mfma4(b[r], v[r][ct], acc[ct]);
}
// acc += b img over the 64 key slots (b = P or dS in the result layout of phase A); SKIP: tile by tile, where b has a non-zero
template <bool SKIP>
__device__ __forceinline__ void regs_times_col(const char* img, const f32x4v (&b)[4], int lr, int gq, f32x4v (&acc)[4]) {
    if constexpr (SKIP) {
        f32x4v v[2][4];
        col_ops4(v[0], img, 0, lr, gq);
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            if (kt < 3) col_ops4(v[(kt + 1) & 1], img, kt + 1, lr, gq);
            if (tile_any(b[kt])) tile_times_col(b[kt], v[kt & 1], acc);
        }
    } else {
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const f32x4v v = col_op_f(img, 16 * kt + 4 * gq + r, lr);
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) acc[ct] = mfma4(b[kt][r], v[ct], acc[ct]);
            }
    }
}
// rows 4 g + r < n_real of the wave's 16-row tile (scaled): 16 bytes per lane, 4 whole rows per store instruction (lanes that
// each own 16 channels of one row -- 64 scattered 16-byte pieces per instruction -- cost the backward pass 19 %: LABLOG 10.8)
__device__ __forceinline__ void store_rows(float* tile, int64_t row_stride, const f32x4v (&acc)[4], float scale, int gq, int n_real) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (4 * gq + r < n_real)
            __builtin_nontemporal_store(f32x4v{acc[0][r], acc[1][r], acc[2][r], acc[3][r]} * scale,
                                        reinterpret_cast<f32x4v*>(tile + (4 * gq + r) * row_stride));
}

// =============================================================== forward
template <bool ADROP>
__global__ __launch_bounds__(256, 3) void blk_fwd_f32_k(const float* __restrict__ qkv, float* __restrict__ o,
                                                        const uint32_t* __restrict__ maskbits, BlkGeom g, int n_units,
                                                        AttnDrop ad, int skew) {
    if constexpr (ADROP) ad.seed += seed_base_of(ad.base);
    __shared__ __attribute__((aligned(1024))) char sm[3 * IMGF];  // Q | K | V
    const char* Qt = sm;
    const char* Kt = sm + IMGF;
    const char* Vt = sm + 2 * IMGF;
    const int lane = threadIdx.x & 63, lr = lane & 15, gq = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int slot = 16 * w + lr;                                // this lane's query slot
    const int n_real = min(max(g.KJ - 16 * (w & 1), 0), 16);     // rows of the wave's 16-slot tile that are joints
    int u = blockIdx.x;                                          // (the grid has at most n_units workgroups)
    BUnit un = decode_bunit(g, u);
    const LaneOffs lo = lane_offs(g, lane, w);
    u32x4v nx[3][4];
    fetch_unit<3>(nx, qkv, nullptr, g, un, lo, w);
    start_skew(skew, 3);
    // the query's mask words, both variants (plain / last block of a shifted layer): no loads besides the prefetch in the loop
    const uint32_t mp0 = maskbits[slot * 2], mp1 = maskbits[slot * 2 + 1], ml0 = maskbits[(64 + slot) * 2], ml1 = maskbits[(64 + slot) * 2 + 1];
    // the unit in the images (tok0, head, ucur, mb0, mb1) and the one in flight (un, nx)
    int64_t tok0;                                                // first token of the wave's 16-slot tile
    int head, ucur;
    uint32_t mb0, mb1;
    bool more;
    auto advance = [&]() {
        tok0 = (w >> 1 ? un.base[1] : un.base[0]) + 16 * (w & 1);
        head = un.head;
        ucur = u;
        mb0 = un.mrow ? ml0 : mp0;
        mb1 = un.mrow ? ml1 : mp1;
        u += gridDim.x;
        more = u < n_units;
        if (more) un = decode_bunit(g, u);
    };
    put_unit<3>(sm, nx, lane, w);
    advance();
    wait_lds_barrier();
    for (;;) {
        // S[q][key 16 kt + 4g + r]: the four key tiles side by side (the forward pass has the registers of three waves per SIMD
        // to live in, and tile-by-tile products with their skip tests were slower here: LABLOG 10.8).  A quarter of the next
        // unit's rows is fetched per step: sixteen loads in one burst from every wave of the CU fill the address queue, and a
        // wave that cannot issue its load issues no MFMA either
        f32x4v s[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) s[kt] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            if (more) fetch_part<3>(nx, m, qkv, nullptr, g, un, lo, w);
            const f32x4v q = row_op_f(Qt, slot, m, gq);
            f32x4v k[4];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) k[kt] = row_op_f(Kt, 16 * kt + lr, m, gq);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) s[kt] = mfma4(k[kt][e], q[e], s[kt]);
            __builtin_amdgcn_sched_barrier(0);
        }
        uint32_t nz;
        const float inv = 1.0f / masked_exp64(s, mb0, mb1, gq, g.KJ, nz);
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) s[kt] *= inv;             // HGATE.py:105
        if constexpr (ADROP) {                                   // HGATE.py:106
            f32x4v keep[4];
            blk_keep16(keep, ad, ucur, slot, gq, g.KJ);
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) s[kt] *= keep[kt];
        }
        f32x4v acc[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[ct] = f32x4v{0.f, 0.f, 0.f, 0.f};
        regs_times_col<false>(Vt, s, lr, gq, acc);              // (products outside any lane-dependent branch: MFMAs want all lanes)
        // the next unit's rows go to the images BEFORE this unit's stores are issued: waiting for those loads (vmcnt counts in
        // order) then does not wait for the stores, which a wait at the top of the loop would (stamps: 18 % of a unit)
        float* otile = o + tok0 * (int64_t)g.d + head * HD + 4 * lr;
        if (!more) {
            store_rows(otile, g.d, acc, 1.0f, gq, n_real);
            break;
        }
        wait_lds_barrier();                                      // every wave is through with the images
        put_unit<3>(sm, nx, lane, w);
        advance();
        store_rows(otile, g.d, acc, 1.0f, gq, n_real);
        wait_lds_barrier();
    }
}

// =============================================================== backward
template <bool ADROP>
__global__ __launch_bounds__(256, 2) void blk_bwd_f32_k(const float* __restrict__ qkv, const float* __restrict__ dO,
                                                        float* __restrict__ dqkv, const uint32_t* __restrict__ maskbits,
                                                        BlkGeom g, int n_units, AttnDrop ad, int skew) {
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
    const int64_t rs = 3 * (int64_t)g.d;                         // qkv row stride (elements)
    const int slot = 16 * w + lr;                                // this lane's query slot in phase A / key slot in phase B
    const bool real = (slot & 31) < g.KJ;
    int n_real = min(max(g.KJ - 16 * (w & 1), 0), 16);           // rows of the wave's 16-slot tile that are joints
#ifdef HWGAT_LAB
    if (skew & 0x10000) n_real = 0;                              // lab ablation: no stores
    const bool no_fetch = skew & 0x20000;                        // lab ablation: the first unit's rows for every unit
    skew &= 0xffff;
#endif
    int u = blockIdx.x;                                          // (the grid has at most n_units workgroups)
    BUnit un = decode_bunit(g, u);
    const LaneOffs lo = lane_offs(g, lane, w);
    u32x4v nx[4][4];
    fetch_unit<4>(nx, qkv, dO, g, un, lo, w);
    start_skew(skew, 2);
    // the query's mask words, both variants (plain / last block of a shifted layer): no loads besides the prefetch in the loop
    const uint32_t mp0 = maskbits[slot * 2], mp1 = maskbits[slot * 2 + 1], ml0 = maskbits[(64 + slot) * 2], ml1 = maskbits[(64 + slot) * 2 + 1];
    const uint32_t vis_p = vis_tiles(mp0, mp1), vis_l = vis_tiles(ml0, ml1);
    // the unit in the images (tok0, head, ucur, mb0, mb1: the mask words of the query) and the one in flight (un, nx)
    int64_t tok0;                                                // first token of the wave's 16-slot tile
    int head, ucur;
    uint32_t mb0, mb1, vis;                                      // vis: bit kt = some query of the wave sees a key of tile kt
    bool more;
    auto advance = [&]() {
        tok0 = (w >> 1 ? un.base[1] : un.base[0]) + 16 * (w & 1);
        head = un.head;
        ucur = u;
        mb0 = un.mrow ? ml0 : mp0;
        mb1 = un.mrow ? ml1 : mp1;
        vis = un.mrow ? vis_l : vis_p;
        u += gridDim.x;
        more = u < n_units;
        if (more) un = decode_bunit(g, u);
    };
    BLK_STAMP_DECL();
    put_unit<4>(sm, nx, lane, w);
    advance();
    wait_lds_barrier();
    for (;;) {
        BLK_STAMP(2);

        // ================================================= phase A: query slot `slot`
        // S[q][key 16 kt + 4g + r], only the key tiles some query of the wave sees.  A quarter of the next unit's rows is
        // fetched per tile: sixteen loads in one burst from every wave of the CU fill the address queue, and a wave that cannot
        // issue its load issues no MFMA either
        f32x4v s[4], dp[4];
        {
            f32x4v q[4];
            row_ops4(q, Qt, slot, gq);
            rows_xyT(s, Kt, q, lr, gq, [&](int kt) {
#ifdef HWGAT_LAB
                if (!no_fetch)
#endif
                if (more) fetch_part<4>(nx, kt, qkv, dO, g, un, lo, w);
                return (vis >> kt) & 1;
            });
        }
        BLK_STAMP(3);
        uint32_t nz;
        const float sum = masked_exp64(s, mb0, mb1, gq, g.KJ, nz);
        const float inv = real ? 1.0f / sum : 0.f;                               // pad query slots: P = dS = 0
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) s[kt] *= inv;                             // P (HGATE.py:105)
        // dP[q][key] = dO V^T, only the tiles where P is not zero throughout (delta and dS take nothing from the others)
        {
            f32x4v gd[4];
            row_ops4(gd, Gt, slot, gq);
            rows_xyT(dp, Vt, gd, lr, gq, [&](int kt) { return tile_any(s[kt]); });
        }
        // attention dropout: A = D o P went into O = A V, so dP = D o dA (dA = dO V^T, in `dp`) and dV = A^T dO; mask recomputed
        f32x4v keep[ADROP ? 4 : 1];
        if constexpr (ADROP) blk_keep16(keep, ad, ucur, slot, gq, g.KJ);
        float delta = 0.f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            if constexpr (ADROP) dp[kt] *= keep[kt];
#pragma unroll
            for (int r = 0; r < 4; ++r) delta = __builtin_fmaf(s[kt][r], dp[kt][r], delta);
        }
        delta = xg_sum(delta);
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) dp[kt][r] = ((nz >> (4 * kt + r)) & 1u) ? s[kt][r] * (dp[kt][r] - delta) : 0.f;   // dS
            if constexpr (ADROP) s[kt] *= keep[kt];                              // the P image feeds dV only: A = D o P
        }
        BLK_STAMP(4);
        float* gtile = dqkv + tok0 * rs + head * HD + 4 * lr;
        // dQ[q][c] = scale * sum_key dS[q][key] K[key][c]
        {
            f32x4v acc[4];
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) acc[ct] = f32x4v{0.f, 0.f, 0.f, 0.f};
            regs_times_col<true>(Kt, dp, lr, gq, acc);
            store_rows(gtile, rs, acc, SCALE, gq, n_real);
        }
        BLK_STAMP(5);
        wait_lds_barrier();                                      // every wave is through with the K and V images
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {                         // [query][key] images: keys 16 kt + 4g .. + 3 = chunk 4 kt + g
            *(lds_f32x4*)(Pt + chunk_off_f(slot, 4 * kt + gq)) = s[kt];
            *(lds_f32x4*)(Dt + chunk_off_f(slot, 4 * kt + gq)) = dp[kt];
        }
        wait_lds_barrier();                                      // P, dS of all four query groups are in the images
        BLK_STAMP(6);

        // ================================================= phase B: key slot `slot`
        {
            f32x4v dk[4], dv[4];
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) { dk[ct] = f32x4v{0.f, 0.f, 0.f, 0.f}; dv[ct] = f32x4v{0.f, 0.f, 0.f, 0.f}; }
            // the lane's key column of the [query][key] images: d[qt][r] = dS[query 16 qt + 4g + r][key], p likewise; a query
            // tile whose dS (P) is zero for every key of the wave adds nothing to dK (dV)
            const uint32_t kcol = (uint32_t)(lr & 3) * 4;
            f32x4v d[4], pq[4];
#pragma unroll
            for (int qt = 0; qt < 4; ++qt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const uint32_t off = chunk_off_f(16 * qt + 4 * gq + r, 4 * w + (lr >> 2)) + kcol;
                    d[qt][r] = *(const lds_f32*)(Dt + off);
                    pq[qt][r] = *(const lds_f32*)(Pt + off);
                }
            // dK[key][c] += sum_q dS[q][key] Q[q][c] and dV[key][c] += sum_q P[q][key] dO[q][c] as eight blocks (qt, dK | dV),
            // the column operands of a block read while the block before multiplies
            f32x4v v[2][4];
            col_ops4(v[0], Qt, 0, lr, gq);
#pragma unroll
            for (int qt = 0; qt < 4; ++qt) {
                col_ops4(v[1], Gt, qt, lr, gq);
                if (tile_any(d[qt])) tile_times_col(d[qt], v[0], dk);
                if (qt < 3) col_ops4(v[0], Qt, qt + 1, lr, gq);
                if (tile_any(pq[qt])) tile_times_col(pq[qt], v[1], dv);
            }
            BLK_STAMP(7);
            // the next unit's rows go to the images BEFORE this unit's last stores are issued: waiting for those loads (vmcnt
            // counts in order) then does not wait for the stores, which a wait at the top of the loop would (stamps: 18 % of a unit)
            if (more) {
                wait_lds_barrier();                              // every wave is through with the images
                BLK_STAMP(8);
                put_unit<4>(sm, nx, lane, w);
            }
            store_rows(gtile + g.d, rs, dk, SCALE, gq, n_real);
            store_rows(gtile + 2 * g.d, rs, dv, 1.0f, gq, n_real);
        }
        if (!more) break;
        BLK_STAMP(0);
        advance();
        BLK_STAMP(1);
        wait_lds_barrier();
        BLK_STAMP_NEXT();
    }
}

}  // namespace

int hwgat_launch_blk_fwd_f32(const void* qkv, void* o, const uint32_t* maskbits, int B, int F, int KJ, int nH, int shifted,
                             uint32_t drop_seed, float drop_p, const uint32_t* seed_base, hipStream_t st) {
    const AttnDrop ad = make_drop(drop_seed, drop_p, seed_base);
    BlkGeom g{F, KJ, nH, F / 2, nH * HD, shifted ? 1 : 0};
    const int64_t units = (int64_t)B * g.f * nH;
    if (units > 0x7fffffff) return HWGAT_ESHAPE;
    const int blocks = (int)min(units, (int64_t)256 * 3);        // 48 KB of LDS: three workgroups per CU
    const int skew = lab_env("HWGAT_BLK_SKEW") ? atoi(lab_env("HWGAT_BLK_SKEW")) : 0;
    if (ad.p > 0.f) blk_fwd_f32_k<true><<<blocks, 256, 0, st>>>((const float*)qkv, (float*)o, maskbits, g, (int)units, ad, skew);
    else blk_fwd_f32_k<false><<<blocks, 256, 0, st>>>((const float*)qkv, (float*)o, maskbits, g, (int)units, ad, skew);
    HWGAT_LAUNCH_CHECK();
}

int hwgat_launch_blk_bwd_f32(const void* qkv, const void* dO, void* dqkv, const uint32_t* maskbits, int B, int F, int KJ,
                             int nH, int shifted, uint32_t drop_seed, float drop_p, const uint32_t* seed_base, hipStream_t st) {
    const AttnDrop ad = make_drop(drop_seed, drop_p, seed_base);
    BlkGeom g{F, KJ, nH, F / 2, nH * HD, shifted ? 1 : 0};
    const int64_t units = (int64_t)B * g.f * nH;
    if (units > 0x7fffffff) return HWGAT_ESHAPE;
    const int blocks = (int)min(units, (int64_t)256 * 2);        // 64 KB of LDS: two workgroups per CU
    const int skew = lab_env("HWGAT_BLK_SKEW") ? atoi(lab_env("HWGAT_BLK_SKEW")) : 0;
    if (ad.p > 0.f)
        blk_bwd_f32_k<true><<<blocks, 256, 0, st>>>((const float*)qkv, (const float*)dO, (float*)dqkv, maskbits, g, (int)units, ad, skew);
    else
        blk_bwd_f32_k<false><<<blocks, 256, 0, st>>>((const float*)qkv, (const float*)dO, (float*)dqkv, maskbits, g, (int)units, ad, skew);
    HWGAT_LAUNCH_CHECK();
}

#ifdef HWGAT_LAB
extern "C" int hwgat_lab_blk_stamps(void* buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_blk_stamps), &buf, sizeof(buf)) == hipSuccess ? 0 : HWGAT_EINVAL;
}
#endif
