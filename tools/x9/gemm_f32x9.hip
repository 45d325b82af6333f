// EXPERIMENT (opt-in, not on the default path): fp32 NT GEMM whose products are formed as nine exact
// bf16 x bf16 partial products on v_mfma_f32_32x32x16_bf16 with fp32 accumulation.
//
//   x = x1 + x2 + x3 exactly, x_i = the i-th group of 8 significand bits of x kept as a bf16 (split by
//   truncation: x1 = top 16 bits of x, x2 = top 16 bits of x - x1, x3 = x - x1 - x2, which has at most
//   8 significant bits left).  a * w = sum_{i,j} a_i w_j with every a_i w_j exact in fp32 (8 x 8 bits),
//   so the result differs from an fp32 FMA chain only by the order / rounding of the accumulation.
//
// Why: the 32x32x16 bf16 MFMA takes 32 cycles, so the nine of them that replace the eight 64-cycle
// v_mfma_f32_32x32x2_f32 of the same 32x32x16 block cost 0.56x the matrix-pipe time.  DESIGN.md section 8
// discusses what this would mean for the headline; this file exists to MEASURE it (tools/x9_one.py,
// tests/test_gpu_gemm.py::test_x9_*).  Only the plain product (no prologue, no epilogue) is built.
//
//   C[M,N] = A[M,K] . W[N,K]^T,  A fp32 (split on the fly while staging to LDS),
//   W given as three bf16 planes [3][N][K] (hwgat_split3_bf16, once per weight).
//   128x128 tile, 4 waves x (64x64), K slabs of 16, three operand planes per side, double-buffered in LDS.
#include <stdlib.h>
#include <type_traits>
#include "common.h"

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

constexpr int BM = 128, BN = 128, BK = 32, LDT = 40;            // LDS rows of 32 bf16 + 8 pad (80 B: conflict-free b128 reads)
constexpr int PLANE = (BM + BN) * LDT;                          // bf16 elements per plane: A rows then W rows

// three truncation pieces of two consecutive floats, each packed as a bf16 pair (low half = first element)
__device__ __forceinline__ void split2(float x, float y, uint32_t (&p)[3]) {
    const uint32_t ux = __float_as_uint(x), uy = __float_as_uint(y);
    const float x1 = __uint_as_float(ux & 0xffff0000u), y1 = __uint_as_float(uy & 0xffff0000u);
    const float rx = x - x1, ry = y - y1;                       // exact
    const uint32_t urx = __float_as_uint(rx), ury = __float_as_uint(ry);
    const float x2 = __uint_as_float(urx & 0xffff0000u), y2 = __uint_as_float(ury & 0xffff0000u);
    const float sx = rx - x2, sy = ry - y2;                     // exact, <= 8 significant bits
    p[0] = (ux >> 16) | (uy & 0xffff0000u);
    p[1] = (urx >> 16) | (ury & 0xffff0000u);
    p[2] = (__float_as_uint(sx) >> 16) | (__float_as_uint(sy) & 0xffff0000u);
}

// K slabs of 16, double-buffered in LDS (3 planes x 256 rows x 48 B x 2 = 72 KB, 2 blocks/CU): one barrier per
// slab, the next slab's split + LDS writes overlap other waves' MFMAs.  (The first version -- slabs of 32,
// single buffer, two barriers -- ran at 0.49 MFMA utilisation.)
constexpr int BK2 = 16, LDT2 = 24, PLANE2 = (BM + BN) * LDT2, BUF2 = 3 * PLANE2;

__global__ __launch_bounds__(256, 2) void gemm_nt_x9_k(const float* __restrict__ A, const uint16_t* __restrict__ Wp,
                                                       float* __restrict__ C, int64_t M, int N, int K) {
    __shared__ __attribute__((aligned(16))) uint16_t sm[2 * BUF2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane & 31, hh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = N / BN;
    const int64_t m0 = (int64_t)(blockIdx.x / tiles_n) * BM;
    const int n0 = (blockIdx.x % tiles_n) * BN;
    const int n_slab = K / BK2;
    const int64_t plane_w = (int64_t)N * K;

    // staging roles: A: rows arow + 64 i, floats ac4..ac4+3;  W: 16-byte chunks idx = tid + 256 q of [3][128][2]
    const int arow = tid >> 2, ac4 = (tid & 3) * 4;
    f32x4 ra[2];
    u32x4 rw[3];
    auto issue = [&](int s) {
        const int k0 = s * BK2;
#pragma unroll
        for (int i = 0; i < 2; ++i) ra[i] = *reinterpret_cast<const f32x4*>(A + (m0 + arow + 64 * i) * K + k0 + ac4);
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int idx = tid + 256 * q, pl = idx >> 8, rem = idx & 255, row = rem >> 1, ch = rem & 1;
            rw[q] = *reinterpret_cast<const u32x4*>(Wp + pl * plane_w + (int64_t)(n0 + row) * K + k0 + ch * 8);
        }
    };
    auto commit = [&](int buf) {
        uint16_t* b = sm + buf * BUF2;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            uint32_t lo[3], hi[3];
            split2(ra[i].x, ra[i].y, lo);
            split2(ra[i].z, ra[i].w, hi);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                u32x2 v = {lo[pl], hi[pl]};
                *reinterpret_cast<u32x2*>(b + pl * PLANE2 + (arow + 64 * i) * LDT2 + ac4) = v;
            }
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int idx = tid + 256 * q, pl = idx >> 8, rem = idx & 255, row = rem >> 1, ch = rem & 1;
            *reinterpret_cast<u32x4*>(b + pl * PLANE2 + (BM + row) * LDT2 + ch * 8) = rw[q];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    issue(0);
    commit(0);
    __syncthreads();
    int buf = 0;
    for (int s = 0; s < n_slab; ++s) {
        const bool have_next = s + 1 < n_slab;
        if (have_next) issue(s + 1);
        const uint16_t* ap = sm + buf * BUF2 + (wm * 64 + lq) * LDT2 + 8 * hh;
        const uint16_t* wp = sm + buf * BUF2 + (BM + wn * 64 + lq) * LDT2 + 8 * hh;
        bf16x8 af[3][2], bfr[3][2];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                af[pl][i] = *reinterpret_cast<const bf16x8*>(ap + pl * PLANE2 + i * 32 * LDT2);
                bfr[pl][i] = *reinterpret_cast<const bf16x8*>(wp + pl * PLANE2 + i * 32 * LDT2);
            }
        // smallest terms first: (pa + pb) descending = 4, 3, 2, 1, 0
#pragma unroll
        for (int sum = 4; sum >= 0; --sum)
#pragma unroll
            for (int pa = 0; pa < 3; ++pa) {
                const int pb = sum - pa;
                if (pb < 0 || pb > 2) continue;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[pa][i], bfr[pb][j], acc[i][j], 0, 0, 0);
            }
        if (have_next) commit(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    // lane (n = lq, hh), reg r -> C[m = crow(r,hh)][n]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                C[(m0 + wm * 64 + i * 32 + crow(r, hh)) * N + n0 + wn * 64 + j * 32 + lq] = acc[i][j][r];
}

// 256x256 tile, 4 waves x (128x128 = 4x4 MFMA tiles, 256 accumulator registers), ONE wave per SIMD, K slabs of 16
// double-buffered (3 planes x 512 rows x 48 B x 2 = 144 KB): 52 flop per staged byte instead of 26 -- the 128x128
// form is bound by L2->LDS operand traffic, not by the MFMAs.
constexpr int BM3 = 256, BN3 = 256, PLANE3 = (BM3 + BN3) * LDT2, BUF3 = 3 * PLANE3;

// LLVM SchedGroupMask bits
constexpr int SG_VALU = 0x002, SG_MFMA = 0x008, SG_VMEM_RD = 0x020, SG_DS_RD = 0x100, SG_DS_WR = 0x200;

// Slab s+2 is requested at the top of slab s into one of TWO register sets and written to LDS during slab s+1
// (a load has a whole slab of 144 MFMAs to land); its staging is cut into 12 small steps (one half-row split,
// ~11 vector instructions, or two LDS writes), one behind every 4th of the last 48 MFMAs, fenced so that it
// stays there: an MFMA holds the vector issue port for 8 of its 32 cycles, the rest of the gap is free.
// PIN = 0 keeps the compiler's own order (staging after the MFMAs) for A/B runs.
template <int PIN>
__global__ __launch_bounds__(256, 1) void gemm_nt_x9_256_k(const float* __restrict__ A, const uint16_t* __restrict__ Wp,
                                                           float* __restrict__ C, int64_t M, int N, int K) {
    __shared__ __attribute__((aligned(16))) uint16_t sm[2 * BUF3];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane & 31, hh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = N / BN3;
    const int64_t m0 = (int64_t)(blockIdx.x / tiles_n) * BM3;
    const int n0 = (blockIdx.x % tiles_n) * BN3;
    const int n_slab = K / BK2;                                  // even (K % 32 == 0)
    const int64_t plane_w = (int64_t)N * K;

    // staging roles: A: rows arow + 64 i (i < 4), floats ac4..ac4+3;  W: 16-byte chunks idx = tid + 256 q of [3][256][2]
    const int arow = tid >> 2, ac4 = (tid & 3) * 4;
    f32x4 ra0[4], ra1[4];
    u32x4 rw0[6], rw1[6];
    auto issue = [&](int s, f32x4 (&ra)[4], u32x4 (&rw)[6]) {
        const int k0 = s * BK2;
#pragma unroll
        for (int i = 0; i < 4; ++i) ra[i] = *reinterpret_cast<const f32x4*>(A + (m0 + arow + 64 * i) * K + k0 + ac4);
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const int idx = tid + 256 * q, pl = idx >> 9, rem = idx & 511, row = rem >> 1, ch = rem & 1;
            rw[q] = *reinterpret_cast<const u32x4*>(Wp + pl * plane_w + (int64_t)(n0 + row) * K + k0 + ch * 8);
        }
    };
    // step k of 12 of writing one slab (registers ra, rw) into LDS buffer nb
    uint32_t plo[3], phi[3];
    auto commit_step = [&](int k, uint16_t* nb, const f32x4 (&ra)[4], const u32x4 (&rw)[6]) {
        if (k < 8) {
            const int i = k >> 1;
            if ((k & 1) == 0) {
                split2(ra[i].x, ra[i].y, plo);
            } else {
                split2(ra[i].z, ra[i].w, phi);
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    u32x2 v = {plo[pl], phi[pl]};
                    *reinterpret_cast<u32x2*>(nb + pl * PLANE3 + (arow + 64 * i) * LDT2 + ac4) = v;
                }
            }
        } else {
            const int q0 = (k - 8) * 2;
#pragma unroll
            for (int q = q0; q < q0 + 2 && q < 6; ++q) {
                const int idx = tid + 256 * q, pl = idx >> 9, rem = idx & 511, row = rem >> 1, ch = rem & 1;
                *reinterpret_cast<u32x4*>(nb + pl * PLANE3 + (BM3 + row) * LDT2 + ch * 8) = rw[q];
            }
        }
    };

    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // slab s computes from LDS buffer s & 1; `cur` = register set that holds slab s+1 (staged during this slab),
    // `nxt` = the set slab s+2 is loaded into (it held slab s, already staged).  After the last slab the staging
    // writes stale registers into the buffer nobody reads again.
    auto slab = [&](int s, f32x4 (&cur_a)[4], u32x4 (&cur_w)[6], f32x4 (&nxt_a)[4], u32x4 (&nxt_w)[6]) {
        const int buf = s & 1;
        if (s + 2 < n_slab) issue(s + 2, nxt_a, nxt_w);
        __builtin_amdgcn_sched_barrier(0);                                          // the loads stay at the top of the slab
        const uint16_t* ap = sm + buf * BUF3 + (wm * 128 + lq) * LDT2 + 8 * hh;
        const uint16_t* wp = sm + buf * BUF3 + (BM3 + wn * 128 + lq) * LDT2 + 8 * hh;
        uint16_t* nb = sm + (buf ^ 1) * BUF3;
        // 144 MFMAs in three passes over the A planes (W fragments of all three planes resident: 48 registers)
        bf16x8 bfr[3][4], af[3][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) af[2][i] = *reinterpret_cast<const bf16x8*>(ap + 2 * PLANE3 + i * 32 * LDT2);
#pragma unroll
        for (int pl = 2; pl >= 0; --pl)
#pragma unroll
            for (int i = 0; i < 4; ++i) bfr[pl][i] = *reinterpret_cast<const bf16x8*>(wp + pl * PLANE3 + i * 32 * LDT2);
#pragma unroll
        for (int pl = 1; pl >= 0; --pl)
#pragma unroll
            for (int i = 0; i < 4; ++i) af[pl][i] = *reinterpret_cast<const bf16x8*>(ap + pl * PLANE3 + i * 32 * LDT2);
#pragma unroll
        for (int pa = 2; pa >= 0; --pa)
#pragma unroll
            for (int pb = 2; pb >= 0; --pb)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[pa][i], bfr[pb][j], acc[i][j], 0, 0, 0);
                        if constexpr (PIN == 1) {
                            const int idx = ((2 - pa) * 3 + (2 - pb)) * 16 + i * 4 + j;
                            if (idx >= 96 && (idx & 3) == 3) {
                                __builtin_amdgcn_sched_barrier(0);
                                commit_step((idx - 96) >> 2, nb, cur_a, cur_w);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                    }
        if constexpr (PIN == 0) {
#pragma unroll
            for (int k = 0; k < 12; ++k) commit_step(k, nb, cur_a, cur_w);
        }
        __syncthreads();
    };

    issue(0, ra0, rw0);
#pragma unroll
    for (int k = 0; k < 12; ++k) commit_step(k, sm, ra0, rw0);
    issue(1, ra1, rw1);
    __syncthreads();
    for (int s = 0; s < n_slab; s += 2) {
        slab(s, ra1, rw1, ra0, rw0);             // stages slab s+1 (set 1), loads slab s+2 into set 0
        slab(s + 1, ra0, rw0, ra1, rw1);         // stages slab s+2 (set 0), loads slab s+3 into set 1
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                C[(m0 + wm * 128 + i * 32 + crow(r, hh)) * N + n0 + wn * 128 + j * 32 + lq] = acc[i][j][r];
}

// the same 256x256 tile with EIGHT waves (2 x 4, wave tile 128x64, two waves per SIMD): one wave's barrier waits and
// staging instructions overlap the other's MFMAs
__global__ __launch_bounds__(512, 2) void gemm_nt_x9_256w8_k(const float* __restrict__ A, const uint16_t* __restrict__ Wp,
                                                             float* __restrict__ C, int64_t M, int N, int K) {
    __shared__ __attribute__((aligned(16))) uint16_t sm[2 * BUF3];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane & 31, hh = lane >> 5;
    const int wm = wave >> 2, wn = wave & 3;
    const int tiles_n = N / BN3;
    const int64_t m0 = (int64_t)(blockIdx.x / tiles_n) * BM3;
    const int n0 = (blockIdx.x % tiles_n) * BN3;
    const int n_slab = K / BK2;
    const int64_t plane_w = (int64_t)N * K;
    // staging roles: A: rows arow + 128 i (i < 2), floats ac4..ac4+3;  W: 16-byte chunks idx = tid + 512 q of [3][256][2]
    const int arow = tid >> 2, ac4 = (tid & 3) * 4;
    f32x4 ra[2];
    u32x4 rw[3];
    auto issue = [&](int s) {
        const int k0 = s * BK2;
#pragma unroll
        for (int i = 0; i < 2; ++i) ra[i] = *reinterpret_cast<const f32x4*>(A + (m0 + arow + 128 * i) * K + k0 + ac4);
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int idx = tid + 512 * q, pl = idx >> 9, rem = idx & 511, row = rem >> 1, ch = rem & 1;
            rw[q] = *reinterpret_cast<const u32x4*>(Wp + pl * plane_w + (int64_t)(n0 + row) * K + k0 + ch * 8);
        }
    };
    auto commit = [&](int buf) {
        uint16_t* b = sm + buf * BUF3;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            uint32_t lo[3], hi[3];
            split2(ra[i].x, ra[i].y, lo);
            split2(ra[i].z, ra[i].w, hi);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                u32x2 v = {lo[pl], hi[pl]};
                *reinterpret_cast<u32x2*>(b + pl * PLANE3 + (arow + 128 * i) * LDT2 + ac4) = v;
            }
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int idx = tid + 512 * q, pl = idx >> 9, rem = idx & 511, row = rem >> 1, ch = rem & 1;
            *reinterpret_cast<u32x4*>(b + pl * PLANE3 + (BM3 + row) * LDT2 + ch * 8) = rw[q];
        }
    };
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    issue(0);
    commit(0);
    __syncthreads();
    int buf = 0;
    for (int s = 0; s < n_slab; ++s) {
        const bool have_next = s + 1 < n_slab;
        if (have_next) issue(s + 1);
        const uint16_t* ap = sm + buf * BUF3 + (wm * 128 + lq) * LDT2 + 8 * hh;
        const uint16_t* wp = sm + buf * BUF3 + (BM3 + wn * 64 + lq) * LDT2 + 8 * hh;
        bf16x8 bfr[3][2], af[4];
#pragma unroll
        for (int pl = 2; pl >= 0; --pl)
#pragma unroll
            for (int j = 0; j < 2; ++j) bfr[pl][j] = *reinterpret_cast<const bf16x8*>(wp + pl * PLANE3 + j * 32 * LDT2);
#pragma unroll
        for (int pa = 2; pa >= 0; --pa) {
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const bf16x8*>(ap + pa * PLANE3 + i * 32 * LDT2);
#pragma unroll
            for (int pb = 2; pb >= 0; --pb)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[pb][j], acc[i][j], 0, 0, 0);
        }
        if (have_next) commit(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                C[(m0 + wm * 128 + i * 32 + crow(r, hh)) * N + n0 + wn * 64 + j * 32 + lq] = acc[i][j][r];
}

__global__ void split3_k(const float* __restrict__ in, uint16_t* __restrict__ out, int64_t n) {
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 2; i < n; i += (int64_t)gridDim.x * 512) {
        uint32_t p[3];
        split2(in[i], in[i + 1], p);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<uint32_t*>(out + pl * n + i) = p[pl];
    }
}

}  // namespace

extern "C" int hwgat_split3_bf16(const float* in, uint16_t* out, int64_t n, void* stream) {
    if (!in || !out || n <= 0 || (n & 1)) return HWGAT_EINVAL;
    const int64_t want = (n / 2 + 255) / 256;
    split3_k<<<(int)(want < 2048 ? want : 2048), 256, 0, (hipStream_t)stream>>>(in, out, n);
    HWGAT_LAUNCH_CHECK();
}

extern "C" int hwgat_linear_nt_f32x9(const float* A, const uint16_t* W3, float* C, int64_t M, int N, int K,
                                     void* stream) {
    if (!A || !W3 || !C || M <= 0 || N <= 0 || K <= 0) return HWGAT_EINVAL;
    if (M % BM || N % BN || K % 32 || (M / BM) * (int64_t)(N / BN) > 0x7fffffff) return HWGAT_ESHAPE;
    static const bool small_only = getenv("HWGAT_X9_TILE") && getenv("HWGAT_X9_TILE")[0] == 's';
    static const int pin = getenv("HWGAT_X9_PIN") ? atoi(getenv("HWGAT_X9_PIN")) : 1;   // 0: compiler order (A/B)
    if (M % BM3 == 0 && N % BN3 == 0 && !small_only) {
        const int grid = (int)((M / BM3) * (N / BN3));
        if (pin == 8) gemm_nt_x9_256w8_k<<<grid, 512, 0, (hipStream_t)stream>>>(A, W3, C, M, N, K);
        else if (pin == 0) gemm_nt_x9_256_k<0><<<grid, 256, 0, (hipStream_t)stream>>>(A, W3, C, M, N, K);
        else gemm_nt_x9_256_k<1><<<grid, 256, 0, (hipStream_t)stream>>>(A, W3, C, M, N, K);
    }
    else
        gemm_nt_x9_k<<<(int)((M / BM) * (N / BN)), 256, 0, (hipStream_t)stream>>>(A, W3, C, M, N, K);
    HWGAT_LAUNCH_CHECK();
}
