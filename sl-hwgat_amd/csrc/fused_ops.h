// Elementwise helpers shared by the fused linear kernels (fp32 and bf16 families).
#pragma once
#include "common.h"

enum { PRO_NONE = 0, PRO_LN = 1, PRO_DROP = 2 };
enum { EPI_BIAS = 0, EPI_BIAS_DROP_RES = 1, EPI_BIAS_GELU_DROP = 2, EPI_GELU_BWD = 3, EPI_NONE = 4 };

// Dropout mask: a counter-based hash of (seed, element index).  One 32-bit hash serves an
// aligned PAIR of elements (16 bits each): keep iff its 16 bits >= p * 65536 (p is thereby
// quantised to 1/65536); survivors are scaled by 1/(1-p).
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

__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.7071067811865476f)); }
__device__ __forceinline__ float gelu_grad(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.7071067811865476f));
    return cdf + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}

