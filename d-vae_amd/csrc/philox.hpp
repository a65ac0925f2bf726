// Counter-based standard normals shared by the noise, reparameterisation and fused encoder-head kernels.
#pragma once
#include "common.hpp"

namespace dvae {

// ------------------------------------------------------------------------------------------------
// Philox4x32-10 standard normals, keyed by (seed, stream) and counted by (image index, quad)
// ------------------------------------------------------------------------------------------------
__device__ inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                     uint32_t (&o)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}

struct PhiloxKey {
    const int32_t* index;   // per-image global index, or NULL: first + b
    long first;
    uint32_t seed_lo, seed_hi, stream_id;
};

// four standard normals for quad q of image img
__device__ inline f32x4 philox_normal4(const PhiloxKey& k, uint64_t img, uint32_t q) {
    uint32_t r[4];
    philox4x32_10(q, (uint32_t)img, (uint32_t)(img >> 32), k.stream_id, k.seed_lo, k.seed_hi, r);
    float n[4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        // Box-Muller on the hardware transcendental units: v_log_f32 (log2), v_sin/v_cos (argument in turns)
        const float u1 = ((float)(r[2 * h] >> 8) + 1.0f) * (1.0f / 16777216.0f);     // (0, 1]
        const float u2 = (float)(r[2 * h + 1] >> 8) * (1.0f / 16777216.0f);          // [0, 1)
        // sqrt(-2 ln u1): v_sqrt_f32 itself (1 ulp; the argument is in [0, 34), no denormal or range fix-ups needed)
        const float rad = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));
        n[2 * h] = rad * __builtin_amdgcn_cosf(u2);
        n[2 * h + 1] = rad * __builtin_amdgcn_sinf(u2);
    }
    f32x4 v = {n[0], n[1], n[2], n[3]};
    return v;
}

}  // namespace dvae
