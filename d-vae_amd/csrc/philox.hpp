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

// Streams whose id has bit 31 set draw EIGHT normals per Philox call, from 16-bit uniforms: u1 = (lo16 + 1) / 65536 in
// (0, 1], u2 = hi16 / 65536 in [0, 1) of each of the four output words.  The 32-bit multiplies of the ten Philox rounds
// are quarter rate and make up most of a draw, so this halves the cost per normal; the price is a coarser grid (|eps| <=
// 4.71, 65 536 radii and angles), which the bf16 inference path -- where eps feeds a product that is rounded to 8
// significant bits -- does not see.  Training and the fp32 path keep 24-bit uniforms (bit 31 clear).
constexpr uint32_t PHILOX_STREAM_16 = 0x80000000u;

__device__ __forceinline__ void box_muller(float u1, float u2, float& n0, float& n1) {
    // on the hardware transcendental units: v_log_f32 (log2), v_sin/v_cos (argument in turns); sqrt(-2 ln u1) by
    // v_sqrt_f32 itself (1 ulp; the argument is in [0, 34), no denormal or range fix-ups needed)
    const float rad = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));
    n0 = rad * __builtin_amdgcn_cosf(u2);
    n1 = rad * __builtin_amdgcn_sinf(u2);
}

// eight standard normals for elements [8 o, 8 o + 8) of image img (16-bit streams)
__device__ inline void philox_normal8(const PhiloxKey& k, uint64_t img, uint32_t o, float (&n)[8]) {
    uint32_t r[4];
    philox4x32_10(o, (uint32_t)img, (uint32_t)(img >> 32), k.stream_id, k.seed_lo, k.seed_hi, r);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float u1 = ((float)(r[j] & 0xffffu) + 1.0f) * (1.0f / 65536.0f);
        const float u2 = (float)(r[j] >> 16) * (1.0f / 65536.0f);
        box_muller(u1, u2, n[2 * j], n[2 * j + 1]);
    }
}

// four standard normals for quad q (elements [4 q, 4 q + 4)) of image img
__device__ inline f32x4 philox_normal4(const PhiloxKey& k, uint64_t img, uint32_t q) {
    if (k.stream_id & PHILOX_STREAM_16) {                 // (callers that can use all eight call philox_normal8)
        float n8[8];
        philox_normal8(k, img, q >> 1, n8);
        const int o = (q & 1u) * 4;
        f32x4 v = {n8[o], n8[o + 1], n8[o + 2], n8[o + 3]};
        return v;
    }
    uint32_t r[4];
    philox4x32_10(q, (uint32_t)img, (uint32_t)(img >> 32), k.stream_id, k.seed_lo, k.seed_hi, r);
    float n[4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const float u1 = ((float)(r[2 * h] >> 8) + 1.0f) * (1.0f / 16777216.0f);     // (0, 1]
        const float u2 = (float)(r[2 * h + 1] >> 8) * (1.0f / 16777216.0f);          // [0, 1)
        box_muller(u1, u2, n[2 * h], n[2 * h + 1]);
    }
    f32x4 v = {n[0], n[1], n[2], n[3]};
    return v;
}

// The noise of an MFMA accumulator's lane: lane (pixel r, half h = lane >> 5) holds, for g = 0..3, the four elements
// base + 8 g + 4 h .. of its pixel (base a multiple of 8), so the two halves of a pixel share every group of eight.  On a
// 16-bit stream each half draws two of the four groups whole and the halves trade the parts they do not own (eight
// v_permlane32_swap): two Philox calls per lane instead of four.
__device__ inline void philox_lane_quads(const PhiloxKey& k, uint64_t img, uint32_t base, int h, f32x4 (&e)[4]) {
    if (k.stream_id & PHILOX_STREAM_16) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float n8[8];
            philox_normal8(k, img, (base >> 3) + 2u * (unsigned)h + (unsigned)j, n8);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                // x: elements 0-3 of the group, y: elements 4-7.  The swap exchanges the upper lanes of x with the lower
                // lanes of y: every lane then holds in x its part of group j and in y its part of group 2 + j.
                const auto sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, n8[i]),
                                                                 __builtin_bit_cast(unsigned, n8[i + 4]), false, false);
                e[j][i] = __builtin_bit_cast(float, (unsigned)sw[0]);
                e[2 + j][i] = __builtin_bit_cast(float, (unsigned)sw[1]);
            }
        }
        return;
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) e[g] = philox_normal4(k, img, (base + 8u * g + 4u * h) >> 2);
}

}  // namespace dvae
