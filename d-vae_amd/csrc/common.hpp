// Shared helpers for the gfx950 kernels of libdvae_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <atomic>

#include "../../include/dvae_hip.h"

namespace dvae {

using bf16 = __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned int;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;

extern thread_local char g_err[512];

#define DVAE_FAIL(code, ...)                          \
    do {                                              \
        snprintf(dvae::g_err, sizeof(dvae::g_err), __VA_ARGS__); \
        return (code);                                \
    } while (0)

#define DVAE_CHECK_LAUNCH(what)                                                            \
    do {                                                                                   \
        hipError_t e__ = hipGetLastError();                                                \
        if (e__ != hipSuccess) DVAE_FAIL(DVAE_ELAUNCH, "%s: %s", what, hipGetErrorString(e__)); \
    } while (0)

static inline int ilog2_exact(int v) {
    if (v <= 0 || (v & (v - 1))) return -1;
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}
static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Elements per 16-byte vector and register <-> float conversion for the two activation dtypes.
template <typename T> struct Act;
template <> struct Act<float> {
    static constexpr int VEC = 4;
    __device__ static inline void load(const float* p, float (&v)[4]) {
        f32x4 r = *reinterpret_cast<const f32x4*>(p);
        v[0] = r[0]; v[1] = r[1]; v[2] = r[2]; v[3] = r[3];
    }
    __device__ static inline void store(float* p, const float (&v)[4]) {
        f32x4 r = {v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(p) = r;
    }
    // 4 consecutive elements (one RGB_ pixel)
    __device__ static inline void load4(const float* p, float (&v)[4]) { load(p, v); }
    __device__ static inline void store4(float* p, const float (&v)[4]) { store(p, v); }
};
template <> struct Act<bf16> {
    static constexpr int VEC = 8;
    __device__ static inline void load(const bf16* p, float (&v)[8]) {
        bf16x8 r = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (float)r[i];
    }
    __device__ static inline void store(bf16* p, const float (&v)[8]) {
        bf16x8 r;
#pragma unroll
        for (int i = 0; i < 8; ++i) r[i] = (bf16)v[i];
        *reinterpret_cast<bf16x8*>(p) = r;
    }
    __device__ static inline void load4(const bf16* p, float (&v)[4]) {
        bf16x4 r = *reinterpret_cast<const bf16x4*>(p);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = (float)r[i];
    }
    __device__ static inline void store4(bf16* p, const float (&v)[4]) {
        bf16x4 r;
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = (bf16)v[i];
        *reinterpret_cast<bf16x4*>(p) = r;
    }
};

__device__ inline float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Sum over a 256-thread block; result valid in every thread.  `red` is >= 4 floats of LDS.
__device__ inline float block_sum_256(float v, float* red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// Opt-in of a kernel to more than 64 KB of dynamic LDS.  The attribute belongs to the (kernel, device) pair and a
// process may drive several devices, so each launch site keeps one bit per device ordinal.
struct LdsOptIn {
    std::atomic<unsigned long long> done{0ull};
    void ensure(const void* fn, int bytes) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        const unsigned long long bit = 1ull << (dev & 63);
        if (done.load(std::memory_order_acquire) & bit) return;
        (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);   // a failure shows up at the launch check
        done.fetch_or(bit, std::memory_order_release);
    }
};

}  // namespace dvae
