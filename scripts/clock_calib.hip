// Calibrates the cycle counter used by the phase stamps: counter ticks per second (against the 100 MHz
// real-time counter) and shader cycles per tick (a dependent chain of 4-cycle VALU adds, and of MFMAs).
// build: hipcc --offload-arch=gfx950 -O3 -o scripts/clock_calib scripts/clock_calib.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

__global__ void calib(unsigned long long* out, float seed, int n, int heavy) {
    float v = seed;
    f32x16 acc; for (int e = 0; e < 16; ++e) acc[e] = seed;
    bf16x8 a, b; for (int e = 0; e < 8; ++e) { a[e] = (__bf16)seed; b[e] = (__bf16)seed; }
    const unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    if (!heavy) {
        for (int i = 0; i < n; ++i) {
#pragma unroll
            for (int k = 0; k < 64; ++k) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v) : "v"(seed));
        }
    } else {
        for (int i = 0; i < n; ++i) {
#pragma unroll
            for (int k = 0; k < 16; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = w1 - w0; }
    if (v == 12345.f || acc[3] == 12345.f) out[2] = 1;
}

int main() {
    unsigned long long* d; hipMalloc(&d, 64);
    unsigned long long h[3];
    for (int heavy = 0; heavy < 2; ++heavy)
        for (int blocks : {1, 1024}) {
            const int n = 20000;
            hipLaunchKernelGGL(calib, dim3(blocks), dim3(heavy ? 256 : 64), 0, 0, d, 1.0f, n, heavy);
            hipDeviceSynchronize();
            hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
            const double ops = heavy ? 16.0 * n : 64.0 * n;
            printf("%s chain, %4d blocks: %llu ticks, %llu x10ns -> counter %.1f MHz; %.2f ticks per op, %.2f ns per op\n",
                   heavy ? "mfma32x32x16" : "v_add_f32  ", blocks, h[0], h[1], h[0] / (h[1] * 0.01), h[0] / ops, h[1] * 10.0 / ops);
        }
    return 0;
}
