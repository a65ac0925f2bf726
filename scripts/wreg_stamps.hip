// Phase timing of the weights-in-registers convolution (layer1 shape: 3x3, 64->64, 32x32, 1024 images).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -pragma-unroll-threshold=1000000 -o scripts/wreg_stamps scripts/wreg_stamps.hip
// Prints, per phase, the mean cycle count over workgroups and tiles (wave 0's view), and the kernel time without stamps
// (the stamped launch is a separate one: stamps cost cycles themselves).
#define WREG_STAMPS 1
#include "../d-vae_amd/csrc/conv_wreg.hpp"
#include <vector>
#include <stdlib.h>

namespace dvae { thread_local char g_err[512]; }
using namespace dvae;

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 1024;
    const int with_res = argc > 2 ? atoi(argv[2]) : 0;
    const int cin = argc > 3 ? atoi(argv[3]) : 64, cout = argc > 4 ? atoi(argv[4]) : 64;
    dvae_conv_desc d = {B, 32, 32, cin, cout, 3, 3, 1, 1, 0};
    GemmPlan P;
    if (build_plan(d, DVAE_OP_FWD, DVAE_BF16, P)) { printf("plan failed: %s\n", g_err); return 1; }
    const size_t nin = (size_t)B * 32 * 32 * cin, nout = (size_t)B * 32 * 32 * cout;
    std::vector<unsigned short> hx(nin > nout ? nin : nout), hw((size_t)P.Npad * P.Ktot);
    for (auto& v : hx) v = ((rand() & 1) ? 0xbc00 : 0x3c00) + (rand() & 0xff);      // random signs and mantissas
    for (auto& v : hw) v = ((rand() & 1) ? 0xb800 : 0x3800) + (rand() & 0xff);
    void *x, *w, *y, *r; unsigned long long* st;
    (void)hipMalloc(&x, nin * 2); (void)hipMalloc(&y, nout * 2); (void)hipMalloc(&r, nout * 2); (void)hipMalloc(&w, hw.size() * 2);
    hipMalloc(&st, 256 * 64 * 8 * 8);
    hipMemcpy(x, hx.data(), nin * 2, hipMemcpyHostToDevice);
    hipMemcpy(r, hx.data(), nout * 2, hipMemcpyHostToDevice);
    hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    hipMemset(st, 0, 256 * 64 * 8 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep)
        if (!try_launch_wreg_bf16(P, x, w, y, nullptr, nullptr, with_res ? r : nullptr, 1, 0, 0, 0)) { printf("not applicable\n"); return 1; }
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int rep = 0; rep < 20; ++rep) try_launch_wreg_bf16(P, x, w, y, nullptr, nullptr, with_res ? r : nullptr, 1, 0, 0, 0);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("B=%d res=%d %d->%d: %.1f us/launch without stamps\n", B, with_res, cin, cout, ms * 50.f);
    try_launch_wreg_bf16(P, x, w, y, nullptr, nullptr, with_res ? r : nullptr, 1, 0, 0, 0, st);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
    std::vector<unsigned long long> hs(256 * 64 * 8);
    hipMemcpy(hs.data(), st, hs.size() * 8, hipMemcpyDeviceToHost);
    const char* names[6] = {"tile start -> step 1", "-> mid block 0", "-> end block 0", "-> mid block 1", "-> end block 1", "barrier"};
    const int tiles_per_wg = B * 4 / 256;
    double sum[7] = {0}; long cnt = 0;
    for (int g = 0; g < 256; ++g)
        for (int it = 1; it < tiles_per_wg && it < 64; ++it) {
            const unsigned long long* s = &hs[((size_t)g * 64 + it) * 8];
            for (int k = 0; k < 6; ++k) sum[k] += (double)(s[k + 1] - s[k]);
            sum[6] += (double)(s[6] - s[0]);
            ++cnt;
        }
    for (int k = 0; k < 6; ++k) printf("  %-24s %8.0f cycles\n", names[k], sum[k] / cnt);
    printf("  %-24s %8.0f cycles per tile (%ld samples)\n", "total", sum[6] / cnt, cnt);
    const unsigned long long* s0 = &hs[0];
    const int lastt = (tiles_per_wg < 64 ? tiles_per_wg : 64) - 1;
    printf("  wg0: first tile %llu cycles, tile-loop total %llu cycles over %d tiles\n", s0[6] - s0[0],
           hs[((size_t)0 * 64 + lastt) * 8 + 6] - s0[0], lastt + 1);
    // life of the workgroups on the constant 100 MHz clock (tile slot 63), relative to the earliest entry
    if (tiles_per_wg < 63) {
        unsigned long long t0 = ~0ull;
        for (int g = 0; g < 256; ++g) { const unsigned long long v = hs[((size_t)g * 64 + 63) * 8]; if (v && v < t0) t0 = v; }
        double mn[4] = {1e30, 1e30, 1e30, 1e30}, mx[4] = {0, 0, 0, 0}, av[4] = {0, 0, 0, 0};
        for (int g = 0; g < 256; ++g)
            for (int k = 0; k < 4; ++k) {
                const double us = (double)(hs[((size_t)g * 64 + 63) * 8 + k] - t0) * 0.01;
                if (us < mn[k]) mn[k] = us;
                if (us > mx[k]) mx[k] = us;
                av[k] += us / 256;
            }
        const char* rn[4] = {"entry", "weights in registers", "tile loop done", "exit"};
        for (int k = 0; k < 4; ++k) printf("  %-22s min %7.2f  mean %7.2f  max %7.2f us after the first entry\n", rn[k], mn[k], av[k], mx[k]);
        // per XCD (workgroup g runs on XCD g % 8): when each finishes
        for (int x = 0; x < 8; ++x) {
            double a = 0, m = 0;
            for (int g = x; g < 256; g += 8) { const double us = (double)(hs[((size_t)g * 64 + 63) * 8 + 3] - t0) * 0.01; a += us / 32; if (us > m) m = us; }
            printf("  XCD %d: exit mean %7.2f max %7.2f us\n", x, a, m);
        }
    }
    // gaps between a tile's end and the next tile's start
    double gap = 0; long gc = 0;
    for (int g = 0; g < 256; ++g)
        for (int it = 1; it < tiles_per_wg && it < 64; ++it) { gap += (double)(hs[((size_t)g * 64 + it) * 8] - hs[((size_t)g * 64 + it - 1) * 8 + 6]); ++gc; }
    printf("  between tiles %8.0f cycles\n", gap / gc);
    return 0;
}
