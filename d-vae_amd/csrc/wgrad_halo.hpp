// Weight gradient of a 3x3 unit-stride 64 -> 64 convolution (bf16), all nine taps from one staged tile.
//
// dW[tap][cx][cd] = sum over pixels of X[pixel + tap offset][cx] * D[pixel][cd].  The slab kernel in igemm.hpp gives
// every (tap, pixel chunk) its own workgroup, so X and D are fetched nine times and each fetch pays the per-row index
// arithmetic of a gather.  Here a workgroup owns a run of 4 x 32 pixel tiles: per tile it stages the 6 x 34 halo of X
// and the 4 x 32 tile of D in LDS once ([pixel][channel] rows of 192 bytes, as the transposed LDS reads want them) and
// takes all nine taps out of that copy -- a tap is just another start row in the halo.  The four waves split the
// 64 x 64 output of a tap into 32 x 32 quadrants; each keeps its quadrant of all nine taps in 144 accumulator
// registers for its whole run of tiles, so the only output is one [576][64] fp32 slab per workgroup, in the layout
// wgrad_reduce_kernel sums.  LDS is double-buffered and the registers run two tiles ahead: a tile's rows are fetched
// into one of two register sets before the MFMA loop of the tile two before it (one loop, 1.1 us, is shorter than a
// memory round trip under load) and written to the idle LDS buffer after the loop of the tile before it; one barrier
// per tile.
#pragma once
#include <type_traits>

#include "igemm.hpp"

namespace dvae {

struct WgradHaloArgs {
    const bf16* xg;            // gathered tensor [B][H][W][64]
    const bf16* d;             // dense tensor    [B][H][W][64]
    float* ws;                 // [gridDim.x][576][64]
    int H, W, tiles_x, tiles_y, ntiles, per;   // tiles of 4 x 32 pixels; `per` consecutive tiles per workgroup
    int bias_mode;             // 0 none, 1 column sums of d, 2 column sums of xg (its centre tap reads each pixel once)
    float* bpart;              // mode 1: [gridDim.x][64]; mode 2: [gridDim.x][576], columns 256..319
};

constexpr int WH_RS = 192, WH_HP = 6 * 34, WH_XB = WH_HP * WH_RS, WH_DB = 128 * WH_RS, WH_BUF = WH_XB + WH_DB;
constexpr int WH_LDS = 2 * WH_BUF;

__global__ __launch_bounds__(256) void wgrad_halo_bf16_kernel(const WgradHaloArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wi = wave >> 1, wj = wave & 1;
    const int c8 = tid & 7, p0 = tid >> 3;              // this thread stages vector c8 of pixels p0 + 32 k
    // `per` is even: the tile loop is unrolled by two (one copy per register set) without a branch between the copies;
    // slots past the last tile are staged as zeros and add nothing
    const int t0 = blockIdx.x * a.per, t1 = t0 + a.per;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    float bs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    u32x4 hregs[2][7], dregs[2][4];                     // register set s holds tile t0 + s (mod 2)
    auto issue = [&](int t, auto setc) {
        constexpr int SET = decltype(setc)::value;
        u32x4 (&hreg)[7] = hregs[SET];
        u32x4 (&dreg)[4] = dregs[SET];
        const bool live = t < a.ntiles;
        const int tx = t % a.tiles_x, r = t / a.tiles_x, ty = r % a.tiles_y, b = r / a.tiles_y;
        const int y0 = ty * 4, x0 = tx * 32;
        const bf16* xb = a.xg + (long)b * a.H * a.W * 64 + c8 * 8;
        const bf16* db = a.d + (long)b * a.H * a.W * 64 + c8 * 8;
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            const int hp = p0 + 32 * k;
            const int hy = hp / 34, hx = hp - hy * 34;
            const int iy = y0 + hy - 1, ix = x0 + hx - 1;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (live && hp < WH_HP && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
                v = *reinterpret_cast<const u32x4*>(xb + (iy * a.W + ix) * 64);
            hreg[k] = v;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int p = p0 + 32 * k;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (live) v = *reinterpret_cast<const u32x4*>(db + ((y0 + (p >> 5)) * a.W + x0 + (p & 31)) * 64);
            dreg[k] = v;
        }
    };
    auto colsum = [&](const u32x4& v) {
#pragma unroll
        for (int c = 0; c < 4; ++c) { bs[2 * c] += __uint_as_float(v[c] << 16); bs[2 * c + 1] += __uint_as_float(v[c] & 0xffff0000u); }
    };
    auto stash = [&](auto bufc, auto setc) {
        constexpr int buf = decltype(bufc)::value, SET = decltype(setc)::value;
        const u32x4 (&hreg)[7] = hregs[SET];
        const u32x4 (&dreg)[4] = dregs[SET];
        unsigned char* xh = smem + buf * WH_BUF;
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            const int hp = p0 + 32 * k;
            if (hp < WH_HP) {
                *reinterpret_cast<u32x4*>(xh + hp * WH_RS + c8 * 16) = hreg[k];
                if (a.bias_mode == 2) {
                    const int hy = hp / 34, hx = hp - hy * 34;
                    if (hy >= 1 && hy <= 4 && hx >= 1 && hx <= 32) colsum(hreg[k]);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            *reinterpret_cast<u32x4*>(xh + WH_XB + (p0 + 32 * k) * WH_RS + c8 * 16) = dreg[k];
            if (a.bias_mode == 1) colsum(dreg[k]);
        }
    };

    // one tile: fetch tile t + 2 into the register set tile t came from, MFMAs on LDS buffer `buf`, then tile t + 1
    // (the other register set, fetched a tile ago) goes to the other buffer
    auto tile_step = [&](int t, auto bufc) {
        constexpr int buf = decltype(bufc)::value;     // LDS buffer and register set of tile t; the other ones hold tile t + 1
        if (t + 2 < t1) issue(t + 2, bufc);
        const unsigned char* xh = smem + buf * WH_BUF;
        const unsigned char* dt = xh + WH_XB;
        // One wave per SIMD: nothing else hides the LDS latency, so the fragment reads are pipelined by hand.  The
        // MFMAs are grouped by halo row (the taps of output row ty that read halo row hy = ty + ky); the six X
        // fragments of the next halo row are requested before the MFMAs of the current one are issued.
        bf16x8 bfr[8], afr[2][6];
#pragma unroll
        for (int s = 0; s < 8; ++s) bfr[s] = tr_frag(dt, (s >> 1) * 32 + (s & 1) * 16, wj * 32, lane);
        auto read_row = [&](int hy, int par) {
#pragma unroll
            for (int xi = 0; xi < 2; ++xi)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) afr[par][xi * 3 + kx] = tr_frag(xh, hy * 34 + xi * 16 + kx, wi * 32, lane);
        };
        read_row(0, 0);
#pragma unroll
        for (int hy = 0; hy < 6; ++hy) {
            if (hy < 5) read_row(hy + 1, (hy + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int xi = 0; xi < 2; ++xi)
#pragma unroll
                for (int ty = 0; ty < 4; ++ty) {
                    const int ky = hy - ty;
                    if (ky < 0 || ky > 2) continue;
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx)
                        acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[hy & 1][xi * 3 + kx], bfr[ty * 2 + xi],
                                                                                   acc[ky * 3 + kx], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (t + 1 < t1) stash(std::integral_constant<int, buf ^ 1>{}, std::integral_constant<int, buf ^ 1>{});
        __syncthreads();
    };

    if (t0 < t1) {
        issue(t0, std::integral_constant<int, 0>{});
        issue(t0 + 1, std::integral_constant<int, 1>{});
        stash(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
    }
    __syncthreads();
    for (int t = t0; t < t1; t += 2) {
        tile_step(t, std::integral_constant<int, 0>{});
        tile_step(t + 1, std::integral_constant<int, 1>{});
    }

    const int r = lane & 31, h = lane >> 5;
    float* ws = a.ws + (long)blockIdx.x * (576 * 64) + (wi * 32) * 64 + wj * 32 + r;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int e = 0; e < 16; ++e) ws[(tap * 64 + (e & 3) + 8 * (e >> 2) + 4 * h) * 64] = acc[tap][e];
    float* red = reinterpret_cast<float*>(smem);        // every wave is past the last barrier of the tile loop
    if (a.bias_mode == 1) wgrad_bias_flush<8, 32>(bs, red, c8, p0, a.bpart + (long)blockIdx.x * 64);
    if (a.bias_mode == 2) wgrad_bias_flush<8, 32>(bs, red, c8, p0, a.bpart + (long)blockIdx.x * 576 + 4 * 64);
}

}  // namespace dvae
