// Weight gradient of a 3x3 unit-stride convolution (bf16), all nine taps from one staged tile.
//
// dW[tap][cx][cd] = sum over pixels of X[pixel + tap offset][cx] * D[pixel][cd].  The slab kernel in igemm.hpp gives
// every (tap, pixel chunk) its own workgroup, so X and D are fetched nine times and each fetch pays the per-row index
// arithmetic of a gather.  Here a workgroup owns a run of 4 x 32 pixel tiles: per tile it stages the 6 x 34 halo of X
// and the 4 x 32 tile of D in LDS once ([pixel][channel] rows of 192 bytes, as the transposed LDS reads want them) and
// takes all nine taps out of that copy -- a tap is just another start row in the halo.  The four waves split the
// CX x CD output of a tap into 32 x 32 quadrants (64 x 64: 2 x 2) and, where a tap has fewer than four quadrants, the
// tile's four pixel rows among themselves (32 x 64: two quadrants x two row pairs; 8 x 32: one quadrant x four rows);
// each wave keeps its quadrant of all nine taps in 144 accumulator registers for its whole run of tiles, so the only
// output is one [9 CX][CD] fp32 slab per workgroup (row groups add their accumulators through LDS at the end), in the
// layout wgrad_reduce_kernel sums.  LDS
// is double-buffered and the registers run two tiles ahead: a tile's rows are fetched into one of two register sets
// before the MFMA loop of the tile two before it (one loop, 1.1 us, is shorter than a memory round trip under load) and
// written to the idle LDS buffer after the loop of the tile before it; one barrier per tile.
// CX, CD: channels per stored pixel of the gathered / dense tensor, 8, 32 or 64 (an 8-channel pixel fills a quarter of
// its quadrant's rows; the rest multiplies whatever the padding of the staged row holds and is never written out).
#pragma once
#include <type_traits>

#include "igemm.hpp"

namespace dvae {

struct WgradHaloArgs {
    const bf16* xg;            // gathered tensor [B][H][W][CX]
    const bf16* d;             // dense tensor    [B][H][W][CD]
    float* ws;                 // [gridDim.x][Kpad][CdPad]
    int H, W, tiles_x, tiles_y, ntiles, per;   // tiles of 4 x 32 pixels; `per` consecutive tiles per workgroup
    int Kpad, CdPad;           // slab dimensions (9 CX and CD rounded up to 64)
    int bias_mode;             // 0 none, 1 column sums of d, 2 column sums of xg (its centre tap reads each pixel once)
    float* bpart;              // mode 1: [slabs][CdPad]; mode 2: [slabs][Kpad], columns 4 CX .. 5 CX
};

constexpr int WH_RS = 192, WH_HP = 6 * 34, WH_XB = WH_HP * WH_RS, WH_DB = 128 * WH_RS, WH_BUF = WH_XB + WH_DB;
constexpr int WH_LDS = 2 * WH_BUF;

template <int CX, int CD>
__global__ __launch_bounds__(256) void wgrad_halo_bf16_kernel(const WgradHaloArgs a) {
    constexpr int VX = CX / 8, VD = CD / 8;                // 16-byte vectors per pixel
    constexpr int NXV = (WH_HP * VX + 255) / 256, NDV = (128 * VD + 255) / 256;   // vectors per thread and tile
    constexpr int WIN = CX >= 32 ? CX / 32 : 1, WJN = CD >= 32 ? CD / 32 : 1, WPN = 4 / (WIN * WJN), RPW = 4 / WPN;
    static_assert(WIN * WJN * WPN == 4 && CX % 8 == 0 && CD % 8 == 0 && CX <= 64 && CD <= 64, "wave split");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wj = wave % WJN, wi = (wave / WJN) % WIN, wp = wave / (WJN * WIN);
    // `per` is even: the tile loop is unrolled by two (one copy per register set) without a branch between the copies;
    // slots past the last tile are staged as zeros and add nothing
    const int t0 = blockIdx.x * a.per, t1 = t0 + a.per;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    float bs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    // this thread stages vectors tid + 256 k: always the same 8 channels of its pixels (256 is a multiple of VX and VD)
    const int cx8 = tid % VX, cd8 = tid % VD;
    u32x4 hregs[2][NXV], dregs[2][NDV];                 // register set s holds tile t0 + s (mod 2)
    auto issue = [&](int t, auto setc) {
        constexpr int SET = decltype(setc)::value;
        u32x4 (&hreg)[NXV] = hregs[SET];
        u32x4 (&dreg)[NDV] = dregs[SET];
        const bool live = t < a.ntiles;
        const int tx = t % a.tiles_x, r = t / a.tiles_x, ty = r % a.tiles_y, b = r / a.tiles_y;
        const int y0 = ty * 4, x0 = tx * 32;
        const bf16* xb = a.xg + (long)b * a.H * a.W * CX + cx8 * 8;
        const bf16* db = a.d + (long)b * a.H * a.W * CD + cd8 * 8;
#pragma unroll
        for (int k = 0; k < NXV; ++k) {
            const int hp = (tid + 256 * k) / VX;
            const int hy = hp / 34, hx = hp - hy * 34;
            const int iy = y0 + hy - 1, ix = x0 + hx - 1;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (live && hp < WH_HP && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
                v = *reinterpret_cast<const u32x4*>(xb + (iy * a.W + ix) * CX);
            hreg[k] = v;
        }
#pragma unroll
        for (int k = 0; k < NDV; ++k) {
            const int p = (tid + 256 * k) / VD;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (live && p < 128) v = *reinterpret_cast<const u32x4*>(db + ((y0 + (p >> 5)) * a.W + x0 + (p & 31)) * CD);
            dreg[k] = v;
        }
    };
    auto colsum = [&](const u32x4& v) {
#pragma unroll
        for (int c = 0; c < 4; ++c) { bs[2 * c] += __uint_as_float(v[c] << 16); bs[2 * c + 1] += __uint_as_float(v[c] & 0xffff0000u); }
    };
    auto stash = [&](auto bufc, auto setc) {
        constexpr int buf = decltype(bufc)::value, SET = decltype(setc)::value;
        const u32x4 (&hreg)[NXV] = hregs[SET];
        const u32x4 (&dreg)[NDV] = dregs[SET];
        unsigned char* xh = smem + buf * WH_BUF;
#pragma unroll
        for (int k = 0; k < NXV; ++k) {
            const int hp = (tid + 256 * k) / VX;
            if (hp < WH_HP) {
                *reinterpret_cast<u32x4*>(xh + hp * WH_RS + cx8 * 16) = hreg[k];
                if (a.bias_mode == 2) {
                    const int hy = hp / 34, hx = hp - hy * 34;
                    if (hy >= 1 && hy <= 4 && hx >= 1 && hx <= 32) colsum(hreg[k]);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < NDV; ++k) {
            const int p = (tid + 256 * k) / VD;
            if (p < 128) {
                *reinterpret_cast<u32x4*>(xh + WH_XB + p * WH_RS + cd8 * 16) = dreg[k];
                if (a.bias_mode == 1) colsum(dreg[k]);
            }
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
        // fragments of the next halo row are requested before the MFMAs of the current one are issued.  A wave that
        // owns only RPW of the tile's four pixel rows (ty in [wp RPW, wp RPW + RPW)) walks the RPW + 2 halo rows they read.
        bf16x8 bfr[2 * RPW], afr[2][6];
        const int ty0 = wp * RPW;
#pragma unroll
        for (int s = 0; s < 2 * RPW; ++s) bfr[s] = tr_frag(dt, (ty0 + (s >> 1)) * 32 + (s & 1) * 16, wj * 32, lane);
        auto read_row = [&](int hy, int par) {
#pragma unroll
            for (int xi = 0; xi < 2; ++xi)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) afr[par][xi * 3 + kx] = tr_frag(xh, hy * 34 + xi * 16 + kx, wi * 32, lane);
        };
        read_row(ty0, 0);
#pragma unroll
        for (int hr = 0; hr < RPW + 2; ++hr) {             // halo row ty0 + hr
            if (hr < RPW + 1) read_row(ty0 + hr + 1, (hr + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int xi = 0; xi < 2; ++xi)
#pragma unroll
                for (int tr = 0; tr < RPW; ++tr) {         // output row ty0 + tr
                    const int ky = hr - tr;
                    if (ky < 0 || ky > 2) continue;
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx)
                        acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[hr & 1][xi * 3 + kx], bfr[tr * 2 + xi],
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

    // ---- the pixel-row groups' accumulators are summed through LDS (fixed order), so a workgroup leaves ONE slab ----
    float* red = reinterpret_cast<float*>(smem);        // every wave is past the last barrier of the tile loop
    if (WPN > 1) {
        static_assert((4 - WIN * WJN) * 9 * 16 * 64 * 4 <= WH_LDS, "accumulator exchange fits the tile buffers");
        if (wp > 0) {
            float* o = red + (long)(wave - WIN * WJN) * (9 * 16 * 64) + lane;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
#pragma unroll
                for (int e = 0; e < 16; ++e) o[(tap * 16 + e) * 64] = acc[tap][e];
        }
        __syncthreads();
        if (wp == 0) {
#pragma unroll
            for (int g = 1; g < WPN; ++g) {
                const float* o = red + (long)(wave + (g - 1) * WIN * WJN) * (9 * 16 * 64) + lane;
#pragma unroll
                for (int tap = 0; tap < 9; ++tap)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[tap][e] += o[(tap * 16 + e) * 64];
            }
        }
        __syncthreads();
    }
    // ---- slab blockIdx.x: rows tap * CX + channel, the channels of this wave's quadrant that exist ----
    const int r = lane & 31, h = lane >> 5;
    if (wp == 0 && wj * 32 + r < CD) {
        float* ws = a.ws + (long)blockIdx.x * a.Kpad * a.CdPad + wj * 32 + r;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int cx = wi * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (cx < CX) ws[(long)(tap * CX + cx) * a.CdPad] = acc[tap][e];
            }
    }
    // ---- bias partial sums: one row per slab ----
    if (a.bias_mode == 1) {
        constexpr int RL = 256 / VD;
#pragma unroll
        for (int i = 0; i < 8; ++i) red[(tid / VD) * CD + cd8 * 8 + i] = bs[i];
        __syncthreads();
        if (tid < CD) {
            float t = 0.f;
            for (int q = 0; q < RL; ++q) t += red[q * CD + tid];
            a.bpart[(long)blockIdx.x * a.CdPad + tid] = t;
        }
    }
    if (a.bias_mode == 2) {
        constexpr int RL = 256 / VX;
#pragma unroll
        for (int i = 0; i < 8; ++i) red[(tid / VX) * CX + cx8 * 8 + i] = bs[i];
        __syncthreads();
        if (tid < CX) {
            float t = 0.f;
            for (int q = 0; q < RL; ++q) t += red[q * CX + tid];
            a.bpart[(long)blockIdx.x * a.Kpad + 4 * CX + tid] = t;
        }
    }
}

}  // namespace dvae
