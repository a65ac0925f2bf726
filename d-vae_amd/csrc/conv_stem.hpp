// First convolution of the stem in bf16: 3x3, unit stride, an RGB image stored as 8-channel pixels (16 bytes), 64 output
// channels.  With 8 gathered channels a tap is one 16-byte vector, i.e. half of the 16-deep K step of
// v_mfma_f32_32x32x16_bf16: the two lane halves of the pixel operand read two DIFFERENT taps (lane half h of step s reads
// tap 2 s + h), so the nine taps take five MFMA steps (the tenth tap has zero weights) instead of the nine 64-deep K
// slabs -- mostly padding -- that the gather GEMM spends on this layer.  What is left is the 128 bytes of output per
// pixel: the kernel is write-bound.
//
// Persistent 256-thread workgroups walk 8 x 32 pixel tiles: halo (10 x 34 pixels, 5.4 KB) to LDS, 20 MFMAs per wave
// (64 pixels x 64 channels), epilogue (scale / shift / ReLU: the folded BatchNorm of the eval path) into the wave's LDS
// patch, whole 128-byte pixel rows to HBM.  Four workgroups share a CU (36 KB of LDS each) and hide each other's
// store latency; the next tile's halo is fetched into registers under the current tile's MFMAs and stores.
#pragma once
#include "igemm.hpp"

namespace dvae {

struct StemArgs {
    const bf16* in;            // [B][H][W][8]
    const bf16* wp;            // packed rows [64][Ktot], K = (tap, channel 0..7)
    bf16* out;                 // [B][H][W][64]
    const float* scale;        // per output channel or NULL (1)
    const float* shift;        // per output channel or NULL (0)
    int relu;
    int H, W, tiles_x, tiles_y, ntiles, Ktot;
    signed char dy[16], dx[16];
};

constexpr int ST_HW = 34, ST_HP = 10 * ST_HW;          // halo of an 8 x 32 tile
constexpr int ST_WR = 10 * 16 + 16;                    // weight row: 10 taps x 16 bytes + pad
constexpr int ST_SRS = 64 * 2 + 16;                    // patch row: 64 channels bf16 + pad

__global__ __launch_bounds__(256) void conv_stem_bf16_kernel(const StemArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char halo[ST_HP * 16];
    __shared__ __attribute__((aligned(16))) unsigned char wlds[64 * ST_WR];
    __shared__ __attribute__((aligned(16))) unsigned char patch[4 * 32 * ST_SRS];
    __shared__ __attribute__((aligned(16))) float ssl[128];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;

    // weights: [64][10 taps][8 channels], tap 9 zero
    for (int v = tid; v < 64 * 10; v += 256) {
        const int n = v / 10, t = v % 10;
        u32x4 w = {0u, 0u, 0u, 0u};
        if (t < 9) w = *reinterpret_cast<const u32x4*>(a.wp + (long)n * a.Ktot + t * 8);
        *reinterpret_cast<u32x4*>(wlds + n * ST_WR + t * 16) = w;
    }
    if (tid < 64) {
        ssl[tid] = a.scale ? a.scale[tid] : 1.f;
        ssl[64 + tid] = a.shift ? a.shift[tid] : 0.f;
    }
    // lane half h of step s reads tap 2 s + h; the tenth tap reads the centre pixel (its weights are zero)
    int toff[5];
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        const int t = 2 * s + h;
        toff[s] = t < 9 ? ((int)a.dy[t] * ST_HW + (int)a.dx[t]) * 16 : 0;
    }
    // wave w owns tile rows 2w, 2w + 1: pixel block p (0, 1) is row 2w + p, lane r its column
    const unsigned char* hrow = halo + ((2 * wave + 1) * ST_HW + r + 1) * 16;
    const unsigned char* wrow = wlds + r * ST_WR + h * 16;
    unsigned char* stg = patch + wave * (32 * ST_SRS);
    const int sp = lane >> 3, sc8 = lane & 7;              // coalesced side: 8 pixels x 8 sixteen-byte chunks per instruction

    u32x4 hv[2];
    auto fetch = [&](int tile) {                           // this thread's two halo pixels of `tile` -> registers
        const int tx = tile % a.tiles_x, q = tile / a.tiles_x, ty = q % a.tiles_y, b = q / a.tiles_y;
        const bf16* ib = a.in + (long)b * a.H * a.W * 8;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int hp = tid + 256 * k;
            const int hy = hp / ST_HW, hx = hp - hy * ST_HW;
            const int iy = ty * 8 + hy - 1, ix = tx * 32 + hx - 1;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (hp < ST_HP && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
                v = *reinterpret_cast<const u32x4*>(ib + (iy * a.W + ix) * 8);
            hv[k] = v;
        }
    };
    auto wave_lds_fence = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    if ((int)blockIdx.x < a.ntiles) fetch(blockIdx.x);
    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        const int tx = tile % a.tiles_x, q = tile / a.tiles_x, ty = q % a.tiles_y, b = q / a.tiles_y;
        const int y0 = ty * 8, x0 = tx * 32;
        __syncthreads();                                   // the previous tile's halo has been read by every wave
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int hp = tid + 256 * k;
            if (hp < ST_HP) *reinterpret_cast<u32x4*>(halo + hp * 16) = hv[k];
        }
        __syncthreads();                                   // halo (and, first time, weights and scale / shift) visible
        if (tile + (int)gridDim.x < a.ntiles) fetch(tile + gridDim.x);   // in flight under this tile's MFMAs and stores

        f32x16 acc[2][2];                                  // [pixel block][channel block]
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[p][j][e] = 0.f;
#pragma unroll
        for (int s = 0; s < 5; ++s) {
            bf16x8 wv[2], pv[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) wv[j] = *reinterpret_cast<const bf16x8*>(wrow + j * 32 * ST_WR + s * 32);
#pragma unroll
            for (int p = 0; p < 2; ++p) pv[p] = *reinterpret_cast<const bf16x8*>(hrow + p * ST_HW * 16 + toff[s]);
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    // D[i = out channel][j = pixel]: a lane ends up with 4 consecutive channels per register quad
                    acc[p][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wv[j], pv[p], acc[p][j], 0, 0, 0);
        }
        // epilogue, one 32-pixel row at a time: lane (pixel r, quad g of channel block j) -> the wave's patch
        // [32 pixels][64 channels] -> whole 128-byte pixel rows (8 pixels per store instruction)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int nl = j * 32 + 8 * g + 4 * h;
                    const f32x4 sc = *reinterpret_cast<const f32x4*>(ssl + nl);
                    const f32x4 sh = *reinterpret_cast<const f32x4*>(ssl + 64 + nl);
                    bf16x4 o;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        float v = acc[p][j][4 * g + k] * sc[k] + sh[k];
                        if (a.relu) v = fmaxf(v, 0.f);
                        o[k] = (bf16)v;
                    }
                    *reinterpret_cast<bf16x4*>(stg + r * ST_SRS + nl * 2) = o;
                }
            wave_lds_fence();
            bf16* ob = a.out + ((long)(b * a.H + y0 + 2 * wave + p) * a.W + x0) * 64;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int px = i * 8 + sp;
                const u32x4 v = *reinterpret_cast<const u32x4*>(stg + px * ST_SRS + sc8 * 16);
                *reinterpret_cast<u32x4*>(ob + px * 64 + sc8 * 8) = v;
            }
            wave_lds_fence();
        }
    }
}

static bool stem_applicable(const GemmPlan& P) {
    if (P.nphase != 1 || P.CgS != 8 || P.N != 64 || P.NS != 64) return false;
    const Phase& ph = P.ph[0];
    if (ph.ntaps != 9 || ph.SY != 1 || ph.SX != 1 || ph.DSY != 1 || ph.DSX != 1 || ph.D0Y || ph.D0X) return false;
    if (P.Hin != P.Hout || P.Win != P.Wout || ph.Hg != P.Hout || ph.Wg != P.Wout) return false;
    for (int t = 0; t < 9; ++t)
        if (ph.dy[t] < -1 || ph.dy[t] > 1 || ph.dx[t] < -1 || ph.dx[t] > 1) return false;
    if (P.Wout % 32 || P.Hout % 8) return false;
    return (long)P.B * P.Hout * P.Wout * 64 <= 0x7fffffffL;
}

static bool try_launch_stem_bf16(const GemmPlan& P, const void* in, const void* wp, void* out, const float* scale,
                                 const float* shift, const void* res, int relu, int accumulate, int out_f32,
                                 hipStream_t st) {
    if (res || accumulate || out_f32 || !stem_applicable(P)) return false;
    StemArgs a;
    a.in = (const bf16*)in; a.wp = (const bf16*)wp; a.out = (bf16*)out; a.scale = scale; a.shift = shift; a.relu = relu;
    a.H = P.Hout; a.W = P.Wout; a.tiles_x = P.Wout / 32; a.tiles_y = P.Hout / 8;
    a.ntiles = P.B * a.tiles_x * a.tiles_y; a.Ktot = P.Ktot;
    memcpy(a.dy, P.ph[0].dy, 16); memcpy(a.dx, P.ph[0].dx, 16);
    const int grid = a.ntiles < 1024 ? a.ntiles : 1024;    // four workgroups per CU (36 KB of LDS each)
    hipLaunchKernelGGL(conv_stem_bf16_kernel, dim3(grid), dim3(256), 0, st, a);
    return true;
}

}  // namespace dvae
