// 3x3 unit-stride bf16 convolution from 32 stored channels to at most 4 output channels, fp32 output, on 32-pixel-wide
// images: the decoders' last layer (32 -> RGB).  HBM-bound: 64 B read and 16 B written per pixel against 1.7 kFLOP.
//
// With three output channels a 32x32 MFMA tile is 29/32 padding and the pixel operand of every tap is a separate LDS
// read.  Here the horizontal taps ride in the output rows of a v_mfma_f32_16x16x32_bf16 instead: row m = 4 dx + o of
// the weight operand holds W[dy][dx][o][0..31] (K = the 32 input channels of ONE pixel), the pixel operand is 16
// consecutive pixels of image row y + dy, and the three dy steps accumulate into
//     D[4 dx + o][q] = sum_dy sum_c W[dy][dx][o][c] x[y + dy][q][c] ,
// three MFMAs and three 16-byte LDS reads per 16 pixels.  The output is out[y][x][o] = D[o][x-1] + D[4+o][x] + D[8+o][x+1]:
// a lane holds the four o of one (dx, q) as its accumulator quad, parks it in the wave's LDS patch and reads the three
// shifted quads of its output pixel back.  Columns -1 and 32 are the zero padding of a 32-wide image, so a tile needs no
// halo columns: its ten input rows are one contiguous span of memory.  Wider images (WIDE) are walked in tiles of 32
// columns: the two columns beside a tile are staged behind the tile's own pixels and a third MFMA group per output row
// computes D[o] of the left one and D[8 + o] of the right one into the patch's edge slots.
//
// A workgroup (4 waves) owns tiles of 8 output rows; a wave two of them.  Persistent over tiles with the next tile's
// rows fetched into registers before the current one is computed; the optional per-(image, channel) scale / shift
// (+ ReLU) of the producer's InstanceNorm is applied between registers and LDS (padding rows stay zero).  33 KB of LDS:
// four workgroups per CU keep ~80 KB of loads in flight per CU.
#pragma once
#include "conv_halo.hpp"

namespace dvae {

struct RgbArgs {
    const void* in;          // bf16 [B][H][32][32]
    const void* wp;          // packed forward weights [Npad][Ktot], K = (tap, channel)
    float* out;              // fp32 [B][H][32][4]
    const float* scale;      // per output channel, or null (1)
    const float* shift;      // per output channel, or null (0)
    const float* in_ss;      // optional input normalisation [B][2][32]: x <- relu?(x * scale + shift) while staging
    int in_relu;
    int B, H, W, N, Ktot, tiles_per_wg;
    int tap[3][3];           // packed tap index of offset (dy - 1, dx - 1)
    TileGrid g;              // tiles of an image: W / 32 per tile row, H / 8 tile rows
};

static constexpr int RGB_HALO_BYTES = 10 * 32 * 64;            // ten rows of 32 pixels x 32 bf16 channels
static constexpr int RGB_PLANE = 34 * 16;                      // one dx plane of one output row: columns -1 .. 32, f32x4 each
static constexpr int RGB_PATCH_BYTES = 2 * 3 * RGB_PLANE;      // per wave: two output rows
static constexpr int RGB_EDGE_BYTES = 10 * 2 * 64;              // WIDE: the columns left and right of the tile, [row][side]
static constexpr int RGB_LDS_BYTES = RGB_HALO_BYTES + RGB_EDGE_BYTES + 4 * RGB_PATCH_BYTES;

template <bool WIDE>
__global__ __launch_bounds__(256, 4) void conv_rgb_bf16_kernel(const RgbArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[RGB_LDS_BYTES];
    unsigned char* halo = smem;
    unsigned char* edge = smem + RGB_HALO_BYTES;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    unsigned char* patch = smem + RGB_HALO_BYTES + RGB_EDGE_BYTES + wave * RGB_PATCH_BYTES;
    const bf16* in = reinterpret_cast<const bf16*>(a.in);
    const bf16* wp = reinterpret_cast<const bf16*>(a.wp);
    const int ntiles = a.B * a.g.tpi;

    // ---- weight operand of the three dy steps: lane (m = lane & 15, kg = lane >> 4) holds W[dy][m >> 2][m & 3][8 kg ..]
    const int q = lane & 15, kg = lane >> 4;
    bf16x8 wa[3];
    {
        const int dxi = q >> 2, o = q & 3;
        const bool live = dxi < 3 && o < a.N;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            u32x4 w = {0u, 0u, 0u, 0u};
            if (live) w = *reinterpret_cast<const u32x4*>(wp + (long)o * a.Ktot + a.tap[d][dxi] * 32 + kg * 8);
            wa[d] = *reinterpret_cast<bf16x8*>(&w);
        }
    }
    f32x4 esc, esh;
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        esc[o] = o < a.N ? (a.scale ? a.scale[o] : 1.f) : 0.f;
        esh[o] = (o < a.N && a.shift) ? a.shift[o] : 0.f;
    }
    // the patch's edge columns are read as the zero padding of the dx = -1 / +1 planes and never written
    // (WIDE: written per tile by the edge group)
    for (int i = tid; i < 4 * RGB_PATCH_BYTES / 16; i += 256)
        *reinterpret_cast<u32x4*>(smem + RGB_HALO_BYTES + RGB_EDGE_BYTES + i * 16) = u32x4{0u, 0u, 0u, 0u};

    // ---- staging: vector v = tid + 256 k of the ten rows (128 vectors per row): row 2k + (tid >> 7), pixel column
    // (tid & 127) >> 2, 8-channel chunk tid & 3.  In LDS the four chunks of a pixel are permuted by a function of the
    // column so that the 16 lanes a ds_read_b128 serves together ({0-3, 12-15, 20-27}, ...) fall on 16 different
    // 16-byte slots of the 256-byte bank row: chunk ^ {0, 2, 3, 1}[(column >> 2) & 3].
    const int scol = (tid & 127) >> 2, schunk = tid & 3, srow0 = tid >> 7;
    const int sw_st = (schunk ^ ((0x1320 >> (((scol >> 2) & 3) * 4)) & 3)) * 16;
    unsigned char* sdst = halo + (srow0 * 32 + scol) * 64 + sw_st;
    // WIDE: the first 80 threads also take chunk tid & 3 of the column left (side 0) or right (side 1) of the tile in row
    // tid >> 3; the others load that vector too (no branch around a load) and do not store it
    constexpr int HV = WIDE ? 6 : 5;
    const int et = tid % 80, erow = et >> 3, eside = (et >> 2) & 1;
    u32x4 hreg[HV];
    unsigned live = 0;                                            // bit k: hreg[k] is a pixel inside the image
    f32x4 isc[2], ish[2];
    auto load_tile = [&](int tile) {
        int b, ty, tx;
        a.g.split(tile, b, ty, tx);
        const int y0 = ty * 8, x0 = tx * 32;
        // every load is unconditional (a load inside a branch is waited for where the branch ends): pixels outside the
        // image re-read an edge pixel and are zeroed on their way to LDS; without a normalisation the scale / shift
        // registers are filled from the input tensor and never used
        const float* p = a.in_ss ? a.in_ss + (long)b * 64 + schunk * 8 : reinterpret_cast<const float*>(a.in);
        isc[0] = *reinterpret_cast<const f32x4*>(p); isc[1] = *reinterpret_cast<const f32x4*>(p + 4);
        ish[0] = *reinterpret_cast<const f32x4*>(p + 32); ish[1] = *reinterpret_cast<const f32x4*>(p + 36);
        const bf16* img = in + (long)b * a.H * a.W * 32 + schunk * 8;
        live = 0;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int iy = y0 - 1 + 2 * k + srow0;
            const int cy = iy < 0 ? 0 : (iy >= a.H ? a.H - 1 : iy);
            hreg[k] = *reinterpret_cast<const u32x4*>(img + (cy * a.W + x0 + scol) * 32);
            if (iy == cy) live |= 1u << k;
        }
        if (WIDE) {
            const int iy = y0 - 1 + erow, ix = eside ? x0 + 32 : x0 - 1;
            const int cy = iy < 0 ? 0 : (iy >= a.H ? a.H - 1 : iy), cx = ix < 0 ? 0 : (ix >= a.W ? a.W - 1 : ix);
            hreg[HV - 1] = *reinterpret_cast<const u32x4*>(img + (cy * a.W + cx) * 32);
            if (iy == cy && ix == cx) live |= 1u << 5;
        }
    };
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    typedef bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    typedef short i16x2_t __attribute__((ext_vector_type(2)));
    const i16x2_t in_lo16 = {(short)(a.in_relu ? 0 : -32768), (short)(a.in_relu ? 0 : -32768)};
    auto store_tile = [&]() {
#pragma unroll
        for (int k = 0; k < HV; ++k) {
            u32x4 v = hreg[k];
            if (!(live >> k & 1u)) v = u32x4{0u, 0u, 0u, 0u};
            else if (a.in_ss) {
                // per pair of channels: two unpacks, a packed fma, a packed conversion, a packed integer max (ReLU on the
                // rounded pair: a bf16 read as a signed 16-bit integer is negative exactly when the number is)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const unsigned u = v[c];
                    const f32x2_t x = {__builtin_bit_cast(float, u << 16), __builtin_bit_cast(float, u & 0xffff0000u)};
                    const f32x2_t sc = {isc[c >> 1][2 * (c & 1)], isc[c >> 1][2 * (c & 1) + 1]};
                    const f32x2_t sh = {ish[c >> 1][2 * (c & 1)], ish[c >> 1][2 * (c & 1) + 1]};
                    const bf16x2_t o = __builtin_convertvector(__builtin_elementwise_fma(x, sc, sh), bf16x2_t);
                    v[c] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(i16x2_t, o), in_lo16));
                }
            }
            if (k < 5) *reinterpret_cast<u32x4*>(sdst + k * (2 * 32 * 64)) = v;
            else if (tid < 80) *reinterpret_cast<u32x4*>(edge + (erow * 2 + eside) * 64 + schunk * 16) = v;
        }
    };

    // ---- MFMA side: pixel operand of (halo row, 16-pixel group g): lane (q, kg) reads chunk kg of pixel 16 g + q
    const unsigned char* frd = halo + q * 64 + ((kg ^ ((0x1320 >> ((q >> 2) * 4)) & 3)) * 16);
    // combine side: lane = (output row rr of the wave's two, column x)
    const int rr = lane >> 5, x = lane & 31;
    auto wave_lds_fence = [&]() {                          // LDS hand-over between lanes of one wave: order, no barrier
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };

    // a workgroup walks a contiguous run of tiles (whole images in order): the two rows neighbouring tiles share are then
    // read by the same CU one tile apart and found in L2
    int tile = blockIdx.x * a.tiles_per_wg;
    const int tend = tile + a.tiles_per_wg < ntiles ? tile + a.tiles_per_wg : ntiles;
    if (tile < tend) load_tile(tile);
    for (; tile < tend; ++tile) {
        store_tile();
        __syncthreads();                                   // the tile's rows (first time: the zeroed patches) are visible
        const int next = tile + 1;
        if (next < tend) load_tile(next);                  // in flight under the MFMAs and the combine
#pragma unroll
        for (int r2 = 0; r2 < 2; ++r2) {
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    const bf16x8 pv = *reinterpret_cast<const bf16x8*>(frd + ((2 * wave + r2 + d) * 32 + 16 * g) * 64);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[d], pv, acc, 0, 0, 0);
                }
                if (kg < 3) *reinterpret_cast<f32x4*>(patch + (r2 * 3 + kg) * RGB_PLANE + (16 * g + q + 1) * 16) = acc;
            }
            if (WIDE) {
                // the edge group: pixel column q & 1 = left / right neighbour of the tile (every lane reads one of the two:
                // identical addresses are broadcast); lane (q = 0, kg = 0) holds D[o] of the left column = the dx = -1
                // term of output column 0, lane (q = 1, kg = 2) D[8 + o] of the right column = the dx = +1 term of column 31
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    const bf16x8 pv = *reinterpret_cast<const bf16x8*>(edge + ((2 * wave + r2 + d) * 2 + (q & 1)) * 64 + kg * 16);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[d], pv, acc, 0, 0, 0);
                }
                if (q == 0 && kg == 0) *reinterpret_cast<f32x4*>(patch + (r2 * 3 + 0) * RGB_PLANE) = acc;
                if (q == 1 && kg == 2) *reinterpret_cast<f32x4*>(patch + (r2 * 3 + 2) * RGB_PLANE + 33 * 16) = acc;
            }
        }
        wave_lds_fence();
        {
            const unsigned char* pr = patch + rr * 3 * RGB_PLANE + x * 16;
            const f32x4 l = *reinterpret_cast<const f32x4*>(pr);                          // dx = -1: D[o][x - 1]
            const f32x4 c = *reinterpret_cast<const f32x4*>(pr + RGB_PLANE + 16);         // dx =  0: D[4 + o][x]
            const f32x4 r = *reinterpret_cast<const f32x4*>(pr + 2 * RGB_PLANE + 32);     // dx = +1: D[8 + o][x + 1]
            f32x4 o4;
#pragma unroll
            for (int o = 0; o < 4; ++o) o4[o] = ((l[o] + c[o]) + r[o]) * esc[o] + esh[o];
            int b, ty, tx;
            a.g.split(tile, b, ty, tx);
            *reinterpret_cast<f32x4*>(a.out + (((long)b * a.H + ty * 8 + 2 * wave + rr) * a.W + tx * 32 + x) * 4) = o4;
        }
        __syncthreads();                                   // every wave is done with these rows (and with its patch)
    }
}

// true if the plan is a launch of the kernel above
static bool rgb_applicable(const GemmPlan& P, int out_f32) {
    if (!out_f32 || P.nphase != 1 || P.CgS != 32 || P.N > 4) return false;
    const Phase& ph = P.ph[0];
    if (ph.SY != 1 || ph.SX != 1 || ph.DSY != 1 || ph.DSX != 1 || ph.D0Y || ph.D0X || ph.ntaps != 9) return false;
    if (P.Hin != P.Hout || P.Win != P.Wout || ph.Hg != P.Hout || ph.Wg != P.Wout) return false;
    if (P.Wout % 32 || P.Hout % 8 || (long)P.Hout * P.Wout * 32 > 0x3fffffffL) return false;   // per-image offsets are 32-bit
    TileGrid tg;
    if (!make_tile_grid(P.Wout / 32, P.Hout / 8, ((long)P.B + 1024) * (P.Wout / 32) * (P.Hout / 8), tg)) return false;
    unsigned seen = 0;
    for (int t = 0; t < 9; ++t) {
        if (ph.dy[t] < -1 || ph.dy[t] > 1 || ph.dx[t] < -1 || ph.dx[t] > 1) return false;
        seen |= 1u << ((ph.dy[t] + 1) * 3 + ph.dx[t] + 1);
    }
    return seen == 0x1ffu;
}

static bool try_launch_rgb_bf16(const GemmPlan& P, const void* in, const void* wp, void* out, const float* scale,
                                const float* shift, const void* res, int relu, int accumulate, int out_f32, hipStream_t st,
                                const float* in_ss, int in_relu, const float* out_stats) {
    if (res || relu || accumulate || out_stats || !rgb_applicable(P, out_f32)) return false;
    const Phase& ph = P.ph[0];
    RgbArgs a;
    a.in = in; a.wp = wp; a.out = reinterpret_cast<float*>(out); a.scale = scale; a.shift = shift;
    a.in_ss = in_ss; a.in_relu = in_relu;
    a.B = P.B; a.H = P.Hout; a.W = P.Wout; a.N = P.N; a.Ktot = P.Ktot;
    for (int t = 0; t < 9; ++t) a.tap[ph.dy[t] + 1][ph.dx[t] + 1] = t;
    make_tile_grid(P.Wout / 32, P.Hout / 8, 0, a.g);
    const long ntiles = (long)P.B * a.g.tpi;
    a.tiles_per_wg = (int)((ntiles + 256 * 4 - 1) / (256 * 4));       // four workgroups per CU
    const int gx = (int)((ntiles + a.tiles_per_wg - 1) / a.tiles_per_wg);
    if (P.Wout == 32) hipLaunchKernelGGL(conv_rgb_bf16_kernel<false>, dim3(gx), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(conv_rgb_bf16_kernel<true>, dim3(gx), dim3(256), 0, st, a);
    return true;
}

}  // namespace dvae
