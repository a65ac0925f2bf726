// ConvTranspose2d k4 s2 p1, 128 -> 64 channels, on 16x16 maps (the decoders' first layer at 32x32 images), weights in
// registers.
//
// The transposed conv is four output-parity phases; phase (py, px) is a 2x2-tap conv of the input map into the output
// pixels (2 y + py, 2 x + px): K = 4 taps x 128 channels = 512, N = 64.  A workgroup is four waves, ONE PER PHASE, one
// per SIMD.  A wave holds its phase's whole weight slab -- 64 rows x 512 columns = 64 A fragments = 256 registers -- in the
// accumulator file for its life (inline-assembly MFMAs with an "a" operand, as conv_wreg.hpp), and computes, for a tile of
// 2 x 16 input positions, its 32 output pixels x 64 channels: 32 K steps of two MFMAs that share one 1 KB pixel fragment
// (0.5 KB of LDS per MFMA; the image-resident kernel of conv_image.hpp streams the weights through LDS, a barrier per
// 128-deep slab, and reads 1.5 KB per MFMA).  All four waves read the same staged input: the tile's 4 x 18 pixel halo
// (272-byte pixels, 5120-byte rows: the 16 lanes a ds_read_b128 serves together fall on 16 different slots).
//
// Between the MFMAs of a tile (fully unrolled, scheduling barriers keep the placement) a wave runs the PREVIOUS tile's
// epilogue (two accumulator sets alternate) into an LDS patch that interleaves the four phases into whole output rows
// -- a tile's output is four full rows of the 32-pixel-wide image: 16 KB contiguous -- and the staging of the tiles ahead:
// each of a thread's five halo vectors is stored for tile + 1 and at once re-loaded for tile + 2.  No memory operation of
// the tile loop is conditional.  One barrier per tile, then the drain of the previous patch.  Persistent over a contiguous
// run of tiles (whole images in order).
#pragma once
#include <type_traits>

#include "conv_wreg.hpp"

namespace dvae {

struct UpArgs {
    const void* in;          // bf16 [B][16][16][128]
    const void* wp;          // packed forward weights [64][Ktot], K = (phase, tap, channel)
    void* out;               // bf16 [B][32][32][64]
    const float* scale;      // per output channel, or null (1)
    const float* shift;      // per output channel, or null (0)
    int B, Ktot, relu;
    int tiles_per_wg;
    int koff[4];             // first K column of each phase
    signed char dy[4][4], dx[4][4];   // [phase][tap]: input offset of the tap
};

static constexpr int UP_PIX = 256 + 16;                        // bytes per staged input pixel
static constexpr int UP_ROW = 5120;                            // bytes per staged row (18 pixels, padded to a multiple of 256)
static constexpr int UP_HALO_BYTES = 4 * UP_ROW;
static constexpr int UP_OPIX = 128 + 8;                        // bytes per pixel of the output patch
static constexpr int UP_OROW = 32 * UP_OPIX + 8 + 8;           // ... per output row: an 8-byte gap after pixel 15 (bank spread)
static constexpr int UP_PATCH_BYTES = 4 * UP_OROW;
static constexpr int UP_WROW = 512 + 16;                       // bytes per row of a wave's weight window (256 columns)
static constexpr int UP_TILE_LDS = 2 * UP_HALO_BYTES + 2 * UP_PATCH_BYTES + 2 * 64 * 4;
static constexpr int UP_LDS_BYTES = 4 * 32 * UP_WROW > UP_TILE_LDS ? 4 * 32 * UP_WROW : UP_TILE_LDS;

__global__ __launch_bounds__(256) void conv_up_bf16_kernel(const UpArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* patch = smem + 2 * UP_HALO_BYTES;
    float* ssl = reinterpret_cast<float*>(smem + 2 * UP_HALO_BYTES + 2 * UP_PATCH_BYTES);      // scale[64], shift[64]
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, r = lane & 31, h = lane >> 5, orow = r >> 4, ox = r & 15;
    const int py = wave >> 1, px = wave & 1;
    const bf16* in = reinterpret_cast<const bf16*>(a.in);
    const bf16* wp = reinterpret_cast<const bf16*>(a.wp);
    const int ntiles = a.B * 8;
    const int t0 = blockIdx.x * a.tiles_per_wg;
    const int t1 = t0 + a.tiles_per_wg < ntiles ? t0 + a.tiles_per_wg : ntiles;
    if (t0 >= t1) return;

    // ---- weights: a wave moves its own phase's slab through a private LDS window, a quarter at a time (32 rows x 256
    // columns, copied as contiguous 512-byte row pieces), and takes its fragments from there: lane (r, h) supplies row
    // 32 j + r, K columns 16 st + 8 h .. of the phase's 512.  Straight-line and unconditional, so that the 64 fragments
    // are plain definitions the "a" operands can pin.
    u32x4 wreg[32][2];                                     // pinned to the accumulator file by the MFMA statements' "a"
    {
        unsigned char* win = smem + wave * (32 * UP_WROW);
        const unsigned char* src = reinterpret_cast<const unsigned char*>(wp + a.koff[wave]);   // uniform
        const int wrow = lane >> 5, wcol = lane & 31;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) {
                u32x4 wt[16];
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    wt[i] = *reinterpret_cast<const u32x4*>(src + (unsigned)((32 * j + 2 * i + wrow) * a.Ktot * 2 + 512 * kh + wcol * 16));
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    *reinterpret_cast<u32x4*>(win + (2 * i + wrow) * UP_WROW + wcol * 16) = wt[i];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                for (int s16 = 0; s16 < 16; ++s16)
                    wreg[16 * kh + s16][j] = *reinterpret_cast<const u32x4*>(win + r * UP_WROW + 32 * s16 + 16 * h);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
    }
    __syncthreads();                                       // the windows are dead: the halo buffers and patches take their place
    if (tid < 64) {
        ssl[tid] = a.scale ? a.scale[tid] : 1.f;
        ssl[64 + tid] = a.shift ? a.shift[tid] : 0.f;
    }

    // ---- staging: vector v = tid + 256 k of the 4 x 18 halo pixels (16 chunks each): pixel (tid >> 4) + 16 k, chunk tid & 15
    constexpr int HV = 5;
    const int chunk = tid & 15;
    int hrc[HV];                                           // halo row << 8 | halo column; -1: no such vector
#pragma unroll
    for (int k = 0; k < HV; ++k) {
        const int p = (tid >> 4) + 16 * k;
        hrc[k] = p < 72 ? ((p / 18) << 8 | (p % 18)) : -1;
    }
    u32x4 hreg[HV];
    unsigned hpad = 0;                                     // bit k: hreg[k] lies outside the map
    // Vector k of `tile` -> hreg[k].  Unconditional and clamped: a load inside a branch is waited for where the branch ends.
    auto load_vec = [&](int tile, int k) {
        const int b = tile >> 3, y0 = (tile & 7) * 2;
        const unsigned char* img = reinterpret_cast<const unsigned char*>(in) + (long)b * (16 * 16 * 256);   // uniform
        const int hr = (hrc[k] < 0 ? 0 : hrc[k]) >> 8, hc = (hrc[k] < 0 ? 0 : hrc[k]) & 255;
        const int iy = y0 - 1 + hr, ix = hc - 1;
        const bool pad = (unsigned)iy >= 16u || (unsigned)ix >= 16u;
        hpad = pad ? hpad | (1u << k) : hpad & ~(1u << k);
        const int cy = iy < 0 ? 0 : (iy > 15 ? 15 : iy), cx = ix < 0 ? 0 : (ix > 15 ? 15 : ix);
        hreg[k] = *reinterpret_cast<const u32x4*>(img + (unsigned)((cy * 16 + cx) * 256 + chunk * 16));
    };
    auto store_vec = [&](int buf, int k) {
        if (hrc[k] < 0) return;                            // (an LDS store in a branch costs nothing)
        u32x4 v = hreg[k];
        if (hpad >> k & 1u) v = u32x4{0u, 0u, 0u, 0u};
        *reinterpret_cast<u32x4*>(smem + buf * UP_HALO_BYTES + (hrc[k] >> 8) * UP_ROW + (hrc[k] & 255) * UP_PIX + chunk * 16) = v;
    };

    // ---- pixel fragment of tap t, slab s: chunk 2 s + h of input pixel (orow + dy, ox + dx); halo pixel (0, 0) is (-1, -1)
    int tbase[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
        tbase[t] = (orow + 1 + (int)a.dy[wave][t]) * UP_ROW + (ox + 1 + (int)a.dx[wave][t]) * UP_PIX + 16 * h;
    auto frag = [&](const unsigned char* hb, int st) -> u32x4 {
        return *reinterpret_cast<const u32x4*>(hb + tbase[st >> 3] + (st & 7) * 32);
    };

    // ---- epilogue of a finished accumulator set, group e = (j, g): channels 32 j + 8 g + 4 h .. of output pixel
    // (2 orow + py, 2 ox + px) -> the patch (4 output rows x 32 pixels; an 8-byte gap after pixel 15 of a row)
    const int pcol = 2 * ox + px;
    const int poff = (2 * orow + py) * UP_OROW + pcol * UP_OPIX + (pcol >> 4) * 8 + 8 * h;
    const short rl = (short)(a.relu ? 0 : -32768);
    typedef short i16x2_t __attribute__((ext_vector_type(2)));
    typedef bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    const i16x2_t relu_lo = {rl, rl};
    f32x4 esc, esh;
    auto epi_load = [&](int e) {
        const int nl = 32 * (e >> 2) + 8 * (e & 3) + 4 * h;
        esc = *reinterpret_cast<const f32x4*>(ssl + nl);
        esh = *reinterpret_cast<const f32x4*>(ssl + 64 + nl);
    };
    auto epi_store = [&](const f32x16 (&acc)[2], int pbuf, int e) {
        const int j = e >> 2, g = e & 3;
        unsigned o[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const f32x2_t f = {acc[j][4 * g + 2 * k] * esc[2 * k] + esh[2 * k], acc[j][4 * g + 2 * k + 1] * esc[2 * k + 1] + esh[2 * k + 1]};
            const bf16x2_t q = __builtin_convertvector(f, bf16x2_t);
            // ReLU on the rounded pair: a bf16 read as a signed 16-bit integer is negative exactly when the number is
            o[k] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(i16x2_t, q), relu_lo));
        }
        *reinterpret_cast<u32x2*>(patch + pbuf * UP_PATCH_BYTES + poff + (32 * j + 8 * g) * 2) = u32x2{o[0], o[1]};
    };
    auto drain = [&](int tile, int pbuf) {                 // the patch as whole pixels: four contiguous output rows
        const int b = tile >> 3, y0 = (tile & 7) * 2;
        unsigned char* ob = reinterpret_cast<unsigned char*>(a.out) + ((long)b * 32 + 2 * y0) * (32 * 128);
        const unsigned char* ps = patch + pbuf * UP_PATCH_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int v = tid + 256 * i, pix = v >> 3, c = v & 7, col = pix & 31;
            const unsigned char* s = ps + (pix >> 5) * UP_OROW + col * UP_OPIX + (col >> 4) * 8 + c * 16;
            const u32x2 lo = *reinterpret_cast<const u32x2*>(s), hi = *reinterpret_cast<const u32x2*>(s + 8);
            *reinterpret_cast<u32x4*>(ob + v * 16) = u32x4{lo[0], lo[1], hi[0], hi[1]};
        }
    };

    // One tile: 64 MFMAs into accN; between them the epilogue of accO (tile - 1) and, for each halo vector, its store
    // (tile + 1) and reload (tile + 2).  Nothing is conditional: the first tile's "previous" epilogue works on an undefined
    // accumulator and is drained to the first tile's own place, which the real drain overwrites one barrier later (same
    // thread, same addresses); the last tiles re-load the last tile and store a halo nobody reads.
    // (cur = parity of the tile's position in the run, a compile-time constant of each of the loop body's two calls: every
    // LDS address below is then a lane register plus an immediate)
    auto phase = [&](auto CUR, f32x16 (&accN)[2], const f32x16 (&accO)[2], int tile, bool first) {
        constexpr int cur = decltype(CUR)::value;
        const unsigned char* hb = smem + cur * UP_HALO_BYTES;
        const int tile2 = tile + 2 < t1 ? tile + 2 : t1 - 1;
        constexpr int PF = 2;                              // fragments are read PF steps ahead of their MFMAs
        u32x4 pv[PF + 1];
#pragma unroll
        for (int st = 0; st < PF; ++st) pv[st] = frag(hb, st);
#pragma unroll
        for (int st = 0; st < 32; ++st) {
            if (st + PF < 32) pv[(st + PF) % (PF + 1)] = frag(hb, st + PF);
            if (st == 0) {
                mfma_first_a(accN[0], wreg[st][0], pv[st % (PF + 1)]);
                mfma_first_a(accN[1], wreg[st][1], pv[st % (PF + 1)]);
            } else {
                mfma_acc_a(accN[0], wreg[st][0], pv[st % (PF + 1)]);
                mfma_acc_a(accN[1], wreg[st][1], pv[st % (PF + 1)]);
            }
            // side work of this step
            if (st % 4 == 1) epi_load(st / 4);
            if (st % 4 == 3) epi_store(accO, cur ^ 1, st / 4);
            if (st % 6 == 2 && st / 6 < HV) {
                store_vec(cur ^ 1, st / 6);
                load_vec(tile2, st / 6);
            }
            __builtin_amdgcn_sched_barrier(0);             // keep reads and side work where they were placed
        }
        __syncthreads();                                   // patch and next halo complete; everyone is done with this halo
        drain(first ? tile : tile - 1, cur ^ 1);
    };

    // first tile's halo, and the registers of the second
#pragma unroll
    for (int k = 0; k < HV; ++k) load_vec(t0, k);
#pragma unroll
    for (int k = 0; k < HV; ++k) {
        store_vec(0, k);
        load_vec(t0 + 1 < t1 ? t0 + 1 : t0, k);
    }
    __syncthreads();
    f32x16 accA[2], accB[2];
    int it = 0, tile = t0;
    bool last_a = true;
#pragma nounroll
    while (tile < t1) {
        phase(std::integral_constant<int, 0>{}, accA, accB, tile, it == 0);
        ++tile; ++it; last_a = true;
        if (tile >= t1) break;
        phase(std::integral_constant<int, 1>{}, accB, accA, tile, false);
        ++tile; ++it; last_a = false;
    }
    // the last tile's epilogue and drain (patch of iteration it - 1)
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");   // the last MFMAs' results -> the VALU reads below
    {
        const int pbuf = (it - 1) & 1;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            epi_load(e);
            if (last_a) epi_store(accA, pbuf, e); else epi_store(accB, pbuf, e);
        }
        __syncthreads();
        drain(t1 - 1, pbuf);
    }
}

// true if the plan is a launch of the kernel above
static bool up_applicable(const GemmPlan& P) {
    if (P.nphase != 4 || P.CgS != 128 || P.Cg != 128 || P.N != 64 || P.NS != 64) return false;
    if (P.Hin != 16 || P.Win != 16 || P.Hout != 32 || P.Wout != 32 || P.Ktot != 2048) return false;
    for (int i = 0; i < 4; ++i) {
        const Phase& ph = P.ph[i];
        if (ph.ntaps != 4 || ph.SY != 1 || ph.SX != 1 || ph.DSY != 2 || ph.DSX != 2 || ph.D0Y != (i >> 1) || ph.D0X != (i & 1) ||
            ph.Hg != 16 || ph.Wg != 16 || ph.koff != 512 * i)
            return false;
        for (int t = 0; t < 4; ++t)
            if (ph.dy[t] < -1 || ph.dy[t] > 1 || ph.dx[t] < -1 || ph.dx[t] > 1) return false;
    }
    return (long)P.B * 8 < 0x7fffffffL;
}

static bool try_launch_up_bf16(const GemmPlan& P, const void* in, const void* wp, void* out, const float* scale,
                               const float* shift, const void* res, int relu, int accumulate, int out_f32, hipStream_t st) {
    if (res || accumulate || out_f32 || !up_applicable(P)) return false;
    UpArgs a;
    a.in = in; a.wp = wp; a.out = out; a.scale = scale; a.shift = shift;
    a.B = P.B; a.Ktot = P.Ktot; a.relu = relu;
    for (int i = 0; i < 4; ++i) {
        a.koff[i] = P.ph[i].koff;
        for (int t = 0; t < 4; ++t) { a.dy[i][t] = P.ph[i].dy[t]; a.dx[i][t] = P.ph[i].dx[t]; }
    }
    const int ntiles = P.B * 8;
    a.tiles_per_wg = (ntiles + 255) / 256;                 // one workgroup per CU (a wave per SIMD with all its registers)
    const int gx = (ntiles + a.tiles_per_wg - 1) / a.tiles_per_wg;
    static LdsOptIn lds_opt;
    lds_opt.ensure(reinterpret_cast<const void*>(&conv_up_bf16_kernel), UP_LDS_BYTES);
    hipLaunchKernelGGL(conv_up_bf16_kernel, dim3(gx), dim3(256), UP_LDS_BYTES, st, a);
    return true;
}

}  // namespace dvae
