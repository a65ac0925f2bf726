// 3x3 stride-2 bf16 convolution, 64 -> 128 channels (the first conv of a down-sampling residual block), weights in
// registers.
//
// A workgroup of 4 waves owns tiles of 2 x 16 output pixels.  Wave cq holds the weights of output channels
// [32 cq, 32 cq + 32) for all nine taps as 36 MFMA A-operand fragments (144 registers, loaded once through LDS so that the
// global reads are coalesced), and computes them for the tile's 32 pixels: one v_mfma_f32_32x32x16_bf16 per (tap,
// 16-channel slab), whose only LDS traffic is the 1 KB pixel fragment.  Two workgroups share a CU: they drift apart, so
// the epilogue, staging and barriers of one run beside the MFMAs of the other (one 8-wave workgroup per CU, whose waves
// reach every phase together, measured 52 us per 1024 images of which 17 were those phases with the matrix pipe idle).
//
// The 5 x 33 input pixels under a tile are staged in LDS once (zero outside the image) and every tap re-addresses that
// copy: lane (output row o, column x) reads input pixel (2 o + ky, 2 x + kx).  Consecutive lanes are two pixels = 256 B
// apart, one whole bank row, so the image is swizzled twice: the eight 16-byte chunks of a pixel are XORed with
// (column >> 1) & 7, and rows 2-3 start one pixel late, which moves the second output row of a lane group to the
// other half of the bank row: the 16 lanes a ds_read_b128 serves together then hit 16 different slots.
//
// A wave's side work rides between its MFMAs, a piece every few steps of the fully unrolled 36-step loop (placed by hand,
// held in place by scheduling barriers): the PREVIOUS tile's epilogue (two accumulator sets alternate; scale / shift /
// ReLU, rounding, into an LDS patch so that a tile leaves as whole 256-byte pixels: two contiguous 4 KB rows), and the
// staging of the tiles ahead -- each of a thread's six halo vectors is written to the other halo buffer (next tile) and
// at once re-loaded from memory for the tile after, so a load has a whole tile period to land.  One barrier per tile,
// followed by the drain of the patch.  With everything serial (MFMAs, then epilogue, then staging) the launch took
// 47 us per 1024 images at 43 % matrix-pipe and 43 % LDS occupancy: the waves of a CU run the same code in step, so a
// second wave per SIMD overlaps nothing by itself.  Workgroups are persistent over a contiguous run of tiles (whole
// images in order, so the input rows two tiles share are found in L2).
#pragma once
#include <type_traits>

#include "conv_halo.hpp"

namespace dvae {

struct DownArgs {
    const void* in;          // bf16 [B][Hin][Win][64]
    const void* wp;          // packed forward weights [128][Ktot], K = (tap, channel)
    void* out;               // bf16 [B][Hout][Wout][128]
    const float* scale;      // per output channel, or null (1)
    const float* shift;      // per output channel, or null (0)
    int B, Hin, Win, Hout, Wout, Ktot, relu;
    int tiles_per_wg;
    TileGrid g;              // tiles of an image: Wout / 16 per tile row, Hout / 2 tile rows
};

static constexpr int DOWN_RP = 34;                              // pixels per staged row (33 + the one-pixel shift)
static constexpr int DOWN_HALO_BYTES = 5 * DOWN_RP * 128;
static constexpr int DOWN_PITCH = 256 + 16;                     // bytes per pixel of the output patch
static constexpr int DOWN_PATCH_BYTES = 32 * DOWN_PITCH;
static constexpr int DOWN_WROW = 576 * 2 + 16;                  // bytes per staged weight row (the pad spreads the banks)
static constexpr int DOWN_TILE_LDS = 2 * DOWN_HALO_BYTES + 2 * DOWN_PATCH_BYTES + 2 * 128 * 4;
static constexpr int DOWN_LDS_BYTES = 64 * DOWN_WROW > DOWN_TILE_LDS ? 64 * DOWN_WROW : DOWN_TILE_LDS;   // the prologue stages 64 weight rows at a time

// WIDE: images wider than one tile (the left halo column holds pixels of the neighbouring tile)
template <bool WIDE>
__global__ __launch_bounds__(256, 2) void conv_down_bf16_kernel(const DownArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* patch = smem + 2 * DOWN_HALO_BYTES;
    float* ssl = reinterpret_cast<float*>(smem + 2 * DOWN_HALO_BYTES + 2 * DOWN_PATCH_BYTES);   // scale[128], shift[128]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int cq = wave, r = lane & 31, h = lane >> 5, orow = r >> 4, ox = r & 15;
    const bf16* in = reinterpret_cast<const bf16*>(a.in);
    const bf16* wp = reinterpret_cast<const bf16*>(a.wp);
    const int ntiles = a.B * a.g.tpi;
    const int t0 = blockIdx.x * a.tiles_per_wg;
    const int t1 = t0 + a.tiles_per_wg < ntiles ? t0 + a.tiles_per_wg : ntiles;
    if (t0 >= t1) return;

    // ---- this wave's weights: lane (r, h) supplies row 32 cq + r, K columns 8 h .. 8 h + 7 of every 16-deep step.
    // Read straight from memory a fragment load touches 32 rows (36 such loads per wave: 6-7 us of the launch); instead
    // the rows go through LDS 64 at a time, copied as contiguous 16-byte vectors.
    bf16x8 wf[36];
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
        const bf16* src = wp + (long)pass * 64 * 576;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            u32x4 wt[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) wt[i] = *reinterpret_cast<const u32x4*>(src + (tid + 256 * (9 * sub + i)) * 8);
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                const int f = tid + 256 * (9 * sub + i), row = f / 72, col = f - row * 72;
                *reinterpret_cast<u32x4*>(smem + row * DOWN_WROW + col * 16) = wt[i];
            }
        }
        __syncthreads();
        if ((cq >> 1) == pass) {
            const unsigned char* wrow = smem + (32 * (cq & 1) + r) * DOWN_WROW + 16 * h;
#pragma unroll
            for (int st = 0; st < 36; ++st) wf[st] = *reinterpret_cast<const bf16x8*>(wrow + 32 * st);
        }
        __syncthreads();
    }
    if (tid < 128) {
        ssl[tid] = a.scale ? a.scale[tid] : 1.f;
        ssl[128 + tid] = a.shift ? a.shift[tid] : 0.f;
    }

    // ---- staging: thread (tid >> 3, chunk tid & 7) takes its vector of each of the five rows.  One tile per image row
    // (!WIDE): the 32 threads columns are the image's 32 columns, halo columns 1 .. 32; halo column 0 is padding, zeroed
    // once.  WIDE: halo columns 0 .. 31, and the first 40 threads also take column 32 of row tid >> 3 (no branches around
    // the loads and stores of the tile loop: the other threads' sixth vector is their fifth again).
    constexpr int HV = WIDE ? 6 : 5;
    const int chunk = tid & 7, scol = (tid >> 3) + (WIDE ? 0 : 1);
    const bool extra = tid < 40;
    const int erow = extra ? (tid >> 3) : 4, ecol = extra ? 32 : scol;
    const int lds_main = scol * 128 + ((chunk ^ ((scol >> 1) & 7)) * 16);      // + (row * RP + row shift) * 128
    const int lds_extra = (erow * DOWN_RP + ((erow >> 1) & 1) + ecol) * 128 + ((chunk ^ ((ecol >> 1) & 7)) * 16);
    if (!WIDE && tid < 80) {                                   // halo column 0 of both buffers: chunk tid & 7 of row (tid >> 3) % 5
        const int row = (tid >> 3) % 5, buf = (tid >> 3) / 5;
        *reinterpret_cast<u32x4*>(smem + buf * DOWN_HALO_BYTES + (row * DOWN_RP + ((row >> 1) & 1)) * 128 + chunk * 16) = u32x4{0u, 0u, 0u, 0u};
    }
    u32x4 hreg[HV];
    unsigned hpad = 0;                                         // bit k: hreg[k] lies outside the image
    // Vector k of `tile` -> hreg[k].  Unconditional and clamped: a load inside a branch is waited for where the branch
    // ends.  (The far edges are inside the image: Hin = 2 Hout, Win = 2 Wout.)
    const unsigned lane_off = (unsigned)chunk * 16u;           // bytes
    auto load_vec = [&](int tile, int k) {
        int b, ty, tx;
        a.g.split(tile, b, ty, tx);
        const unsigned char* img = reinterpret_cast<const unsigned char*>(in) + (long)b * a.Hin * a.Win * 128;   // uniform
        const int iy0 = 4 * ty - 1, ix0 = 32 * tx - 1;
        const int iy = iy0 + (k < 5 ? k : erow), ix = ix0 + (k < 5 ? scol : ecol);
        const bool pad = WIDE ? (iy < 0 || ix < 0) : iy < 0;
        hpad = pad ? hpad | (1u << k) : hpad & ~(1u << k);
        const unsigned off = (unsigned)((iy < 0 ? 0 : iy) * a.Win + (ix < 0 ? 0 : ix)) * 128u + lane_off;
        hreg[k] = *reinterpret_cast<const u32x4*>(img + off);
    };
    auto store_vec = [&](int buf, int k) {
        unsigned char* hb = smem + buf * DOWN_HALO_BYTES;
        u32x4 v = hreg[k];
        if (hpad >> k & 1u) v = u32x4{0u, 0u, 0u, 0u};
        if (k < 5) *reinterpret_cast<u32x4*>(hb + lds_main + (k * DOWN_RP + ((k >> 1) & 1)) * 128) = v;
        else *reinterpret_cast<u32x4*>(hb + lds_extra) = v;
    };

    // ---- pixel fragment of tap (ky, kx), slab s: chunk 2 s + h of input pixel (2 orow + ky, 2 ox + kx)
    // = base[kx == 2] ^ (ky == 2 ? 128 : 0) ^ (32 s), + (ky * RP + kx) * 128.  The chunk swizzle depends on
    // (ox + (kx >> 1)) & 7: two lane-dependent bases.  The row shift (orow + (ky >> 1)) & 1 is bit 7 of the address (the
    // other terms of the pixel index are even), flipped for ky = 2; the slab's 32 s falls on bits 5-6, where the bases
    // hold the swizzled chunk; everything else is immediate.
    int base[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
        base[j] = (2 * orow * DOWN_RP + (orow & 1) + 2 * ox) * 128 + ((h ^ ((ox + j) & 7)) * 16);
    auto frag = [&](const unsigned char* hb, int st) -> bf16x8 {
        const int t = st >> 2, ky = t / 3, kx = t % 3;
        int bs = base[kx >> 1];
        asm volatile("" : "+v"(bs));                           // the XOR is made here: hoisted addresses cost the prefetch its registers
        return *reinterpret_cast<const bf16x8*>(hb + (bs ^ ((st & 3) * 32 + (ky >> 1) * 128)) + (ky * DOWN_RP + kx) * 128);
    };

    // ---- epilogue of a finished accumulator set, group g: channels 32 cq + 8 g + 4 h .. of pixel r, in two pieces (the
    // LDS operands are read a few steps ahead of the arithmetic)
    f32x4 esc, esh;
    auto epi_load = [&](int g) {
        const int nl = 32 * cq + 8 * g + 4 * h;
        esc = *reinterpret_cast<const f32x4*>(ssl + nl);
        esh = *reinterpret_cast<const f32x4*>(ssl + 128 + nl);
    };
    auto epi_store = [&](const f32x16& acc, int pbuf, int g) {
        const int nl = 32 * cq + 8 * g + 4 * h;
        bf16x4 q;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float f = acc[4 * g + k] * esc[k] + esh[k];
            if (a.relu) f = fmaxf(f, 0.f);
            q[k] = (bf16)f;
        }
        *reinterpret_cast<bf16x4*>(patch + pbuf * DOWN_PATCH_BYTES + r * DOWN_PITCH + nl * 2) = q;
    };
    auto drain = [&](int tile, int pbuf) {                    // the patch as whole pixels: two contiguous 4 KB rows
        int b, ty, tx;
        a.g.split(tile, b, ty, tx);
        bf16* ob = reinterpret_cast<bf16*>(a.out) + (((long)b * a.Hout + 2 * ty) * a.Wout + 16 * tx) * 128;
        const unsigned char* ps = patch + pbuf * DOWN_PATCH_BYTES;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int j = tid + 256 * i, px = j >> 4, cv = j & 15;
            const u32x4 v = *reinterpret_cast<const u32x4*>(ps + px * DOWN_PITCH + cv * 16);
            *reinterpret_cast<u32x4*>(ob + ((px >> 4) * a.Wout + (px & 15)) * 128 + cv * 8) = v;
        }
    };

    // One tile: 36 MFMAs into accN; between them the epilogue of accO (tile - 1, if any) and, for each halo vector, its
    // store (tile + 1) and reload (tile + 2).
    // (cur = parity of the tile's position in the run, a compile-time constant of each of the loop body's two calls)
    auto phase = [&](auto CUR, f32x16& accN, const f32x16& accO, int tile, bool first) {
        constexpr int cur = decltype(CUR)::value;
        const unsigned char* hb = smem + cur * DOWN_HALO_BYTES;
        // Nothing below is conditional (a memory operation inside a branch is waited for where the branch ends).  The
        // first tile has no predecessor: its "previous" epilogue works on an undefined accumulator and is drained to the
        // first tile's own place, which the real drain overwrites one barrier later (same thread, same addresses).  The
        // last tiles re-load the last tile and store a halo nobody reads.
        const int tile2 = tile + 2 < t1 ? tile + 2 : t1 - 1;
#pragma unroll
        for (int e = 0; e < 16; ++e) accN[e] = 0.f;
        constexpr int PF = WIDE ? 1 : 2;                       // fragments are read PF steps ahead of their MFMA (WIDE: the sixth halo vector takes the registers)
        bf16x8 pv[PF + 1];
#pragma unroll
        for (int st = 0; st < PF; ++st) pv[st] = frag(hb, st);
#pragma unroll
        for (int st = 0; st < 36; ++st) {
            if (st + PF < 36) pv[(st + PF) % (PF + 1)] = frag(hb, st + PF);
            accN = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[st], pv[st % (PF + 1)], accN, 0, 0, 0);
            // side work of this step
            if (st % 8 == (WIDE ? 5 : 2)) epi_load(st / 8);
            if (st % 8 == 6) epi_store(accO, cur ^ 1, st / 8);
            if (st % 4 == 0 && st / 4 < HV) {
                store_vec(cur ^ 1, st / 4);
                load_vec(tile2, st / 4);
            }
            __builtin_amdgcn_sched_barrier(0);                 // keep reads and side work where they were placed
        }
        __syncthreads();                                       // patch and next halo complete; everyone is done with this halo
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
    f32x16 accA, accB;
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
    {
        const int pbuf = (it - 1) & 1;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            epi_load(g);
            if (last_a) epi_store(accA, pbuf, g); else epi_store(accB, pbuf, g);
        }
        __syncthreads();
        drain(t1 - 1, pbuf);
    }
}

// true if the plan is a launch of the kernel above
static bool down_applicable(const GemmPlan& P) {
    if (P.nphase != 1 || P.CgS != 64 || P.Cg != 64 || P.N != 128 || P.NS != 128 || P.Ktot != 576) return false;
    const Phase& ph = P.ph[0];
    if (ph.SY != 2 || ph.SX != 2 || ph.DSY != 1 || ph.DSX != 1 || ph.D0Y || ph.D0X || ph.ntaps != 9) return false;
    if (ph.Hg != P.Hout || ph.Wg != P.Wout || P.Hin != 2 * P.Hout || P.Win != 2 * P.Wout) return false;
    if (P.Wout % 16 || P.Hout % 2) return false;
    for (int t = 0; t < 9; ++t)
        if (ph.dy[t] != t / 3 - 1 || ph.dx[t] != t % 3 - 1) return false;
    if ((long)P.Hin * P.Win * 64 > 0x3fffffffL) return false;  // per-image offsets are 32-bit
    TileGrid tg;
    return make_tile_grid(P.Wout / 16, P.Hout / 2, (long)P.B * (P.Wout / 16) * (P.Hout / 2), tg);
}

static bool try_launch_down_bf16(const GemmPlan& P, const void* in, const void* wp, void* out, const float* scale,
                                 const float* shift, const void* res, int relu, int accumulate, int out_f32, hipStream_t st) {
    if (res || accumulate || out_f32 || !down_applicable(P)) return false;
    DownArgs a;
    a.in = in; a.wp = wp; a.out = out; a.scale = scale; a.shift = shift;
    a.B = P.B; a.Hin = P.Hin; a.Win = P.Win; a.Hout = P.Hout; a.Wout = P.Wout; a.Ktot = P.Ktot; a.relu = relu;
    make_tile_grid(P.Wout / 16, P.Hout / 2, 0, a.g);
    const int ntiles = P.B * a.g.tpi;
    a.tiles_per_wg = (ntiles + 511) / 512;                 // two workgroups per CU
    const int gx = (ntiles + a.tiles_per_wg - 1) / a.tiles_per_wg;
    if (P.Wout == 16) {
        static LdsOptIn lds_opt;
        lds_opt.ensure(reinterpret_cast<const void*>(&conv_down_bf16_kernel<false>), DOWN_LDS_BYTES);
        hipLaunchKernelGGL(conv_down_bf16_kernel<false>, dim3(gx), dim3(256), DOWN_LDS_BYTES, st, a);
    } else {
        static LdsOptIn lds_opt;
        lds_opt.ensure(reinterpret_cast<const void*>(&conv_down_bf16_kernel<true>), DOWN_LDS_BYTES);
        hipLaunchKernelGGL(conv_down_bf16_kernel<true>, dim3(gx), dim3(256), DOWN_LDS_BYTES, st, a);
    }
    return true;
}

}  // namespace dvae
