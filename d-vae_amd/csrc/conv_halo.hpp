// 3x3 (radius-1) stride-1 bf16 convolution with the input tile resident in LDS.
//
// The gather GEMM of igemm.hpp re-reads every input pixel once per tap (9x for a 3x3 conv) through
// L2; at bf16 MFMA rates that traffic, not the matrix pipe, bounds it.  Here one workgroup owns a
// TH x TW patch of output pixels (256 pixels) of one image, stages the (TH+2) x (TW+2) input halo in
// LDS once (zero-filled outside the image), and forms the pixel fragment of every tap by re-addressing
// that tile: lane r of a 32-pixel MFMA block reads halo pixel (py+dy, px+dx) -- consecutive
// lanes hit consecutive 144-byte (64 ch) / 80-byte (32 ch) pixel rows, conflict-free for ds_read_b128.
//
// Workgroups are persistent (one per CU, 8 waves).  The weights of all nine taps stay in LDS for the
// workgroup's lifetime (unpadded rows, 16-byte chunks XOR-swizzled by the row index so that the 32 rows
// a fragment read touches spread over all banks).  A CU's vector-memory path moves 64 B/clk, so the ~110 KB a
// tile reads and writes would cost as much as its MFMAs if issued in a burst; instead every global access
// of a tile is issued from inside the fully unrolled MFMA loop, one instruction every few MFMAs:
//   - the previous tile's output, parked as bf16 in the wave's private LDS patch, is drained to HBM as whole
//     128-byte pixel rows (the MFMA leaves a lane with 4-channel quads of one pixel; the patch transposes);
//   - this tile's residual / old output rows are fetched (coalesced), to be re-read from the patch in
//     register layout after the loop and added in fp32, so the result is rounded once;
//   - the next tile's halo is fetched into registers.
// A tile therefore costs two barriers, no weight traffic, and no burst on the memory path.
// The MFMA is issued as D[channel][pixel] = W * X^T (lane = pixel, register quad = 4 consecutive channels).
// Serves every single-phase 9-tap radius-1 unit-stride plan: Conv2d 3x3 s1 forward and input-gradient,
// ConvTranspose2d 3x3 s1 forward and input-gradient (the plan supplies the taps).
#pragma once
#include "igemm.hpp"

namespace dvae {

// Tile grid of one image: ntx tiles per tile row, tpi tiles per image; tile index = (image * nty + ty) * ntx + tx.
// Dividing by the two counts is a multiply-high by ceil(2^32 / d): exact for every dividend where d is a power of two
// (32x32 and 64x64 images), otherwise for dividends below 2^32 / d (make_tile_grid checks the largest one).
struct TileGrid {
    int ntx, tpi;
    int one_ntx, one_tpi;    // 1 if the divisor is 1 (its multiplier would be 2^32), else 0
    unsigned inv_ntx, inv_tpi;
    // Branch-free on purpose: a select between a shift and a multiply-high compiles to branches, and a branch inside a
    // kernel's unrolled tile loop ends the basic block its side work is scheduled in.
    __device__ __forceinline__ int div_ntx(int n) const { return (int)__umulhi((unsigned)n, inv_ntx) + n * one_ntx; }
    __device__ __forceinline__ int div_tpi(int n) const { return (int)__umulhi((unsigned)n, inv_tpi) + n * one_tpi; }
    // tile -> image and tile coordinates inside it
    __device__ __forceinline__ void split(int tile, int& b, int& ty, int& tx) const {
        b = div_tpi(tile);
        const int rem = tile - b * tpi;
        ty = div_ntx(rem);
        tx = rem - ty * ntx;
    }
};
// false if the grid's divisions would not be exact for `max_dividend`
static bool make_tile_grid(int ntx, int nty, long max_dividend, TileGrid& g) {
    g.ntx = ntx; g.tpi = ntx * nty;
    g.one_ntx = ntx == 1; g.one_tpi = g.tpi == 1;
    g.inv_ntx = ntx > 1 ? (unsigned)(((1ull << 32) + ntx - 1) / ntx) : 0u;
    g.inv_tpi = g.tpi > 1 ? (unsigned)(((1ull << 32) + g.tpi - 1) / g.tpi) : 0u;
    return max_dividend >= 0 && max_dividend * g.tpi < (1ll << 32) && max_dividend < 0x7fffffffL;
}

struct HaloArgs {
    const void* in;
    const void* wp;
    void* out;
    const float* scale;
    const float* shift;
    const void* add;         // bf16 tensor shaped like the output: residual (added before the ReLU) or the old output
    const float* in_ss;      // optional input normalisation [B][2][CIN]: x <- relu?(x * scale + shift) while staging
    int in_relu;
    int B, H, W, N, NS, Ktot;
    int relu, add_after_relu, out_f32;
    TileGrid g;              // tiles of an image
    signed char dy[16], dx[16];
    float* stats;            // optional [B][tiles per image * 8][2][NS]: sum and sum of squares of the stored bf16 output
                             // over each wave's 32 pixels (for the normalisation that follows the conv), or null
    unsigned long long* stamps;   // profiling aid: [grid.x][64 tiles][8] cycle-counter samples of wave 0, or null
    int rev;                 // conv_wreg, whole-image walks: a workgroup takes its images last to first
};

// One cycle-counter sample per phase boundary of a tile (wave 0, lane 0 of each workgroup; first 64 tiles).
#define HALO_STAMP(i)                                                                               \
    do {                                                                                            \
        if (a.stamps && tid == 0 && it < 64)                                                        \
            a.stamps[((long)blockIdx.x * 64 + it) * 8 + (i)] = __builtin_readcyclecounter();        \
    } while (0)

// value of `x` in the lane selected by a DPP control word (row-local permutations; all lanes active)
template <int CTRL>
__device__ __forceinline__ float dpp_f(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}

// IN_SS / STATS: compiled-in support for the input-normalisation prologue / the output statistics (their registers
// are only paid for where a launch uses them).
template <int CIN, int NT, int TW, bool IN_SS, bool STATS>
__global__ __launch_bounds__(512) void conv_halo_bf16_kernel(const HaloArgs a) {
    constexpr int TH = 256 / TW, HWD = TW + 2, HP = (TH + 2) * HWD;
    constexpr int RS = CIN * 2 + 16;          // bytes per staged halo pixel (16-byte pad spreads banks)
    constexpr int WR = CIN * 2;               // bytes per weight row (swizzled, unpadded)
    constexpr int VPP = CIN * 2 / 16;         // 16-byte vectors per pixel / weight row
    constexpr int WSH = (CIN == 64) ? 1 : 2;  // weight rows per 256-byte bank row, log2: swizzle key = (row >> WSH) & (VPP - 1)
    constexpr int HALO_BYTES = HP * RS;
    constexpr int WTAP = NT * WR;             // one tap's weight slab
    constexpr int TN = NT / 32, KS = CIN / 16, NSTEP = 9 * KS;
    constexpr int HV = (HP * VPP + 511) / 512;
    // epilogue patch of one wave: 32 pixels x NT channels bf16, padded rows
    constexpr int SRS = NT * 2 + 16, VPS = NT / 8, PPI = 64 / VPS, NI = 32 / PPI;
    // global accesses of a tile, spread over the MFMA steps: NI+1 drain slots, NI add fetches, HV halo fetches
    constexpr int NOPS = 2 * NI + 1 + HV, OPSTRIDE = NSTEP / NOPS;
    static_assert(OPSTRIDE >= 1, "more memory operations than MFMA steps");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* wlds = smem;                                  // [9][NT][WR]
    unsigned char* halo = smem + 9 * WTAP;                       // [HP][RS]
    float* ssl = reinterpret_cast<float*>(smem + 9 * WTAP + HALO_BYTES);   // scale[NT], shift[NT]

    const int tid = threadIdx.x;
    const int n0 = blockIdx.y * NT;
    const bf16* in = reinterpret_cast<const bf16*>(a.in);
    const bf16* wp = reinterpret_cast<const bf16*>(a.wp);
    const int ntiles = a.B * a.g.tpi;

    // ---- weights (every tap) and the epilogue's per-channel scale/shift -> LDS, once -----------------
    // (every load issued before the first LDS write: a load-store loop waits out one memory latency per round)
    float ssl_sc = 0.f, ssl_sh = 0.f;
    if (tid < NT) {
        const int n = n0 + tid;
        const bool live = n < a.N;
        ssl_sc = (live && a.scale) ? a.scale[n] : (live ? 1.f : 0.f);
        ssl_sh = (live && a.shift) ? a.shift[n] : 0.f;
    }
    {
        constexpr int WV = (9 * NT * VPP + 511) / 512;
        u32x4 wtmp[WV];
#pragma unroll
        for (int i = 0; i < WV; ++i) {
            const int v = tid + 512 * i, t = v / (NT * VPP), rem = v % (NT * VPP), n = rem / VPP, cv = rem % VPP;
            if (v < 9 * NT * VPP) wtmp[i] = *reinterpret_cast<const u32x4*>(wp + (long)(n0 + n) * a.Ktot + t * CIN + cv * 8);
        }
#pragma unroll
        for (int i = 0; i < WV; ++i) {
            const int v = tid + 512 * i, t = v / (NT * VPP), rem = v % (NT * VPP), n = rem / VPP, cv = rem % VPP;
            if (v < 9 * NT * VPP) *reinterpret_cast<u32x4*>(wlds + t * WTAP + n * WR + ((cv ^ ((n >> WSH) & (VPP - 1))) * 16)) = wtmp[i];
        }
    }
    if (tid < NT) { ssl[tid] = ssl_sc; ssl[NT + tid] = ssl_sh; }
    int toff[9];                                          // LDS byte offset of each tap relative to the centre pixel
#pragma unroll
    for (int t = 0; t < 9; ++t) toff[t] = ((int)a.dy[t] * HWD + (int)a.dx[t]) * RS;

    // ---- this thread's share of a halo tile: HV 16-byte vectors; everything that does not depend on the tile is fixed here
    u32x4 hreg[HV];
    int hrel[HV], hyx[HV];                                // element offset from the tile's origin pixel; (row << 16 | col) in the halo
    static_assert(512 % VPP == 0, "a thread keeps one channel group across its halo vectors");
#pragma unroll
    for (int k = 0; k < HV; ++k) {
        const int v = tid + 512 * k, hp = v / VPP, cv = v % VPP;
        const int hy = hp / HWD, hx = hp % HWD;
        hrel[k] = ((hy - 1) * a.W + (hx - 1)) * CIN + cv * 8;
        hyx[k] = (v < HP * VPP) ? (hy << 16 | hx) : -1;
    }
    unsigned inimg = 0;                                   // bit k: hreg[k] is a pixel inside the image
    f32x4 isc[2], ish[2];                                 // input normalisation of this thread's 8 channels
    auto tile_origin = [&](int tile, int& y0, int& x0, int& b) {
        int ty, tx;
        a.g.split(tile, b, ty, tx);
        x0 = tx * TW; y0 = ty * TH;
    };
    auto load_halo_vec = [&](int tile, int k) {           // k-th vector of `tile`'s halo -> hreg[k]
        int y0, x0, b;
        tile_origin(tile, y0, x0, b);
        if (k == 0) {
            inimg = 0;
            if (IN_SS && a.in_ss) {
                const float* p = a.in_ss + (long)b * 2 * CIN + (tid % VPP) * 8;
                isc[0] = *reinterpret_cast<const f32x4*>(p); isc[1] = *reinterpret_cast<const f32x4*>(p + 4);
                ish[0] = *reinterpret_cast<const f32x4*>(p + CIN); ish[1] = *reinterpret_cast<const f32x4*>(p + CIN + 4);
            }
        }
        u32x4 val = {0u, 0u, 0u, 0u};
        const int iy = y0 - 1 + (hyx[k] >> 16), ix = x0 - 1 + (hyx[k] & 0xffff);
        if (hyx[k] >= 0 && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W) {
            val = *reinterpret_cast<const u32x4*>(in + ((long)(b * a.H + y0) * a.W + x0) * CIN + hrel[k]);
            inimg |= 1u << k;
        }
        hreg[k] = val;
    };
    auto store_halo = [&]() {
#pragma unroll
        for (int k = 0; k < HV; ++k) {
            const int v = tid + 512 * k;
            if (v >= HP * VPP) continue;
            u32x4 val = hreg[k];
            if (IN_SS && a.in_ss && (inimg >> k & 1u)) {  // zero padding stays zero: it pads the normalised tensor
                bf16x8 x = *reinterpret_cast<bf16x8*>(&val);
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    float f = (float)x[c] * isc[c >> 2][c & 3] + ish[c >> 2][c & 3];
                    if (a.in_relu) f = fmaxf(f, 0.f);
                    x[c] = (bf16)f;
                }
                val = *reinterpret_cast<u32x4*>(&x);
            }
            *reinterpret_cast<u32x4*>(halo + (v / VPP) * RS + (v % VPP) * 16) = val;
        }
    };

    const int wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    // wave w owns pixels [32w, 32w+32) of the tile; lane r is pixel 32w + r
    const int pl = wave * 32 + r;
    const unsigned char* hrow = halo + ((pl / TW + 1) * HWD + (pl % TW) + 1) * RS + 16 * h;
    int woff[KS];                                         // this lane's swizzled chunk of weight row r, per 16-channel slab
#pragma unroll
    for (int sl = 0; sl < KS; ++sl) woff[sl] = r * WR + (((2 * sl + h) ^ ((r >> WSH) & (VPP - 1))) * 16);
    unsigned char* stg = smem + 9 * WTAP + HALO_BYTES + 2 * NT * 4 + wave * (32 * SRS);
    const int sp = lane / VPS, sc8 = lane % VPS;          // coalesced side: lane = (pixel sp of a PPI group, 8-channel chunk sc8)
    const bool chunk_live = n0 + sc8 * 8 < a.NS;
    auto out_off = [&](int tile, int i) -> long {         // element offset of this lane's 16-byte piece i of the output tile
        int y0, x0, b;
        tile_origin(tile, y0, x0, b);
        const int p = wave * 32 + i * PPI + sp;
        return ((long)(b * a.H + y0 + p / TW) * a.W + x0 + p % TW) * a.NS + n0 + sc8 * 8;
    };
    auto wave_lds_fence = [&]() {                          // LDS hand-over between lanes of one wave: order, no barrier
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    // output statistics of this wave's 32 pixels, gathered while the patch is drained (coalesced layout: 8 channels per lane)
    float ssum[8], ssq[8];
    auto stats_add = [&](const u32x4& piece) {
        const bf16x8 x = *reinterpret_cast<const bf16x8*>(&piece);
#pragma unroll
        for (int c = 0; c < 8; ++c) { const float f = (float)x[c]; ssum[c] += f; ssq[c] += f * f; }
    };
    auto stats_flush = [&](int tile) {
        // Sum over the lanes that hold the same 8 channels (VPS apart): DPP butterflies inside a 16-lane row (no LDS
        // queue traffic), cross-lane shuffles across rows.
        float v[16];
#pragma unroll
        for (int c = 0; c < 8; ++c) { v[c] = ssum[c]; v[8 + c] = ssq[c]; }
        if (VPS == 4) {
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] += dpp_f<0x1B>(dpp_f<0x141>(v[i]));      // lane ^ 4: half-row mirror, then quad reverse
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] += dpp_f<0x128>(v[i]);                        // lane ^ 8: rotate the row by 8
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] += __shfl_xor(v[i], 16, 64);
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] += __shfl_xor(v[i], 32, 64);
        if (sp == 0 && chunk_live) {
            float* dst = a.stats + (((long)a.g.div_tpi(tile) * 8 + wave) * 2) * a.NS + n0 + sc8 * 8;
            f32x4 v0 = {v[0], v[1], v[2], v[3]}, v1 = {v[4], v[5], v[6], v[7]};
            f32x4 q0 = {v[8], v[9], v[10], v[11]}, q1 = {v[12], v[13], v[14], v[15]};
            *reinterpret_cast<f32x4*>(dst) = v0; *reinterpret_cast<f32x4*>(dst + 4) = v1;
            *reinterpret_cast<f32x4*>(dst + a.NS) = q0; *reinterpret_cast<f32x4*>(dst + a.NS + 4) = q1;
        }
    };
    // operand fragments of flattened step st = (tap, 16-channel slab): one pixel vector, TN weight vectors
    auto frag = [&](int st, bf16x8& pv, bf16x8 (&wv)[TN]) {
        const int t = st / KS, sl = st % KS;
        pv = *reinterpret_cast<const bf16x8*>(hrow + toff[t] + 32 * sl);
#pragma unroll
        for (int j = 0; j < TN; ++j) wv[j] = *reinterpret_cast<const bf16x8*>(wlds + woff[sl] + t * WTAP + j * 32 * WR);
    };
    const bool staged = !a.out_f32;
    bf16* outb = reinterpret_cast<bf16*>(a.out);
    const bf16* addb = reinterpret_cast<const bf16*>(a.add);

    // residual + ReLU of a bf16 output are applied where the layouts agree: on the coalesced side, while the tile is
    // drained (the residual rows arrive as whole pixel rows too), instead of a round trip through the patch per tile
    // (plain instantiation only: the feature variants have no registers to carry the residual rows across tiles)
    const bool res_drain = !IN_SS && !STATS && staged && a.add != nullptr && !a.add_after_relu;
    u32x4 radd[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) radd[i] = u32x4{0u, 0u, 0u, 0u};
    auto finish_piece = [&](const u32x4& piece, const u32x4& res) -> u32x4 {
        const bf16x8 x = *reinterpret_cast<const bf16x8*>(&piece), rr = *reinterpret_cast<const bf16x8*>(&res);
        bf16x8 o;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float f = (float)x[c] + (float)rr[c];
            if (a.relu) f = fmaxf(f, 0.f);
            o[c] = (bf16)f;
        }
        return *reinterpret_cast<u32x4*>(&o);
    };
    // Tile order.  Plain launches stride the tiles over the grid.  The feature variants walk whole images (a
    // workgroup's image j is blockIdx.x + j * gridDim.x, tiles in order).  With statistics the sums then run over all
    // tiles of an image, the cross-lane reduction and the store happen once per image -- per tile they cost 10 % of the
    // kernel -- and an image's chunks do not depend on the batch it is in.  The normalising consumer follows the same
    // order so that an image stays on the XCD (blockIdx.x mod 8) whose L2 its producer wrote it through.
    constexpr bool IMO = IN_SS || STATS;
    const int tpi = a.g.tpi;
    auto tile_at = [&](int i) -> int {
        if (!IMO) return blockIdx.x + i * (int)gridDim.x;
        const int j = a.g.div_tpi(i);
        return (blockIdx.x + j * (int)gridDim.x) * tpi + (i - j * tpi);
    };
    auto in_image = [&](int i) -> int { return i - a.g.div_tpi(i) * tpi; };   // position of iteration i's tile in its image (IMO)
    const int nit = IMO ? ((a.B - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x) * tpi
                        : (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    int tile = tile_at(0);
    if (nit > 0) {
#pragma unroll
        for (int k = 0; k < HV; ++k) load_halo_vec(tile, k);
    }
    int it = 0;
#pragma nounroll                                           // also keeps the first iteration from being peeled off
    for (; it < nit; ++it, tile = tile_at(it)) {
        HALO_STAMP(0);
        store_halo();
        HALO_STAMP(1);
        __syncthreads();                                   // halo (and, first time, weights) visible
        HALO_STAMP(2);
        const int next = tile_at(it + 1), prev = tile_at(it > 0 ? it - 1 : 0);
        const bool drain = staged && it > 0, fetch_add = staged && a.add != nullptr, fetch_next = it + 1 < nit;
        // statistics: the drained tile (it - 1) opens / closes an image
        const int st_pos = STATS ? in_image(it > 0 ? it - 1 : 0) : 0;
        const bool st_open = st_pos == 0, st_close = st_pos == tpi - 1;
        u32x4 dreg;
        // q-th global access of this tile (compile-time q): drain reads/stores, add fetches, halo fetches
        auto mem_op = [&](int q) {
            if (q <= NI) {
                if (drain) {
                    if (q == 0 && STATS && a.stats && st_open) {
#pragma unroll
                        for (int c = 0; c < 8; ++c) { ssum[c] = 0.f; ssq[c] = 0.f; }
                    }
                    if (q > 0) {
                        if (res_drain) dreg = finish_piece(dreg, radd[q - 1]);   // radd still holds the previous tile's rows
                        if (chunk_live) *reinterpret_cast<u32x4*>(outb + out_off(prev, q - 1)) = dreg;
                        if (STATS && a.stats) stats_add(dreg);
                    }
                    if (q < NI) dreg = *reinterpret_cast<const u32x4*>(stg + (q * PPI + sp) * SRS + sc8 * 16);
                    if (q == NI && STATS && a.stats && st_close) stats_flush(prev);
                }
            } else if (q <= 2 * NI) {
                const int i = q - NI - 1;
                const u32x4 z = {0u, 0u, 0u, 0u};
                radd[i] = (fetch_add && chunk_live) ? *reinterpret_cast<const u32x4*>(addb + out_off(tile, i)) : z;
            } else {
                if (fetch_next) load_halo_vec(next, q - 2 * NI - 1);
            }
        };

        f32x16 acc[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
        {
            // fully unrolled over the 9 taps x KS slabs; fragments are read PF steps ahead of the MFMAs that use them
            constexpr int PF = 1;
            bf16x8 pv[PF + 1], wv[PF + 1][TN];
#pragma unroll
            for (int st = 0; st < PF; ++st) frag(st, pv[st], wv[st]);
#pragma unroll
            for (int st = 0; st < NSTEP; ++st) {
                if (st + PF < NSTEP) frag(st + PF, pv[(st + PF) % (PF + 1)], wv[(st + PF) % (PF + 1)]);
                if (st % OPSTRIDE == 0 && st / OPSTRIDE < NOPS) mem_op(st / OPSTRIDE);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    // D[i = out channel][j = pixel]: each lane ends up with 4 consecutive channels per register quad
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wv[st % (PF + 1)][j], pv[st % (PF + 1)], acc[j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);         // keep the memory operations where they were placed
            }
        }
        HALO_STAMP(4);

        if (a.out_f32) {
            // ---- fp32 output (narrow decoder heads): straight from registers, lane = pixel, quad g = channels 8g+4h.. ----
            int y0, x0, b;
            tile_origin(tile, y0, x0, b);
            const long pix = (long)(b * a.H + y0 + pl / TW) * a.W + x0 + pl % TW;
            float* op = reinterpret_cast<float*>(a.out);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int nl = j * 32 + 8 * g + 4 * h;
                    const int nn = n0 + nl;
                    if (nn >= a.NS) continue;
                    const f32x4 sc = *reinterpret_cast<const f32x4*>(ssl + nl);
                    const f32x4 sh = *reinterpret_cast<const f32x4*>(ssl + NT + nl);
                    float v[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = acc[j][4 * g + k] * sc[k] + sh[k];
                    const long o = pix * a.NS + nn;
                    if (a.add && !a.add_after_relu) {
                        const f32x4 rv = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(a.add) + o);
#pragma unroll
                        for (int k = 0; k < 4; ++k) v[k] += rv[k];
                    }
                    if (a.relu) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], 0.f);
                    }
                    if (a.add && a.add_after_relu) {
                        const f32x4 ov = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(a.add) + o);
#pragma unroll
                        for (int k = 0; k < 4; ++k) v[k] += ov[k];
                    }
                    f32x4 st4 = {v[0], v[1], v[2], v[3]};
                    *reinterpret_cast<f32x4*>(op + o) = st4;
                }
            }
        } else {
            // ---- bf16 output: finish in registers, park in the wave's LDS patch; the next tile's MFMA loop drains it ----
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int nl = j * 32 + 8 * g + 4 * h;
                    const f32x4 sc = *reinterpret_cast<const f32x4*>(ssl + nl);
                    const f32x4 sh = *reinterpret_cast<const f32x4*>(ssl + NT + nl);
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[j][4 * g + k] = acc[j][4 * g + k] * sc[k] + sh[k];
                }
            auto add_staged = [&]() {
#pragma unroll
                for (int i = 0; i < NI; ++i)
                    *reinterpret_cast<u32x4*>(stg + (i * PPI + sp) * SRS + sc8 * 16) = radd[i];
                wave_lds_fence();
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const bf16x4 q = *reinterpret_cast<const bf16x4*>(stg + r * SRS + (j * 32 + 8 * g + 4 * h) * 2);
#pragma unroll
                        for (int k = 0; k < 4; ++k) acc[j][4 * g + k] += (float)q[k];
                    }
                wave_lds_fence();
            };
            if (a.add && !a.add_after_relu && !res_drain) add_staged();
            if (a.relu && !res_drain) {
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[j][e] = fmaxf(acc[j][e], 0.f);
            }
            if (a.add && a.add_after_relu) add_staged();
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    bf16x4 q;
#pragma unroll
                    for (int k = 0; k < 4; ++k) q[k] = (bf16)acc[j][4 * g + k];
                    *reinterpret_cast<bf16x4*>(stg + r * SRS + (j * 32 + 8 * g + 4 * h) * 2) = q;
                }
            wave_lds_fence();
        }
        HALO_STAMP(5);
        __syncthreads();                                   // every wave is done reading this halo
        HALO_STAMP(6);
    }
    // ---- the last tile's output is still parked in the patch ----
    if (staged && it > 0) {
        const int last = tile_at(it - 1);
        if (STATS && in_image(it - 1) == 0) {              // one tile per image: the last tile opens its image too
#pragma unroll
            for (int c = 0; c < 8; ++c) { ssum[c] = 0.f; ssq[c] = 0.f; }
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            u32x4 piece = *reinterpret_cast<const u32x4*>(stg + (i * PPI + sp) * SRS + sc8 * 16);
            if (res_drain) piece = finish_piece(piece, radd[i]);
            if (chunk_live) *reinterpret_cast<u32x4*>(outb + out_off(last, i)) = piece;
            if (STATS && a.stats) stats_add(piece);
        }
        if (STATS && a.stats) stats_flush(last);
    }
}
#undef HALO_STAMP

static constexpr int halo_lds_bytes_of(int CIN, int NT, int TW) {
    // weights of 9 taps + one halo tile + epilogue scale/shift + the 8 waves' output patches
    return 9 * NT * (CIN * 2) + (256 / TW + 2) * (TW + 2) * (CIN * 2 + 16) + 2 * NT * 4 + 8 * 32 * (NT * 2 + 16);
}
template <int CIN, int NT, int TW>
static constexpr int halo_lds_bytes() { return halo_lds_bytes_of(CIN, NT, TW); }

static constexpr int HALO_LDS_MAX = 160 * 1024;

// Tile geometry of a plan the halo kernel can run, or TW = 0.
struct HaloGeom { int TW, NT; };
static HaloGeom halo_geom(const GemmPlan& P) {
    const HaloGeom none = {0, 0};
    if (P.nphase != 1) return none;
    const Phase& ph = P.ph[0];
    if (ph.SY != 1 || ph.SX != 1 || ph.DSY != 1 || ph.DSX != 1 || ph.D0Y || ph.D0X) return none;
    if (P.Hin != P.Hout || P.Win != P.Wout || ph.Hg != P.Hout || ph.Wg != P.Wout) return none;
    if (ph.ntaps != 9) return none;
    for (int t = 0; t < ph.ntaps; ++t)
        if (ph.dy[t] < -1 || ph.dy[t] > 1 || ph.dx[t] < -1 || ph.dx[t] > 1) return none;
    if (P.CgS != 32 && P.CgS != 64) return none;
    const int TW = (P.Wout % 32 == 0) ? 32 : ((P.Wout % 16 == 0) ? 16 : 0);
    if (!TW || P.Hout % (256 / TW)) return none;
    TileGrid tg;
    // (the largest dividend is the index of a tile one grid stride past the last image)
    if (!make_tile_grid(P.Wout / TW, P.Hout / (256 / TW), ((long)P.B + 1024) * (P.Wout / TW) * (P.Hout / (256 / TW)), tg)) return none;
    const int NT = (P.Npad % 64 == 0) ? 64 : 32;
    if (halo_lds_bytes_of(P.CgS, NT, TW) > HALO_LDS_MAX) return none;
    const HaloGeom g = {TW, NT};
    return g;
}

// true if the plan runs on the halo kernel
static bool halo_applicable(const GemmPlan& P) { return halo_geom(P).TW != 0; }
// ... and with the input-normalisation prologue / output statistics (built for 32-wide tiles only)
static bool halo_features_applicable(const GemmPlan& P) { return halo_geom(P).TW == 32; }

static bool try_launch_halo_bf16(const GemmPlan& P, const void* in, const void* wp, void* out, const float* scale,
                                 const float* shift, const void* res, int relu, int accumulate, int out_f32,
                                 hipStream_t st, const float* in_ss = nullptr, int in_relu = 0,
                                 unsigned long long* stamps = nullptr, float* stats = nullptr) {
    const HaloGeom hg = halo_geom(P);
    if (!hg.TW) return false;
    const Phase& ph = P.ph[0];
    const int TW = hg.TW, NT = hg.NT;
    HaloArgs a;
    if (res && accumulate) return false;                  // one additive operand per launch
    a.in = in; a.wp = wp; a.out = out; a.scale = scale; a.shift = shift;
    a.add = res ? res : (accumulate ? out : nullptr); a.add_after_relu = accumulate ? 1 : 0;
    if (stats && out_f32) return false;                   // statistics ride on the bf16 drain
    if ((stats || in_ss) && hg.TW != 32) return false;
    a.in_ss = in_ss; a.in_relu = in_relu; a.stamps = stamps; a.stats = stats;
    a.B = P.B; a.H = P.Hout; a.W = P.Wout; a.N = P.N; a.NS = out_f32 ? act_channels(P.N) : P.NS; a.Ktot = P.Ktot;
    a.relu = relu; a.out_f32 = out_f32;
    make_tile_grid(P.Wout / TW, P.Hout / (256 / TW), 0, a.g);
    memcpy(a.dy, ph.dy, 16); memcpy(a.dx, ph.dx, 16);
    const int ntiles = P.B * a.g.tpi;
    // persistent workgroups, tiles strided over the grid: one per CU where the LDS footprint allows no more, two for the
    // small shapes (32 -> 32 channels: 66 KB), whose tiles are mostly serial latency that a second workgroup hides
    const int per_cu = 2 * halo_lds_bytes_of(P.CgS, NT, TW) <= HALO_LDS_MAX ? 2 : 1;
    int gx = ntiles < 256 * per_cu ? ntiles : 256 * per_cu;
    if ((stats || in_ss) && gx > P.B) gx = P.B;           // the feature variants walk whole images
    dim3 grid(gx, P.Npad / NT), block(512);
    const bool f_in = in_ss != nullptr, f_st = stats != nullptr;
#define HALO_CASE(C, N, T, FI, FS)                                                                           \
    if (P.CgS == C && NT == N && TW == T && f_in == FI && f_st == FS) {                                      \
        constexpr int lds = halo_lds_bytes<C, N, T>();                                                       \
        if (lds > HALO_LDS_MAX) return false;                                                                \
        static LdsOptIn lds_opt;                                                                             \
        lds_opt.ensure(reinterpret_cast<const void*>(&conv_halo_bf16_kernel<C, N, T, FI, FS>), lds);         \
        hipLaunchKernelGGL((conv_halo_bf16_kernel<C, N, T, FI, FS>), grid, block, lds, st, a);               \
        return true;                                                                                         \
    }
#define HALO_SHAPES(T, FI, FS) HALO_CASE(64, 64, T, FI, FS) HALO_CASE(64, 32, T, FI, FS) HALO_CASE(32, 64, T, FI, FS) HALO_CASE(32, 32, T, FI, FS)
    HALO_SHAPES(32, false, false) HALO_SHAPES(32, false, true) HALO_SHAPES(32, true, false) HALO_SHAPES(32, true, true)
    HALO_SHAPES(16, false, false)
#undef HALO_SHAPES
#undef HALO_CASE
    return false;
}

}  // namespace dvae
