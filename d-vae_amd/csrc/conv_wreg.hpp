// 3x3 (radius-1) stride-1 bf16 convolution with the WEIGHTS resident in registers.
//
// conv_halo.hpp keeps the nine taps' weights in LDS and re-reads them for every MFMA: two of every three bytes a wave
// reads from LDS there are weights, the LDS array runs at 75 % of its bandwidth next to a full matrix pipe, and the serial
// parts of a tile (halo registers -> LDS, epilogue, two barriers) have nothing to hide behind.  Here a workgroup is four
// waves, one per SIMD, each with the whole 512-entry register file: a wave holds the weight matrix (9 taps x CIN x NT
// bf16; 64 -> 64: eight taps = the 256 registers of the accumulator file, the ninth in 32 architectural VGPRs) as
// ready-made MFMA A-operand fragments for the life of the workgroup.  What is left for LDS is the pixel operand (1 KB per
// two MFMAs), so one wave per SIMD keeps its matrix pipe fed, and everything else a tile needs is issued from inside the
// fully unrolled MFMA stream, in the shadow of the MFMAs:
//   - the next tile's halo: global -> registers, one 16-byte vector every few MFMA steps, and registers -> the other halo
//     buffer a fixed number of steps later (two halo buffers, ONE workgroup barrier per tile);
//   - the previous 32-pixel block's epilogue (scale / shift / residual / ReLU / bf16, all in fp32, rounded once) into the
//     wave's private LDS patch, which transposes the MFMA's 4-channel quads into whole 128-byte pixel rows, while the next
//     block's MFMAs run on a second accumulator set;
//   - that patch's drain to HBM as 16-byte pieces of whole pixel rows;
//   - the residual / old-output rows of the block after: fetched coalesced, parked in a second patch, read back in
//     accumulator layout by the epilogue.
// The MFMA is issued as D[channel][pixel] = W * X^T (lane = pixel, register quad = 4 consecutive channels), as in
// conv_halo.hpp; a wave owns two rows (64 pixels) of an 8 x 32 output tile and runs them as two blocks of one row.
// The MFMAs are inline assembly: hipcc allocates builtin MFMA operands in the 256 architectural VGPRs and treats the
// accumulator file as spill space (four v_accvgpr_read per MFMA); the "a" constraint pins the weights there.  hipcc pads no
// hazards for an asm statement: an accumulator set is first read by the epilogue at least four MFMAs after its last write
// (G_MIN), and the tail code waits explicitly.
// Serves the plans of conv_halo.hpp's plain instantiation that have 64 input channels and 32-pixel-wide tiles: Conv2d 3x3
// s1 forward and input-gradient, ConvTranspose2d 3x3 s1 forward and input-gradient, bf16 output.
// Measured (64 -> 64 at 32x32, 1024 images per launch): 72 us against 93 us for conv_halo.hpp when either kernel runs
// alone in a loop, 5 870 against 7 900 shader cycles per tile; inside the purification step, where every launch around it
// is as dense, the card settles at a lower clock and both take 79 us.
#pragma once
#include <math.h>

#include "conv_halo.hpp"

namespace dvae {

// Profiling builds (scripts/wreg_stamps.hip) define WREG_STAMPS: wave 0 of a workgroup then samples the cycle counter at
// fixed MFMA steps of its first 64 tiles into HaloArgs.stamps [grid.x][64][8].  Product builds carry none of it.
#ifdef WREG_STAMPS
#define WREG_STAMP(i)                                                                                    \
    do {                                                                                                 \
        if (a.stamps && tid == 0 && it < 64)                                                             \
            a.stamps[((long)blockIdx.x * 64 + it) * 8 + (i)] = __builtin_readcyclecounter();             \
    } while (0)
// ... and the constant 100 MHz clock (the same on every CU) at four points of the workgroup's life, into tile slot 63:
// entry, weights in registers, tile loop done, exit.
#define WREG_RSTAMP(i)                                                                                   \
    do {                                                                                                 \
        if (a.stamps && tid == 0) a.stamps[((long)blockIdx.x * 64 + 63) * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define WREG_STAMP(i) do { } while (0)
#define WREG_RSTAMP(i) do { } while (0)
#endif

// D = A * B (zero C) / D += A * B with the A fragment in the accumulator file or in VGPRs
__device__ __forceinline__ void mfma_first_a(f32x16& acc, const u32x4& w, const u32x4& x) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc) : "a"(w), "v"(x));
}
__device__ __forceinline__ void mfma_acc_a(f32x16& acc, const u32x4& w, const u32x4& x) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(w), "v"(x));
}
__device__ __forceinline__ void mfma_first_v(f32x16& acc, const u32x4& w, const u32x4& x) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc) : "v"(w), "v"(x));
}
__device__ __forceinline__ void mfma_acc_v(f32x16& acc, const u32x4& w, const u32x4& x) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(w), "v"(x));
}

#ifdef WREG_X16_TIMING
// Timing-only build: every 32x32x16 MFMA is issued as two 16x16x32 MFMAs on quarters of the same accumulator (results
// are wrong; same FLOPs, registers and dependency distances).
typedef f32x4 wacc_t[4];
__device__ __forceinline__ void mfma16_first_a(f32x4& acc, const u32x4& w, const u32x4& x) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(acc) : "a"(w), "v"(x));
}
__device__ __forceinline__ void mfma16_acc_a(f32x4& acc, const u32x4& w, const u32x4& x) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(w), "v"(x));
}
__device__ __forceinline__ void mfma16_first_v(f32x4& acc, const u32x4& w, const u32x4& x) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(acc) : "v"(w), "v"(x));
}
__device__ __forceinline__ void mfma16_acc_v(f32x4& acc, const u32x4& w, const u32x4& x) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(w), "v"(x));
}
#define WACC(acc, j, i) acc[j][(i) >> 2][(i) & 3]
#else
typedef f32x16 wacc_t;
#define WACC(acc, j, i) acc[j][i]
#endif

// ---- the side-operation schedule of one tile ----------------------------------------------------------------------
// A tile is 2 blocks x G gaps (a gap = the issue slot in front of one MFMA).  An MFMA executes for 32 cycles and holds the
// issue port for 8; a wave issues one other instruction every 4-5 cycles, so about FIVE instructions per gap are free and
// every further one delays the MFMA behind it for good (measured: the cost of a group of side operations is its
// instruction count beyond that budget, not its LDS or memory traffic).  The schedule therefore books instructions:
// every gap has a capacity (5, less the pixel-fragment read + its wait in the first gap of a step, less the address add
// at a tap's first slab), every micro-operation a cost, and a micro-operation goes to the first gap at or behind its
// evenly-spread target that still has room and respects the distance to the operation it depends on.
//   Block u runs a fixed list on the block BEFORE it (block 0: the previous tile's block 1; block 1: this tile's block 0):
//     [R x NR: fetch of its OWN residual rows]  EL0 EL1 | EA0 [ER0] EB0 EC0 EL2 | EA1 [ER1] EB1 EC1 EL3 | ... | ORD |
//     DL0 DL1 DS0 DL2 DS1 ... | [PARK x NR: its own residual rows -> residual patch]  ORD
//   (EL = epilogue operand loads, EA = scale and shift, ER = residual addition, EB / EC = rounding, ReLU and the patch
//   write, DL / DS = patch read / global store of a drain piece, ORD = compiler-level ordering point of the wave's LDS
//   accesses).
//   The next tile's halo vector k is fetched (HL) near gap k * (2G - HDIST) / HV and written to the other halo buffer (HS)
//   at least HDIST gaps later.  (Fetching two tiles ahead, a whole tile between fetch and store, measured 5 % slower:
//   global latency is not what the waits see.)
enum { W_NONE = -1, W_R = 0, W_EL, W_EA, W_ER, W_EB, W_EC, W_ORD, W_DL, W_DS, W_PARK, W_HL, W_HS };
constexpr int W_SLOTS = 4;
template <int G2>
struct WregSched {
    signed char kind[G2][W_SLOTS], idx[G2][W_SLOTS];
    int first_ea[2];                                       // gap (within its block) of each block's first EA
    int over;                                              // operations that found no gap with room (0 in a schedule that is used)
    int cap;                                               // the instructions-per-gap budget the schedule was built for
};
__host__ __device__ constexpr int wreg_op_cost(int kind, bool has_add, bool in_ss = false, bool stats = false) {
    switch (kind) {
    case W_R: return 2;
    case W_EL: return has_add ? 3 : 2;
    case W_EA: return 2;
    case W_ER: return 6;
    case W_EB: return 3;
    case W_EC: return 2;
    case W_DL: return 1;
    case W_DS: return stats ? 27 : 3;                      // + sum and sum of squares of the piece's 8 channels
    case W_PARK: return 2;
    case W_HL: return 4;
    case W_HS: return in_ss ? 34 : 2;                      // + scale / shift / ReLU of the vector's 8 channels on its way to LDS
    default: return 0;
    }
}
template <int G, int TN, int KS, int NEPI, int NI, int NR, int HV, int HDIST, int G_MIN, bool HAS_ADD, bool IN_SS = false, bool STATS = false>
__host__ __device__ constexpr WregSched<2 * G> wreg_try_sched(int CAP) {
    constexpr int G2 = 2 * G, LAT = 4;                     // LAT: gaps between an LDS read and the first use of its data
    WregSched<G2> s{};
    s.cap = CAP;
    int cap[G2] = {};
    for (int g = 0; g < G2; ++g) {
        for (int k = 0; k < W_SLOTS; ++k) { s.kind[g][k] = W_NONE; s.idx[g][k] = 0; }
        const int gb = g % G, st = gb / TN, j = gb % TN;
        cap[g] = CAP - (j == 0 ? 2 : 0) - (j == 0 && st % KS == 0 ? 1 : 0);
    }
    for (int g = G + 2; g < G + 10 && g < G2; g += 2) cap[g] = 0;         // the next tile's origin and masks are computed here
    s.over = 0;
    auto place = [&](int g, int gmax, int kind, int ix) -> int {        // first gap in [g, gmax) with room; books it
        const int cost = wreg_op_cost(kind, HAS_ADD, IN_SS, STATS);
        int at = -1;
        for (int q = g; q < gmax; ++q) {
            // (an operation bigger than any gap goes to one nothing else has been booked into)
            if (cap[q] < cost && cost > 0 && (s.kind[q][0] != W_NONE || cap[q] < 2)) continue;
            int sl = 0;
            while (sl < W_SLOTS && s.kind[q][sl] != W_NONE) ++sl;
            if (sl < W_SLOTS) { at = q; break; }
        }
        if (at < 0) { s.over += 1; return g < gmax ? g : gmax - 1; }     // this budget does not do: the caller raises it
        int sl = 0;
        while (s.kind[at][sl] != W_NONE) ++sl;
        s.kind[at][sl] = (signed char)kind;
        s.idx[at][sl] = (signed char)ix;
        cap[at] -= cost;
        return at;
    };
    // the block list
    constexpr int ML = NR + 5 * NEPI + 1 + 2 * NI + NR + 1;
    int lk[ML] = {}, li[ML] = {};
    int n = 0;
    for (int i = 0; i < NR; ++i) { lk[n] = W_R; li[n++] = i; }
    lk[n] = W_EL; li[n++] = 0;
    if (NEPI > 1) { lk[n] = W_EL; li[n++] = 1; }
    for (int e = 0; e < NEPI; ++e) {
        lk[n] = W_EA; li[n++] = e;
        if (HAS_ADD) { lk[n] = W_ER; li[n++] = e; }
        lk[n] = W_EB; li[n++] = e;
        lk[n] = W_EC; li[n++] = e;
        if (e + 2 < NEPI) { lk[n] = W_EL; li[n++] = e + 2; }
    }
    lk[n] = W_ORD; li[n++] = 0;
    lk[n] = W_DL; li[n++] = 0;
    if (NI > 1) { lk[n] = W_DL; li[n++] = 1; }
    for (int i = 0; i < NI; ++i) {
        lk[n] = W_DS; li[n++] = i;
        if (i + 2 < NI) { lk[n] = W_DL; li[n++] = i + 2; }
    }
    for (int i = 0; i < NR; ++i) { lk[n] = W_PARK; li[n++] = i; }
    lk[n] = W_ORD; li[n++] = 1;
    for (int u = 0; u < 2; ++u) {
        s.first_ea[u] = -1;
        int g_el[NEPI] = {}, g_ea[NEPI] = {}, g_dl[NI] = {}, g_r = 0, prev = 0;
        for (int m = 0; m < n; ++m) {
            const int kind = lk[m], ix = li[m];
            int g = m * G / n;                             // the evenly-spread target
            if (g < prev) g = prev;                        // list order is program order
            if (kind == W_EA) { if (g < G_MIN + 1) g = G_MIN + 1; if (g < g_el[ix] + LAT) g = g_el[ix] + LAT; }
            if ((kind == W_ER || kind == W_EB || kind == W_EC) && g < prev + 1) g = prev + 1;   // a dependent chain
            if (kind == W_DS && g < g_dl[ix] + LAT) g = g_dl[ix] + LAT;
            if (kind == W_PARK && g < g_r + HDIST) g = g_r + HDIST;
            if (g > G - 1) g = G - 1;
            const int at = place(u * G + g, (u + 1) * G, kind, ix) - u * G;
            prev = at;
            if (kind == W_EL) g_el[ix] = at;
            if (kind == W_EA) { g_ea[ix] = at; if (s.first_ea[u] < 0) s.first_ea[u] = at; }
            if (kind == W_DL) g_dl[ix] = at;
            if (kind == W_R) g_r = at;
        }
    }
    // the halo operations, into the room the blocks left
    int g_hl[HV] = {};
    for (int k = 0; k < HV; ++k) g_hl[k] = place(k * (G2 - HDIST - 1) / HV, G2 - HDIST, W_HL, k);
    for (int k = 0; k < HV; ++k) place(g_hl[k] + HDIST, G2, W_HS, k);
    return s;
}
// The schedule for the smallest instructions-per-gap budget (from the 5 that are free) that places every operation: the
// shapes with one MFMA per step cannot do with 5, and then the excess is spread over all gaps instead of piling up.
template <int G, int TN, int KS, int NEPI, int NI, int NR, int HV, int HDIST, int G_MIN, bool HAS_ADD, bool IN_SS = false, bool STATS = false>
__host__ __device__ constexpr WregSched<2 * G> wreg_make_sched() {
    for (int cap = 5; cap < 40; ++cap) {
        const auto s = wreg_try_sched<G, TN, KS, NEPI, NI, NR, HV, HDIST, G_MIN, HAS_ADD, IN_SS, STATS>(cap);
        if (s.over == 0) return s;
    }
    return wreg_try_sched<G, TN, KS, NEPI, NI, NR, HV, HDIST, G_MIN, HAS_ADD, IN_SS, STATS>(40);
}

// WTL: the last WTL taps' fragments are held in architectural VGPRs instead of the accumulator file (64 -> 64: nine taps
// are 288 registers, the accumulator file has 256)
// HAS_ADD: a residual / old-output tensor is added in the epilogue (compiled in, so that launches without one carry none
// of its operations and branches)
// IN_SS / STATS: the optional features of conv_halo.hpp's kernel, for launches that walk whole images (the launcher caps
// the grid at the image count): the per-(image, channel) scale / shift (+ ReLU) of the normalisation in front of the conv
// is applied to every halo vector on its way from registers to LDS (zero padding stays zero), and the sums / sums of
// squares of the stored output are gathered from the drain pieces, one chunk per tile row (8 per image), flushed when a
// workgroup has drained an image's last tile.
template <int CIN, int NT, int WTL, bool HAS_ADD, bool IN_SS = false, bool STATS = false>
__global__ __launch_bounds__(256) void conv_wreg_bf16_kernel(const HaloArgs a) {
    constexpr int TW = 32, TH = 8, HWD = TW + 2, HP = (TH + 2) * HWD;
    constexpr int RS = CIN * 2 + 16;          // bytes per staged halo pixel (16-byte pad spreads banks)
    constexpr int WR = CIN * 2;               // bytes per weight row in the LDS weight images (swizzled, unpadded)
    constexpr int VPP = CIN * 2 / 16;         // 16-byte vectors per pixel / weight row
    constexpr int WSH = (CIN == 64) ? 1 : 2;
    constexpr int HALO_BYTES = HP * RS;
    constexpr int WTAP = NT * WR;
    constexpr int TN = NT / 32, KS = CIN / 16, NSTEP = 9 * KS, NS2 = 2 * NSTEP;
    constexpr int HV = (HP * VPP + 255) / 256;            // 16-byte halo vectors per thread and tile
    constexpr int PSTEP = 256 / VPP;                      // halo pixels between a thread's consecutive vectors
    constexpr int SRS = NT * 2 + 16, VPS = NT / 8, PPI = 64 / VPS, NI = 32 / PPI;
    constexpr int TR = 9 - WTL;                           // taps held in registers
    constexpr int NEPI = 4 * TN;
    static_assert(9 * WTAP <= 2 * HALO_BYTES, "the start-up weight image is staged in the halo buffers");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* ssl = reinterpret_cast<float*>(smem + 2 * HALO_BYTES);          // scale[NT], shift[NT]

    const int tid = threadIdx.x;
    WREG_RSTAMP(0);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // provably uniform: it enters buffer-descriptor offsets
    const int lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int n0 = blockIdx.y * NT;
    const bf16* wp = reinterpret_cast<const bf16*>(a.wp);
    const int tpi = a.g.tpi;
    const int ntiles = a.B * tpi;
    // Tile order: with at least one image per workgroup a workgroup walks whole images (image j of workgroup g is
    // g + j * gridDim.x, tiles in order), so the two halo rows neighbouring tiles share are re-read from the cache level
    // that just served them; otherwise tiles are strided over the grid.
    const bool imo = IN_SS || STATS || a.B >= (int)gridDim.x;    // (the feature launches have at most one workgroup per image)
    const int tgrp = imo ? tpi : 1;                                   // (one branch-free form for both orders)
    const int imo_i = imo ? 1 : 0;
    // Reversed walk (a.rev, whole-image order only): the workgroup's images last to first, so that the launch as a whole
    // sweeps the tensors from their end -- where the launch before it, sweeping forwards, has just finished writing, and
    // the memory-side cache still holds what it wrote.
    const int rev_i = (imo && a.rev) ? 1 : 0;
    const int nimg_wg = (a.B - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    auto tile_at = [&](int i) -> int {
        const int j = i + (a.g.div_tpi(i) - i) * imo_i;               // imo ? i / tpi : i, without the branch a ?: becomes
        const int jr = j + (nimg_wg - 1 - 2 * j) * rev_i;
        return ((int)blockIdx.x + jr * (int)gridDim.x) * tgrp + (i - j * tgrp);
    };
    const int nit = imo ? ((a.B - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x) * tpi
                        : (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    auto tile_origin = [&](int tile, int& y0, int& x0, int& b) {
        int ty, tx;
        a.g.split(tile, b, ty, tx);
        x0 = tx * TW; y0 = ty * TH;
    };

    // ---- this thread's share of a halo tile: vectors v = tid + 256 k, i.e. halo pixel p0 + k * PSTEP, channel chunk cv.
    // Everything that does not depend on the tile is fixed here: the element offset of each vector from the halo's top-left
    // pixel, and bit masks (bit k = vector k) of the vectors that exist and of those in the halo's border rows / columns.
    const int p0 = tid / VPP, cv = tid % VPP;
    // A vector's address is descriptor base + hoff[k] (per lane) + horg (per tile, uniform).  The descriptor's base lies
    // one image row and one pixel in front of the tensor so that hoff[k] >= 0; its range check looks at the per-lane
    // part only, so a lane that must not read (halo pixel outside the image, vector beyond the halo) sets bit 31 of it
    // -- beyond any tensor this kernel accepts -- and gets zeros.
    const unsigned hbias = (unsigned)((a.W + 1) * CIN * 2);
    unsigned hoff[HV];
    unsigned m_left = 0, m_right = 0, m_top = 0, m_bot = 0, m_valid = 0;
#pragma unroll
    for (int k = 0; k < HV; ++k) {
        const int hp = p0 + k * PSTEP;
        const int hy = (hp * 241) >> 13, hx = hp - hy * HWD;          // hp / 34 for hp < 4000
        hoff[k] = (unsigned)((((hy - 1) * a.W + hx - 1) * CIN + cv * 8) * 2) + hbias;
        if (hp < HP) m_valid |= 1u << k; else hoff[k] = 0x80000000u;
        if (hx == 0) m_left |= 1u << k;
        if (hx == HWD - 1) m_right |= 1u << k;
        if (hy == 0) m_top |= 1u << k;
        if (hy == TH + 1) m_bot |= 1u << k;
    }
    // Stores and the residual fetch mask lanes the same way with an offset beyond the buffer: no EXEC-mask branches, no
    // registers zeroed by hand.
    constexpr unsigned OOB = 0xfffffff0u;
    const unsigned out_bytes = (unsigned)a.B * a.H * a.W * a.NS * 2u;
    const auto in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(a.in)) - hbias, 0, 0x80000000u, 0x00020000);
    u32x4 hreg[HV];
    unsigned hbad = 0;                                    // bit k: vector k of the tile being fetched lies outside the image
    unsigned horg = 0;                                    // byte offset of its origin pixel
    // (two halves, so that the loop can put them into different gaps)
    auto halo_origin = [&](int tile, int& y0, int& x0) -> int {       // pixel index of the tile's origin
        int b;
        tile_origin(tile, y0, x0, b);
        return (b * a.H + y0) * a.W + x0;
    };
    auto halo_mask_x = [&](int x0) -> unsigned {
        unsigned bad = 0;
        if (x0 == 0) bad |= m_left;
        if (x0 + TW == a.W) bad |= m_right;
        return bad;
    };
    auto halo_mask_y = [&](int y0) -> unsigned {
        unsigned bad = 0;
        if (y0 == 0) bad |= m_top;
        if (y0 + TH == a.H) bad |= m_bot;
        return bad;
    };
    int tpix = 0;                                          // origin pixel of the tile halo_tile() was last called for
    auto halo_tile = [&](int tile) {
        int y0, x0;
        tpix = halo_origin(tile, y0, x0);
        hbad = halo_mask_x(x0) | halo_mask_y(y0);
        horg = (unsigned)(tpix * CIN * 2);
    };
    auto load_halo_vec = [&](int k) {
        const unsigned off = (__builtin_amdgcn_ubfe(hbad, k, 1) << 31) | hoff[k];
        hreg[k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, off, horg, 0));
    };
    unsigned char* hst = smem + p0 * RS + cv * 16;                    // this thread's first vector in halo buffer 0
    // Only the last vector can lie beyond the halo; a thread that has none there writes into the padding of pixel 0
    // instead of branching around the store.
    static_assert((HV - 1) * 256 < HP * VPP, "every vector but the last exists for every thread");
    unsigned char* hst_last = (m_valid >> (HV - 1) & 1u) ? hst + (HV - 1) * PSTEP * RS : smem + CIN * 2;
    // input normalisation of this thread's 8 channels (chunk cv) for the image of the tile being fetched
    f32x4 isc[2], ish[2];
    const float in_lo = a.in_relu ? 0.f : -__builtin_inff();
    auto load_in_ss = [&](int b) {
        const float* p = a.in_ss + (long)b * 2 * CIN + cv * 8;
        isc[0] = *reinterpret_cast<const f32x4*>(p); isc[1] = *reinterpret_cast<const f32x4*>(p + 4);
        ish[0] = *reinterpret_cast<const f32x4*>(p + CIN); ish[1] = *reinterpret_cast<const f32x4*>(p + CIN + 4);
    };
    auto store_halo_vec = [&](int buf, int k) {
        unsigned char* at = k < HV - 1 ? hst + k * PSTEP * RS : hst_last;
        u32x4 val = hreg[k];
        if (IN_SS) {
            bf16x8 x = __builtin_bit_cast(bf16x8, val);
#pragma unroll
            for (int c = 0; c < 8; ++c) x[c] = (bf16)fmaxf((float)x[c] * isc[c >> 2][c & 3] + ish[c >> 2][c & 3], in_lo);
            const u32x4 nv = __builtin_bit_cast(u32x4, x);
            const bool pad = __builtin_amdgcn_ubfe(hbad, k, 1) != 0;       // outside the image: the zero padding stays zero
#pragma unroll
            for (int c = 0; c < 4; ++c) val[c] = pad ? 0u : nv[c];
        }
        *reinterpret_cast<u32x4*>(at + buf * HALO_BYTES) = val;
    };

    // ---- weights: global -> LDS images (coalesced) -> this wave's A-operand fragments, once ------------------------
    // Every load of the image is issued before the first LDS write (a load-store loop waits out one memory latency per
    // round: 18 rounds were 7 of the launch's 75 us), and ahead of the first tile's halo, which nothing needs before the
    // fragments are in registers (loads return in order).
    constexpr int WV = 9 * NT * VPP / 256;
    static_assert(9 * NT * VPP % 256 == 0, "the weight image is a whole number of vectors per thread");
    float ssl_sc = 0.f, ssl_sh = 0.f;                      // epilogue scale / shift of channel n0 + tid (tid < NT)
    if (tid < NT) {
        const int n = n0 + tid;
        const bool live = n < a.N;
        ssl_sc = (live && a.scale) ? a.scale[n] : (live ? 1.f : 0.f);
        ssl_sh = (live && a.shift) ? a.shift[n] : 0.f;
    }
    {
        u32x4 wtmp[WV];
#pragma unroll
        for (int i = 0; i < WV; ++i) {
            const int v = tid + 256 * i, t = v / (NT * VPP), rem = v % (NT * VPP), n = rem / VPP, c = rem % VPP;
            wtmp[i] = *reinterpret_cast<const u32x4*>(wp + (long)(n0 + n) * a.Ktot + t * CIN + c * 8);
        }
        int tile0 = tile_at(0);
        if (nit > 0) {
            halo_tile(tile0);
#pragma unroll
            for (int k = 0; k < HV; ++k) load_halo_vec(k);
            if (IN_SS) load_in_ss(a.g.div_tpi(tile0));
        }
#pragma unroll
        for (int i = 0; i < WV; ++i) {
            const int v = tid + 256 * i, t = v / (NT * VPP), rem = v % (NT * VPP), n = rem / VPP, c = rem % VPP;
            *reinterpret_cast<u32x4*>(smem + t * WTAP + n * WR + ((c ^ ((n >> WSH) & (VPP - 1))) * 16)) = wtmp[i];
        }
    }
    int tile = tile_at(0);
    if (tid < NT) { ssl[tid] = ssl_sc; ssl[NT + tid] = ssl_sh; }
    __syncthreads();
    int woff[KS];                                          // this lane's swizzled chunk of weight row r, per 16-channel slab
#pragma unroll
    for (int sl = 0; sl < KS; ++sl) woff[sl] = r * WR + (((2 * sl + h) ^ ((r >> WSH) & (VPP - 1))) * 16);
    u32x4 wreg[TR][KS][TN];                                // pinned to the accumulator file by the MFMA statements' "a"
    u32x4 wvr[WTL > 0 ? WTL : 1][KS][TN];                  // the taps that do not fit there: architectural VGPRs
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int sl = 0; sl < KS; ++sl)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const u32x4 w = *reinterpret_cast<const u32x4*>(smem + t * WTAP + j * 32 * WR + woff[sl]);
                if (t < TR) wreg[t][sl][j] = w; else wvr[t >= TR ? t - TR : 0][sl][j] = w;
            }
    __syncthreads();                                       // every wave has its fragments: the staging image is dead
    WREG_RSTAMP(1);
    if (nit > 0) {
#pragma unroll
        for (int k = 0; k < HV; ++k) store_halo_vec(0, k);
    }
    __syncthreads();

    int toff[9];                                           // LDS byte offset of each tap relative to the centre pixel
#pragma unroll
    for (int t = 0; t < 9; ++t) toff[t] = ((int)a.dy[t] * HWD + (int)a.dx[t]) * RS;
    // centre pixel of this lane in block 0 (tile row 2 * wave) of halo buffer 0; block 1 is one halo row further
    const int hrow0 = ((2 * wave + 1) * HWD + r + 1) * RS + 16 * h;
    unsigned char* stg = smem + 2 * HALO_BYTES + 2 * NT * 4 + wave * (32 * SRS);       // output patch
    unsigned char* rsg = stg + 4 * 32 * SRS;                                           // residual patch
    const int sp = lane / VPS, sc8 = lane % VPS;           // coalesced side: lane = (pixel sp of a PPI group, 8-channel chunk)
    const bool chunk_live = n0 + sc8 * 8 < a.NS;
    constexpr bool has_add = HAS_ADD;
    const auto out_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, out_bytes, 0x00020000);
    const auto add_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.add ? a.add : a.out), 0, out_bytes, 0x00020000);
    // this lane's byte offset inside a PPI-pixel piece of an output row, or beyond the buffer for a dead channel chunk
    const unsigned lane_off = chunk_live ? (unsigned)(sp * a.NS + n0 + sc8 * 8) * 2u : OOB;
    auto row_base = [&](int pix, int u) -> unsigned {      // byte offset of pixel 0 of block u (tile row 2 wave + u) of the
        return (unsigned)((pix + (2 * wave + u) * a.W) * a.NS) * 2u;    // tile whose origin is pixel pix
    };
    // LDS hand-over between lanes of ONE wave (patch written in accumulator layout, read in row layout): the LDS executes a
    // wave's instructions in order, so all that is needed is that the compiler keeps them in order -- a fence would also
    // wait for every outstanding LDS read (the pixel prefetches) to return
    auto wave_lds_order = [&]() { asm volatile("" ::: "memory"); };
    u32x4 radd[NI];
    // Epilogue group e = (j, g) of a finished accumulator set: 4 consecutive channels of this lane's pixel -> the output
    // patch, in three micro-operations: the LDS operands (scale, shift, residual quad), read well ahead of the arithmetic
    // (with one wave per SIMD a wait right behind a read exposes the whole LDS latency); channels 0-1; channels 2-3 + store.
    constexpr int EBUF = 3;
    f32x4 esc[EBUF], esh[EBUF];
    bf16x4 erq[EBUF];
    auto epi_load = [&](int e) {
        const int nl = (e / 4) * 32 + 8 * (e % 4) + 4 * h;
#ifdef WREG_ABL_SS
        esc[e % EBUF] = f32x4{1.f, 1.f, 1.f, 1.f}; esh[e % EBUF] = f32x4{0.f, 0.f, 0.f, 0.f};
#else
        esc[e % EBUF] = *reinterpret_cast<const f32x4*>(ssl + nl);
        esh[e % EBUF] = *reinterpret_cast<const f32x4*>(ssl + NT + nl);
#endif
        if (has_add) erq[e % EBUF] = *reinterpret_cast<const bf16x4*>(rsg + r * SRS + nl * 2);
    };
    // ReLU on the rounded pair: a bf16 read as a signed 16-bit integer is negative exactly when the number is, so one
    // packed integer max against 0 clamps both halves (against -32768 it changes nothing: launches without ReLU)
    typedef bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    typedef short i16x2_t __attribute__((ext_vector_type(2)));
    const i16x2_t relu_lo = {(short)(a.relu ? 0 : -32768), (short)(a.relu ? 0 : -32768)};
    auto pack_relu = [&](float f0, float f1) -> unsigned {
        typedef float f32x2_t __attribute__((ext_vector_type(2)));
        const f32x2_t f = {f0, f1};
        const bf16x2_t o = __builtin_convertvector(f, bf16x2_t);      // (bit casts take named objects, never a vector element)
        return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(i16x2_t, o), relu_lo));
    };
    auto epi_a = [&](const wacc_t (&acc)[TN], int e) {
        const int j = e / 4, g = e % 4, b = e % EBUF;
#pragma unroll
        for (int k = 0; k < 4; ++k) esh[b][k] = WACC(acc, j, 4 * g + k) * esc[b][k] + esh[b][k];
        asm volatile("" : "+v"(esh[b]));                   // computed HERE: the optimiser would sink it to its use
    };
    auto epi_r = [&](int e) {
        const int b = e % EBUF;
#pragma unroll
        for (int k = 0; k < 4; ++k) esh[b][k] += (float)erq[b][k];
        asm volatile("" : "+v"(esh[b]));
    };
    unsigned eo01[EBUF];                                   // channels 0-1 of the group: rounded, ReLU applied
    unsigned eo23[EBUF];                                   // channels 2-3: rounded
    auto epi_b = [&](int e) {
        const int b = e % EBUF;
        const float f0 = esh[b][0], f1 = esh[b][1], f2 = esh[b][2], f3 = esh[b][3];
        eo01[b] = pack_relu(f0, f1);
        typedef float f32x2_t __attribute__((ext_vector_type(2)));
        const f32x2_t f23 = {f2, f3};
        const bf16x2_t q = __builtin_convertvector(f23, bf16x2_t);
        eo23[b] = __builtin_bit_cast(unsigned, q);
        asm volatile("" : "+v"(eo01[b]), "+v"(eo23[b]));
    };
    auto epi_c = [&](int e) {
        const int nl = (e / 4) * 32 + 8 * (e % 4) + 4 * h, b = e % EBUF;
        const unsigned q = eo23[b];
        const unsigned o23 = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(i16x2_t, q), relu_lo));
        const u32x2 o = {eo01[b], o23};
        *reinterpret_cast<u32x2*>(stg + r * SRS + nl * 2) = o;
    };
    u32x4 dpc[2];
    auto drain_load = [&](int i) { dpc[i % 2] = *reinterpret_cast<const u32x4*>(stg + (i * PPI + sp) * SRS + sc8 * 16); };
    // base: byte offset of the block's pixel 0 (uniform).  The bounds check covers the per-lane offset only, so "do not
    // store" must arrive there (voff = OOB), never in the uniform part.
    auto drain_store = [&](unsigned base, unsigned voff, int i) {
        __builtin_amdgcn_raw_buffer_store_b128(dpc[i % 2], out_rsrc, voff, base + (unsigned)(i * PPI * a.NS * 2), 0);
    };
    auto fetch_add = [&](unsigned base, int i) {
        radd[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(add_rsrc, lane_off, base + (unsigned)(i * PPI * a.NS * 2), 0));
    };
    auto park_add = [&](int i) { *reinterpret_cast<u32x4*>(rsg + (i * PPI + sp) * SRS + sc8 * 16) = radd[i]; };

    // ---- output statistics: per lane (coalesced layout: pixel sp of a piece, 8-channel chunk sc8) the sums over the drained
    // pieces of block 0 (tile row 2 wave: chunk 2 wave of the image) and of block 1 (chunk 2 wave + 1) ----
    float ssum[2][8], ssq[2][8];
#pragma unroll
    for (int c = 0; c < 8; ++c) { ssum[0][c] = ssum[1][c] = 0.f; ssq[0][c] = ssq[1][c] = 0.f; }
    auto stats_add = [&](int blk, const u32x4& piece) {
        const bf16x8 x = __builtin_bit_cast(bf16x8, piece);
#pragma unroll
        for (int c = 0; c < 8; ++c) { const float f = (float)x[c]; ssum[blk][c] += f; ssq[blk][c] += f * f; }
    };
    auto stats_flush = [&](int blk, int img) {            // reduce over the lanes that hold the same 8 channels, store, reset
        float v[16];
#pragma unroll
        for (int c = 0; c < 8; ++c) { v[c] = ssum[blk][c]; v[8 + c] = ssq[blk][c]; ssum[blk][c] = 0.f; ssq[blk][c] = 0.f; }
        if (VPS == 4) {
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] += dpp_f<0x1B>(dpp_f<0x141>(v[i]));      // lane ^ 4
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] += dpp_f<0x128>(v[i]);                        // lane ^ 8 (row rotate by 8)
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] += __shfl_xor(v[i], 16, 64);
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] += __shfl_xor(v[i], 32, 64);
        if (sp == 0 && chunk_live) {
            float* dst = a.stats + (((long)img * 8 + 2 * wave + blk) * 2) * a.NS + n0 + sc8 * 8;
            f32x4 v0 = {v[0], v[1], v[2], v[3]}, v1 = {v[4], v[5], v[6], v[7]};
            f32x4 q0 = {v[8], v[9], v[10], v[11]}, q1 = {v[12], v[13], v[14], v[15]};
            *reinterpret_cast<f32x4*>(dst) = v0; *reinterpret_cast<f32x4*>(dst + 4) = v1;
            *reinterpret_cast<f32x4*>(dst + a.NS) = q0; *reinterpret_cast<f32x4*>(dst + a.NS + 4) = q1;
        }
    };
    auto in_image = [&](int i) -> int { return i - a.g.div_tpi(i) * tpi; };   // position of iteration i's tile in its image

    // ---- the schedule (wreg_make_sched) ----
    constexpr int G = NSTEP * TN;
    constexpr int HDIST = (2 * G * 5) / 16;                // gaps between a halo vector's fetch and its LDS store
    constexpr int NR = HAS_ADD ? NI : 0;
    // An accumulator set is first read (first EA) at least four MFMAs after its last MFMA: hipcc pads no hazards for the
    // asm MFMAs.
    constexpr int G_MIN = 3;
    constexpr auto SCH = wreg_make_sched<G, TN, KS, NEPI, NI, NR, HV, HDIST, G_MIN, HAS_ADD, IN_SS, STATS>();
    static_assert(SCH.first_ea[0] >= 4 && SCH.first_ea[1] >= 4, "an accumulator set is read too soon after its last MFMA");
    static_assert(SCH.over == 0, "a side operation has no place in the schedule");

    wacc_t acc0[TN], acc1[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) { WACC(acc0, j, e) = 0.f; WACC(acc1, j, e) = 0.f; }

    // loop-carried tile state: the tile being computed, the one whose halo is being fetched, and the offsets they need
    int it = 0;
    unsigned out_cur0 = row_base(tpix, 0), out_cur1 = row_base(tpix, 1);      // (halo_tile was last called for the first tile)
    int next = nit > 1 ? tile_at(1) : tile;
    halo_tile(next);
    if (IN_SS && nit > 0) load_in_ss(a.g.div_tpi(next));   // (the first tile's halo is in LDS, normalised, by now)
    int pix_next = tpix;
    unsigned out_prev1 = 0, voff_prev = OOB;               // block 1 of the previous tile (nothing to drain at first)
#pragma nounroll
    for (; it < nit; ++it) {
        const int cur = it & 1;
        const unsigned char* hrow = smem + cur * HALO_BYTES + hrow0;
        WREG_STAMP(0);
        // statistics: block 0 of THIS tile is drained during its block 1, block 1 of the PREVIOUS tile during this block 0
        const int st_pos = STATS ? in_image(it) : 0;
        const bool st_close0 = STATS && st_pos == tpi - 1;            // this tile closes its image
        const bool st_close1 = STATS && it > 0 && st_pos == 0;        // the previous tile closed its image
        const int st_img = STATS ? a.g.div_tpi(tile) : 0;             // (imo: tile_at(it) = this workgroup's tile `it`)
        const int st_img_prev = STATS && it > 0 ? a.g.div_tpi(tile_at(it - 1)) : 0;
        unsigned out_next0 = 0, out_next1 = 0;
        // the state of the next iteration, computed in the middle of this one (scalar instructions between MFMAs)
        int nx_next = 0, nx_pix = 0, nx_y0 = 0, nx_x0 = 0;
        unsigned nx_hbad = 0;
        {
            constexpr int PF = 4;                          // one wave per SIMD: LDS latency is covered by distance alone
            u32x4 pv[PF + 1];
            auto frag = [&](int s) -> u32x4 {              // pixel fragment of flattened step s = (block, tap, slab)
                const int u = s / NSTEP, st = s % NSTEP, t = st / KS, sl = st % KS;
                return *reinterpret_cast<const u32x4*>(hrow + u * (HWD * RS) + toff[t] + 32 * sl);
            };
#pragma unroll
            for (int s = 0; s < PF; ++s) pv[s] = frag(s);
#pragma unroll
            for (int s = 0; s < NS2; ++s) {
                const int u = s / NSTEP, st = s % NSTEP, t = st / KS, sl = st % KS;
                if (s == 1) WREG_STAMP(1);
                if (s == NSTEP / 2) WREG_STAMP(2);
                if (s == NSTEP) WREG_STAMP(3);
                if (s == NSTEP + NSTEP / 2) WREG_STAMP(4);
#ifndef WREG_ABL_PV
                if (s + PF < NS2) pv[(s + PF) % (PF + 1)] = frag(s + PF);
#endif
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int gt = u * G + st * TN + j;                // gap index in the tile
#ifndef WREG_ABL_SIDE
                    // next iteration's halo fetch parameters (the fetches of this tile still use the current ones)
                    // (an opaque copy of an input pins each piece to its gap: plain arithmetic on loop-carried values
                    // would be hoisted to the top of the tile, into one gap)
                    if (gt == G + 2) {
                        int i2 = it + 2;
                        asm volatile("" : "+s"(i2));
                        nx_next = tile_at(i2 < nit ? i2 : nit - 1);    // (past the end: a tile again whose halo nobody reads)
                    }
                    if (gt == G + 4) { asm volatile("" : "+s"(nx_next)); nx_pix = halo_origin(nx_next, nx_y0, nx_x0); }
                    if (gt == G + 6) { asm volatile("" : "+s"(nx_x0)); nx_hbad = halo_mask_x(nx_x0); }
                    if (gt == G + 8) { asm volatile("" : "+s"(nx_y0)); nx_hbad |= halo_mask_y(nx_y0); }
                    if (gt == 9) {
                        int pn = pix_next;
                        asm volatile("" : "+s"(pn));
                        out_next0 = row_base(pn, 0); out_next1 = row_base(pn, 1);
                    }
#pragma unroll
                    for (int sl2 = 0; sl2 < W_SLOTS; ++sl2) {
                        const int kind = SCH.kind[gt][sl2], ix = SCH.idx[gt][sl2];
#ifndef WREG_ABL_HALO
                        if (kind == W_HL) load_halo_vec(ix);
                        if (kind == W_HS) store_halo_vec(cur ^ 1, ix);
#endif
                        if (kind == W_R) fetch_add(u == 0 ? out_cur0 : out_cur1, ix);
#ifndef WREG_ABL_EPI
                        if (kind == W_EL) epi_load(ix);
                        if (kind == W_EA) { if (u == 0) epi_a(acc1, ix); else epi_a(acc0, ix); }
                        if (kind == W_ER) epi_r(ix);
                        if (kind == W_EB) epi_b(ix);
                        if (kind == W_EC) epi_c(ix);
#endif
                        if (kind == W_ORD) wave_lds_order();
#ifndef WREG_ABL_DRAIN
                        if (kind == W_DL) drain_load(ix);
                        if (kind == W_DS) {
                            drain_store(u == 0 ? out_prev1 : out_cur0, u == 0 ? voff_prev : lane_off, ix);
                            if (STATS) {
                                if (u == 1 || it > 0) stats_add(u == 0 ? 1 : 0, dpc[ix % 2]);
                                if (ix == NI - 1) {
                                    if (u == 1 && st_close0) stats_flush(0, st_img);
                                    if (u == 0 && st_close1) stats_flush(1, st_img_prev);
                                }
                            }
                        }
#endif
                        if (kind == W_PARK) park_add(ix);
                    }
#endif
                    // ---- the MFMA of this gap ----
#ifdef WREG_X16_TIMING
                    {
                        wacc_t& acc = u == 0 ? acc0[j] : acc1[j];
                        const int q0 = (2 * st) & 3, q1 = (2 * st + 1) & 3;
                        if (t < TR) {
                            if (st < 2) { mfma16_first_a(acc[q0], wreg[t < TR ? t : 0][sl][j], pv[s % (PF + 1)]);
                                          mfma16_first_a(acc[q1], wreg[t < TR ? t : 0][sl][j], pv[s % (PF + 1)]); }
                            else { mfma16_acc_a(acc[q0], wreg[t < TR ? t : 0][sl][j], pv[s % (PF + 1)]);
                                   mfma16_acc_a(acc[q1], wreg[t < TR ? t : 0][sl][j], pv[s % (PF + 1)]); }
                        } else {
                            mfma16_acc_v(acc[q0], wvr[t >= TR ? t - TR : 0][sl][j], pv[s % (PF + 1)]);
                            mfma16_acc_v(acc[q1], wvr[t >= TR ? t - TR : 0][sl][j], pv[s % (PF + 1)]);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
#else
                    f32x16& acc = u == 0 ? acc0[j] : acc1[j];
                    if (t < TR) {
                        if (st == 0) mfma_first_a(acc, wreg[t < TR ? t : 0][sl][j], pv[s % (PF + 1)]);
                        else mfma_acc_a(acc, wreg[t < TR ? t : 0][sl][j], pv[s % (PF + 1)]);
                    } else {
                        if (st == 0) mfma_first_v(acc, wvr[t >= TR ? t - TR : 0][sl][j], pv[s % (PF + 1)]);
                        else mfma_acc_v(acc, wvr[t >= TR ? t - TR : 0][sl][j], pv[s % (PF + 1)]);
                    }
                    __builtin_amdgcn_sched_barrier(0);     // keep everything where it was placed
#endif
                }
            }
        }
        WREG_STAMP(5);
        __syncthreads();                                   // next halo complete; every wave is done reading this one
        WREG_STAMP(6);
        out_prev1 = out_cur1; voff_prev = lane_off;
        out_cur0 = out_next0; out_cur1 = out_next1;
        tile = next; next = nx_next; hbad = nx_hbad; horg = (unsigned)(nx_pix * CIN * 2);
        pix_next = nx_pix;
        if (IN_SS) load_in_ss(a.g.div_tpi(next));          // for the halo stores of the next iteration (its tile after next)
    }
    WREG_RSTAMP(2);
    // ---- block 1 of the last tile is still in its accumulators (its residual rows are parked already) ----
    if (it > 0) {
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // the last MFMA's result -> the VALU reads below
#pragma unroll
        for (int e = 0; e < NEPI; ++e) { epi_load(e); epi_a(acc1, e); if (has_add) epi_r(e); epi_b(e); epi_c(e); }
        wave_lds_order();
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            drain_load(i);
            drain_store(out_prev1, lane_off, i);
            if (STATS) stats_add(1, dpc[i % 2]);
        }
        if (STATS) stats_flush(1, a.g.div_tpi(tile_at(it - 1)));      // a workgroup's last tile closes an image
    }
    WREG_RSTAMP(3);
}

static constexpr int wreg_lds_taps(int CIN, int NT) { return (CIN == 64 && NT == 64) ? 1 : 0; }
static constexpr int wreg_lds_bytes_of(int CIN, int NT) {
    // two halo buffers + epilogue scale/shift + per wave an output and a residual patch of one block
    return 2 * (10 * 34) * (CIN * 2 + 16) + 2 * NT * 4 + 8 * 32 * (NT * 2 + 16);
}

// true if the plan (a halo-kernel plan with 32-wide tiles) runs on the weights-in-registers kernel
static bool wreg_applicable(const GemmPlan& P) {
    const HaloGeom hg = halo_geom(P);
    return hg.TW == 32 && P.Hout % 8 == 0 && P.CgS == 64;
}

static bool try_launch_wreg_bf16(const GemmPlan& P, const void* in, const void* wp, void* out, const float* scale,
                                 const float* shift, const void* res, int relu, int accumulate, int out_f32, hipStream_t st,
                                 unsigned long long* stamps = nullptr, const float* in_ss = nullptr, int in_relu = 0,
                                 float* stats = nullptr, int reverse_walk = 0) {
    // (ReLU before the addition of the old output is the one order this kernel does not have; nothing asks for it)
    if (out_f32 || !wreg_applicable(P) || (res && accumulate) || (relu && accumulate)) return false;
    const bool feat = in_ss != nullptr || stats != nullptr;
    if (feat && (res || accumulate)) return false;         // the feature instantiations carry no additive operand
    // Bit 31 of a lane's input offset is the "return zeros" marker and the tile origins are 32-bit byte offsets: tensors
    // stay below 2 GB less the descriptor's lead-in (16 384 images of 32 x 32 x 64 and more go to the LDS-weights kernel).
    if ((long)P.B * P.Hout * P.Wout * (P.CgS > P.NS ? P.CgS : P.NS) * 2 >= 0x80000000L - (long)(P.Wout + 1) * P.CgS * 2 - 16)
        return false;
    const HaloGeom hg = halo_geom(P);
    const Phase& ph = P.ph[0];
    const int NT = hg.NT;
    HaloArgs a;
    memset(&a, 0, sizeof(a));
    a.in = in; a.wp = wp; a.out = out; a.scale = scale; a.shift = shift;
    a.add = res ? res : (accumulate ? out : nullptr); a.add_after_relu = 0;
    a.in_ss = in_ss; a.in_relu = in_relu; a.stats = stats;
    a.B = P.B; a.H = P.Hout; a.W = P.Wout; a.N = P.N; a.NS = P.NS; a.Ktot = P.Ktot;
    a.relu = relu; a.out_f32 = 0; a.stamps = stamps;
    a.rev = reverse_walk;
    make_tile_grid(P.Wout / 32, P.Hout / 8, 0, a.g);
    memcpy(a.dy, ph.dy, 16); memcpy(a.dx, ph.dx, 16);
    const int ntiles = P.B * a.g.tpi;
    int gx = ntiles < 256 ? ntiles : 256;                  // persistent: one workgroup per CU
    if (feat && gx > P.B) gx = P.B;                        // the feature variants walk whole images
    dim3 grid(gx, P.Npad / NT), block(256);
    const bool f_in = in_ss != nullptr, f_st = stats != nullptr;
#define WREG_CASE(C, N, A, FI, FS)                                                                       \
    if (P.CgS == C && NT == N && (a.add != nullptr) == A && f_in == FI && f_st == FS) {                  \
        constexpr int lds = wreg_lds_bytes_of(C, N);                                                     \
        static LdsOptIn lds_opt;                                                                         \
        lds_opt.ensure(reinterpret_cast<const void*>(&conv_wreg_bf16_kernel<C, N, wreg_lds_taps(C, N), A, FI, FS>), lds); \
        hipLaunchKernelGGL((conv_wreg_bf16_kernel<C, N, wreg_lds_taps(C, N), A, FI, FS>), grid, block, lds, st, a); \
        return true;                                                                                     \
    }
    // (64 input channels only: with 32 a block is 18 or 36 MFMAs, too short for its epilogue chain, and the halo kernel
    // measured faster)
    WREG_CASE(64, 64, false, false, false) WREG_CASE(64, 32, false, false, false)
    WREG_CASE(64, 64, true, false, false) WREG_CASE(64, 32, true, false, false)
    // the decoders' launches: 64 -> 64 leaving statistics, 64 -> 32 normalising its input and leaving statistics
    WREG_CASE(64, 64, false, false, true) WREG_CASE(64, 32, false, true, true) WREG_CASE(64, 32, false, true, false)
#undef WREG_CASE
    return false;
}

}  // namespace dvae
