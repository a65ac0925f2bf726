// 3x3 (radius-1) stride-1 bf16 convolution with the WEIGHTS resident in registers.
//
// conv_halo.hpp keeps the nine taps' weights in LDS and re-reads them for every MFMA: two of every three bytes a wave
// reads from LDS there are weights, the LDS array runs at 75 % of its bandwidth next to a full matrix pipe, and the serial
// parts of a tile (halo registers -> LDS, epilogue, two barriers) have nothing to hide behind.  Here a workgroup is four
// waves, one per SIMD, each with the whole 512-entry register file: a wave holds the weight matrix (9 taps x CIN x NT
// bf16; 64 -> 64: eight taps = 256 registers, the ninth stays in LDS) in the accumulator half of the register file as
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
// hazards for an asm statement: an accumulator set is first read by the epilogue at least six MFMAs after its last write
// (E_FIRST), and the tail code waits explicitly.
// Serves the same plans as conv_halo.hpp's plain instantiation with 32-pixel-wide tiles: Conv2d 3x3 s1 forward and
// input-gradient, ConvTranspose2d 3x3 s1 forward and input-gradient, bf16 output.
#pragma once
#include <math.h>

#include "conv_halo.hpp"

namespace dvae {

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

// max(f, lo) as ONE instruction (fmaxf costs a second v_max that quiets signalling NaNs first)
__device__ __forceinline__ float vmax1(float f, float lo) {
    float o;
    asm("v_max_f32 %0, %1, %2" : "=v"(o) : "v"(f), "v"(lo));
    return o;
}

// WTL: the last WTL taps' weights stay in LDS (read one step ahead of their MFMAs) instead of in registers
// HAS_ADD: a residual / old-output tensor is added in the epilogue (compiled in, so that launches without one carry none
// of its operations and branches)
template <int CIN, int NT, int WTL, bool HAS_ADD>
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
    static_assert(TR * WTAP <= 2 * HALO_BYTES, "the start-up weight image is staged in the halo buffers");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* ssl = reinterpret_cast<float*>(smem + 2 * HALO_BYTES);          // scale[NT], shift[NT]
    unsigned char* wtl = smem + 2 * HALO_BYTES + 2 * NT * 4 + 8 * 32 * SRS;   // [WTL][NT][WR]: the taps that stay in LDS

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // provably uniform: it enters buffer-descriptor offsets
    const int lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int n0 = blockIdx.y * NT;
    const bf16* wp = reinterpret_cast<const bf16*>(a.wp);
    const int mx = (1 << a.lx) - 1, my = (1 << a.ly) - 1, ltpi = a.lx + a.ly, tpi_m = (1 << ltpi) - 1;
    const int ntiles = a.B << ltpi;
    // Tile order: with at least one image per workgroup a workgroup walks whole images (image j of workgroup g is
    // g + j * gridDim.x, tiles in order), so the two halo rows neighbouring tiles share are re-read from the cache level
    // that just served them; otherwise tiles are strided over the grid.
    const bool imo = a.B >= (int)gridDim.x;
    auto tile_at = [&](int i) -> int {
        if (!imo) return blockIdx.x + i * (int)gridDim.x;
        return ((blockIdx.x + (i >> ltpi) * (int)gridDim.x) << ltpi) + (i & tpi_m);
    };
    const int nit = imo ? ((a.B - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x) << ltpi
                        : (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    auto tile_origin = [&](int tile, int& y0, int& x0, int& b) {
        x0 = (tile & mx) * TW; y0 = ((tile >> a.lx) & my) * TH; b = tile >> ltpi;
    };

    // ---- this thread's share of a halo tile: vectors v = tid + 256 k, i.e. halo pixel p0 + k * PSTEP, channel chunk cv.
    // Everything that does not depend on the tile is fixed here: the element offset of each vector from the halo's top-left
    // pixel, and bit masks (bit k = vector k) of the vectors that exist and of those in the halo's border rows / columns.
    const int p0 = tid / VPP, cv = tid % VPP;
    int hrel[HV];                                         // byte offset of vector k from the tile's origin pixel (may be < 0)
    unsigned m_valid = 0, m_left = 0, m_right = 0, m_top = 0, m_bot = 0;
#pragma unroll
    for (int k = 0; k < HV; ++k) {
        const int hp = p0 + k * PSTEP;
        const int hy = (hp * 241) >> 13, hx = hp - hy * HWD;          // hp / 34 for hp < 4000
        hrel[k] = (((hy - 1) * a.W + hx - 1) * CIN + cv * 8) * 2;
        if (hp < HP) m_valid |= 1u << k;
        if (hx == 0) m_left |= 1u << k;
        if (hx == HWD - 1) m_right |= 1u << k;
        if (hy == 0) m_top |= 1u << k;
        if (hy == TH + 1) m_bot |= 1u << k;
    }
    // Global accesses go through buffer descriptors: a lane that must not access memory (halo pixel outside the image,
    // dead channel chunk, nothing to drain yet) gets an offset beyond the buffer, which loads zeros and drops stores --
    // no EXEC-mask branches, no registers zeroed by hand.
    constexpr unsigned OOB = 0xfffffff0u;
    const unsigned in_bytes = (unsigned)a.B * a.H * a.W * CIN * 2u, out_bytes = (unsigned)a.B * a.H * a.W * a.NS * 2u;
    const auto in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.in), 0, in_bytes, 0x00020000);
    u32x4 hreg[HV];
    unsigned hok = 0;                                     // bit k: vector k of the tile being fetched lies inside the image
    unsigned horg = 0;                                    // byte offset of its origin pixel
    auto halo_tile = [&](int tile) {
        int y0, x0, b;
        tile_origin(tile, y0, x0, b);
        unsigned bad = 0;
        if (x0 == 0) bad |= m_left;
        if (x0 + TW == a.W) bad |= m_right;
        if (y0 == 0) bad |= m_top;
        if (y0 + TH == a.H) bad |= m_bot;
        hok = m_valid & ~bad;
        horg = (unsigned)(((b * a.H + y0) * a.W + x0) * CIN * 2);
    };
    auto load_halo_vec = [&](int k) {
        const unsigned off = (hok >> k & 1u) ? horg + (unsigned)hrel[k] : OOB;
        hreg[k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, off, 0, 0));
    };
    unsigned char* hst = smem + p0 * RS + cv * 16;                    // this thread's first vector in halo buffer 0
    auto store_halo_vec = [&](int buf, int k) {
        if (m_valid >> k & 1u) *reinterpret_cast<u32x4*>(hst + buf * HALO_BYTES + k * PSTEP * RS) = hreg[k];
    };

    int tile = tile_at(0);
    if (nit > 0) {
        halo_tile(tile);
#pragma unroll
        for (int k = 0; k < HV; ++k) load_halo_vec(k);
    }

    // ---- weights: global -> LDS images (coalesced) -> this wave's A-operand fragments, once ------------------------
    for (int v = tid; v < 9 * NT * VPP; v += 256) {
        const int t = v / (NT * VPP), rem = v % (NT * VPP), n = rem / VPP, c = rem % VPP;
        unsigned char* dst = t < TR ? smem + t * WTAP : wtl + (t - TR) * WTAP;
        *reinterpret_cast<u32x4*>(dst + n * WR + ((c ^ ((n >> WSH) & (VPP - 1))) * 16)) =
            *reinterpret_cast<const u32x4*>(wp + (long)(n0 + n) * a.Ktot + t * CIN + c * 8);
    }
    if (tid < NT) {
        const int n = n0 + tid;
        const bool live = n < a.N;
        ssl[tid] = (live && a.scale) ? a.scale[n] : (live ? 1.f : 0.f);
        ssl[NT + tid] = (live && a.shift) ? a.shift[n] : 0.f;
    }
    __syncthreads();
    int woff[KS];                                          // this lane's swizzled chunk of weight row r, per 16-channel slab
#pragma unroll
    for (int sl = 0; sl < KS; ++sl) woff[sl] = r * WR + (((2 * sl + h) ^ ((r >> WSH) & (VPP - 1))) * 16);
    u32x4 wreg[TR][KS][TN];
#pragma unroll
    for (int t = 0; t < TR; ++t)
#pragma unroll
        for (int sl = 0; sl < KS; ++sl)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                wreg[t][sl][j] = *reinterpret_cast<const u32x4*>(smem + t * WTAP + j * 32 * WR + woff[sl]);
    __syncthreads();                                       // every wave has its fragments: the staging image is dead
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
    const float lo = a.relu ? 0.f : -INFINITY;             // ReLU as max(f, lo)
    const auto out_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, out_bytes, 0x00020000);
    const auto add_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.add ? a.add : a.out), 0, out_bytes, 0x00020000);
    // this lane's byte offset inside a PPI-pixel piece of an output row, or beyond the buffer for a dead channel chunk
    const unsigned lane_off = chunk_live ? (unsigned)(sp * a.NS + n0 + sc8 * 8) * 2u : OOB;
    auto row_base = [&](int t, int u) -> unsigned {        // byte offset of pixel 0 of block u (tile row 2 wave + u) of tile t
        int y0, x0, b;
        tile_origin(t, y0, x0, b);
        return (unsigned)(((b * a.H + y0 + 2 * wave + u) * a.W + x0) * a.NS) * 2u;
    };
    auto wave_lds_fence = [&]() {                          // LDS hand-over between lanes of one wave: order, no barrier
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    u32x4 radd[NI];
    // epilogue group (j, g) of a finished accumulator set: 4 consecutive channels of this lane's pixel -> the output patch
    auto epi_group = [&](const f32x16 (&acc)[TN], int j, int g) {
        const int nl = j * 32 + 8 * g + 4 * h;
        const f32x4 sc = *reinterpret_cast<const f32x4*>(ssl + nl);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(ssl + NT + nl);
        float f[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) f[k] = acc[j][4 * g + k] * sc[k] + sh[k];
        if (has_add) {
            const bf16x4 q = *reinterpret_cast<const bf16x4*>(rsg + r * SRS + nl * 2);
#pragma unroll
            for (int k = 0; k < 4; ++k) f[k] += (float)q[k];
        }
        bf16x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = (bf16)vmax1(f[k], lo);
        *reinterpret_cast<bf16x4*>(stg + r * SRS + nl * 2) = o;
    };
    // base: byte offset of the block's pixel 0 (uniform), or OOB when there is nothing to store / fetch
    auto drain_piece = [&](unsigned base, int i) {
        const u32x4 piece = *reinterpret_cast<const u32x4*>(stg + (i * PPI + sp) * SRS + sc8 * 16);
        __builtin_amdgcn_raw_buffer_store_b128(piece, out_rsrc, lane_off, base + (unsigned)(i * PPI * a.NS * 2), 0);
    };
    auto fetch_add = [&](unsigned base, int i) {
        radd[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(add_rsrc, lane_off, base + (unsigned)(i * PPI * a.NS * 2), 0));
    };
    auto park_add = [&](int i) { *reinterpret_cast<u32x4*>(rsg + (i * PPI + sp) * SRS + sc8 * 16) = radd[i]; };

    // Side operations, placed at fixed MFMA steps of a tile (2 * NSTEP steps: block 0, then block 1).
    //   halo: vector k of the next tile is fetched at step hl_step(k) and written to the other buffer HDIST steps later.
    //   block u's own residual rows: fetched at its first NI steps, parked in the residual patch at its last steps.
    //   the block before u (block 0: the previous tile's block 1; block 1: this tile's block 0): epilogue groups from step
    //   E_FIRST on, one every E_EVERY steps, then the drain pieces.
    constexpr int HDIST = (NSTEP * 2) / 3;
    constexpr int E_FIRST = 4, E_EVERY = (NSTEP - E_FIRST - 2 * NI - 3) / NEPI > 0 ? (NSTEP - E_FIRST - 2 * NI - 3) / NEPI : 1;
    constexpr int E_LAST = E_FIRST + (NEPI - 1) * E_EVERY;
    constexpr int D_FIRST = E_LAST + 2;                   // fence at D_FIRST - 1
    constexpr int P_FIRST = NSTEP - NI - 1;               // parks, then the fence at NSTEP - 1
    static_assert(D_FIRST + NI <= NSTEP && P_FIRST > E_LAST, "a block has too few MFMA steps for its side operations");
    static_assert(TN * E_FIRST >= 4, "an accumulator set is read too soon after its last MFMA");
    auto hl_step = [](int k) constexpr { return k * (NS2 - HDIST) / HV; };
    f32x16 acc0[TN], acc1[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) { acc0[j][e] = 0.f; acc1[j][e] = 0.f; }

    int it = 0;
#pragma nounroll
    for (; it < nit; ++it, tile = tile_at(it)) {
        const int cur = it & 1;
        const unsigned char* hrow = smem + cur * HALO_BYTES + hrow0;
        // after the last tile the halo fetches run once more on the same tile (into the dead buffer): cheaper than a branch
        // around every one of them
        const int next = it + 1 < nit ? tile_at(it + 1) : tile, prev = tile_at(it - 1);
        const bool have_prev = it > 0;
        halo_tile(next);
        // uniform byte offsets of the rows this tile touches
        const unsigned out_prev1 = have_prev ? row_base(prev, 1) : OOB;        // drained during block 0
        const unsigned out_cur0 = row_base(tile, 0);                           // drained during block 1
        const unsigned add_cur0 = out_cur0, add_cur1 = row_base(tile, 1);
        {
            constexpr int PF = 2;
            u32x4 pv[PF + 1], wl[2][TN];                   // wl: LDS-resident taps' fragments, one step ahead
            auto frag = [&](int s) -> u32x4 {              // pixel fragment of flattened step s = (block, tap, slab)
                const int u = s / NSTEP, st = s % NSTEP, t = st / KS, sl = st % KS;
                return *reinterpret_cast<const u32x4*>(hrow + u * (HWD * RS) + toff[t] + 32 * sl);
            };
            auto wfrag = [&](int s, u32x4 (&w)[TN]) {
                const int st = s % NSTEP, t = st / KS, sl = st % KS;
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    w[j] = *reinterpret_cast<const u32x4*>(wtl + (t - TR) * WTAP + j * 32 * WR + woff[sl]);
            };
#pragma unroll
            for (int s = 0; s < PF; ++s) pv[s] = frag(s);
#pragma unroll
            for (int s = 0; s < NS2; ++s) {
                const int u = s / NSTEP, st = s % NSTEP, t = st / KS, sl = st % KS;
                if (s + PF < NS2) pv[(s + PF) % (PF + 1)] = frag(s + PF);
                if (WTL > 0 && s + 1 < NS2 && ((s + 1) % NSTEP) / KS >= TR) wfrag(s + 1, wl[(s + 1) % 2]);
                // ---- side operations of this step ----
#pragma unroll
                for (int k = 0; k < HV; ++k) {
                    if (s == hl_step(k)) load_halo_vec(k);
                    if (s == hl_step(k) + HDIST) store_halo_vec(cur ^ 1, k);
                }
                if (has_add) {
#pragma unroll
                    for (int i = 0; i < NI; ++i) {
                        if (st == i) fetch_add(u == 0 ? add_cur0 : add_cur1, i);
                        if (st == P_FIRST + i) park_add(i);
                    }
                }
#pragma unroll
                for (int e = 0; e < NEPI; ++e)
                    if (st == E_FIRST + e * E_EVERY) { if (u == 0) epi_group(acc1, e / 4, e % 4); else epi_group(acc0, e / 4, e % 4); }
                if (st == D_FIRST - 1) wave_lds_fence();
#pragma unroll
                for (int i = 0; i < NI; ++i)
                    if (st == D_FIRST + i) drain_piece(u == 0 ? out_prev1 : out_cur0, i);
                if (st == NSTEP - 1) wave_lds_fence();
                // ---- the MFMAs of this step ----
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    f32x16& acc = u == 0 ? acc0[j] : acc1[j];
                    if (t < TR) {
                        if (st == 0) mfma_first_a(acc, wreg[t < TR ? t : 0][sl][j], pv[s % (PF + 1)]);
                        else mfma_acc_a(acc, wreg[t < TR ? t : 0][sl][j], pv[s % (PF + 1)]);
                    } else {
                        if (st == 0) mfma_first_v(acc, wl[s % 2][j], pv[s % (PF + 1)]);
                        else mfma_acc_v(acc, wl[s % 2][j], pv[s % (PF + 1)]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);         // keep the side operations where they were placed
            }
        }
        __syncthreads();                                   // next halo complete; every wave is done reading this one
    }
    // ---- block 1 of the last tile is still in its accumulators (its residual rows are parked already) ----
    if (it > 0) {
        const int last = tile_at(it - 1);
        const unsigned base = row_base(last, 1);
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // the last MFMA's result -> the VALU reads below
#pragma unroll
        for (int e = 0; e < NEPI; ++e) epi_group(acc1, e / 4, e % 4);
        wave_lds_fence();
#pragma unroll
        for (int i = 0; i < NI; ++i) drain_piece(base, i);
    }
}

static constexpr int wreg_lds_taps(int CIN, int NT) { return (CIN == 64 && NT == 64) ? 1 : 0; }
static constexpr int wreg_lds_bytes_of(int CIN, int NT) {
    // two halo buffers + epilogue scale/shift + per wave an output and a residual patch of one block + the LDS-resident taps
    return 2 * (10 * 34) * (CIN * 2 + 16) + 2 * NT * 4 + 8 * 32 * (NT * 2 + 16) + wreg_lds_taps(CIN, NT) * NT * CIN * 2;
}

// true if the plan (a halo-kernel plan with 32-wide tiles) runs on the weights-in-registers kernel
static bool wreg_applicable(const GemmPlan& P) {
    const HaloGeom hg = halo_geom(P);
    if (hg.TW != 32 || P.Hout % 8) return false;
    return ilog2_exact(P.Hout / 8) >= 0;
}

static bool try_launch_wreg_bf16(const GemmPlan& P, const void* in, const void* wp, void* out, const float* scale,
                                 const float* shift, const void* res, int relu, int accumulate, int out_f32, hipStream_t st) {
    // (ReLU before the addition of the old output is the one order this kernel does not have; nothing asks for it)
    if (out_f32 || !wreg_applicable(P) || (res && accumulate) || (relu && accumulate)) return false;
    if ((long)P.B * P.Hout * P.Wout * (P.CgS > P.NS ? P.CgS : P.NS) * 2 >= 0xfffffff0L) return false;
    const HaloGeom hg = halo_geom(P);
    const Phase& ph = P.ph[0];
    const int NT = hg.NT;
    HaloArgs a;
    memset(&a, 0, sizeof(a));
    a.in = in; a.wp = wp; a.out = out; a.scale = scale; a.shift = shift;
    a.add = res ? res : (accumulate ? out : nullptr); a.add_after_relu = 0;
    a.B = P.B; a.H = P.Hout; a.W = P.Wout; a.N = P.N; a.NS = P.NS; a.Ktot = P.Ktot;
    a.relu = relu; a.out_f32 = 0;
    a.lx = ilog2_exact(P.Wout / 32); a.ly = ilog2_exact(P.Hout / 8);
    memcpy(a.dy, ph.dy, 16); memcpy(a.dx, ph.dx, 16);
    const int ntiles = P.B << (a.lx + a.ly);
    const int gx = ntiles < 256 ? ntiles : 256;            // persistent: one workgroup per CU
    dim3 grid(gx, P.Npad / NT), block(256);
#define WREG_CASE(C, N, A)                                                                               \
    if (P.CgS == C && NT == N && (a.add != nullptr) == A) {                                              \
        constexpr int lds = wreg_lds_bytes_of(C, N);                                                     \
        static LdsOptIn lds_opt;                                                                         \
        lds_opt.ensure(reinterpret_cast<const void*>(&conv_wreg_bf16_kernel<C, N, wreg_lds_taps(C, N), A>), lds); \
        hipLaunchKernelGGL((conv_wreg_bf16_kernel<C, N, wreg_lds_taps(C, N), A>), grid, block, lds, st, a); \
        return true;                                                                                     \
    }
    WREG_CASE(64, 64, false) WREG_CASE(64, 32, false) WREG_CASE(32, 64, false) WREG_CASE(32, 32, false)
    WREG_CASE(64, 64, true) WREG_CASE(64, 32, true) WREG_CASE(32, 64, true) WREG_CASE(32, 32, true)
#undef WREG_CASE
    return false;
}

}  // namespace dvae
