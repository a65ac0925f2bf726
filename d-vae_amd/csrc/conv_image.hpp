// bf16 transposed-gather convolution with a 16x16 tile of the input (a whole image, for 16x16 maps) resident in LDS.
//
// ConvTranspose2d(k4, s2, p1) forward -- and the input-gradient of a stride-2 Conv2d -- decompose into output-parity
// phases: phase (py, px) is a 2x2-tap unit-stride convolution of the 16x16 input that writes every other output
// pixel.  All phases read the same input, so one workgroup keeps the whole 18x18 halo of one tile (128 channels) in
// LDS for all of them, and streams the weights through LDS one (phase, tap) slab of 64 x 128 at a time, double-buffered:
// slabs are fetched into a register ring three steps ahead (an L2 round trip under load is longer than one step's
// MFMAs) and written to the idle LDS buffer one step ahead, before the step's barrier.  The gather GEMM of igemm.hpp re-fetches the input rows once per tap and per phase through L2 and
// synchronises twice per 64-deep K slab; here an image's input is read from HBM once and a barrier buys 16 MFMAs per wave.
//
// Persistent workgroups (one per CU, 8 waves) walk the tiles: maps whose sides are multiples of 16 are cut into 16x16
// tiles of the phase grid, a tile's halo comes from its neighbours (zeros outside the map).  A wave owns 32 pixels of the 16x16 phase grid and all
// 64 output channels; the MFMA is issued as D[channel][pixel] = W * X^T, the finished phase is parked as bf16 in the
// wave's private LDS patch and drained to HBM as whole 128-byte pixel rows from inside the next step's MFMA loop, where
// the next image's halo is prefetched into registers as well.
#pragma once
#include <type_traits>

#include "conv_halo.hpp"
#include "igemm.hpp"

namespace dvae {

struct ImageConvArgs {
    const void* in;
    const void* wp;
    void* out;
    const float* shift;      // per-output-channel bias or NULL
    const float* scale;      // per-output-channel scale or NULL (FULL kernels)
    const void* res;         // residual, laid out like out, added before the ReLU, or NULL (FULL kernels)
    int relu;
    int B, Hout, Wout, N, NS, Ktot;
    int Hin, Win;            // input map = phase grid, multiples of 16
    TileGrid g;              // 16x16 tiles of the input map
    int nslab, DS;           // (phase, tap) slabs per image; output step of a phase grid step
    // per slab: halo offset of the tap (pixels), first K column in the packed rows, phase << 1 | last tap of its phase;
    // per phase: output offset.  Copied to LDS at kernel start: indexing kernel arguments with a run-time index costs a
    // vector load and a wait on every outstanding load each time
    int slab_toff[32], slab_k0[32], slab_phase[32];
    int phase_d0y[4], phase_d0x[4], phase_n0[4];   // per phase: first output pixel and first output channel
    // fused encoder head (ZF kernels): `in` is the stem output, the weight rows are [mu | log_var] of a 1x1 conv with
    // 2 * zL outputs, `shift` its 2 * zL biases; slab q computes mu and log_var of latent channels [32 q, 32 q + 32) and
    // the epilogue writes z = eps * exp(log_var / 2) + mu (bf16, [B][256][zL]) with eps drawn in place
    int zL;
    PhiloxKey zkey;
};

// NSLAB (phase, tap) slabs per image; bit q of LASTMASK: slab q is the last tap of its phase.  Both are compile-time so
// that an image is one straight-line block: with the register ring, the halo prefetch and the drain stores all in
// flight at once, the wait counts the compiler derives are exact only if no branch separates a load from its use.
// FULL: per-channel scale, residual and more than NT output channels (a "phase" is then a block of NT channels of the
// same pixels): the folded-BatchNorm + shortcut + ReLU epilogue of a residual block's second conv.
template <int CIN, int NT, int NSLAB, unsigned LASTMASK, bool ZF = false, bool FULL = false>
__global__ __launch_bounds__(512) void conv_image_bf16_kernel(const ImageConvArgs a) {
    constexpr int S = 16, HWD = S + 2, HP = HWD * HWD;    // input image side; halo side; halo pixels
    constexpr int RS = CIN * 2 + 16;                      // bytes per staged pixel / weight row (16-byte pad spreads banks)
    constexpr int VPP = CIN * 2 / 16;                     // 16-byte vectors per pixel / weight row
    constexpr int HALO_BYTES = HP * RS, WBUF = NT * RS;
    constexpr int TN = NT / 32, KS = CIN / 16;
    constexpr int HV = (HP * VPP + 511) / 512;            // halo vectors per thread
    constexpr int WV = NT * VPP / 512;                    // weight-slab vectors per thread
    constexpr int OC = ZF ? NT / 2 : NT;                  // channels a finished slab group leaves in the patch
    constexpr int SSLN = ZF ? 16 * NSLAB * 4 : (FULL ? 4 * NT : NT);   // biases in LDS (ZF: all 2 zL = 64 NSLAB; FULL: shift | scale of 2 NT)
    static_assert(!(ZF && FULL), "one epilogue flavour per kernel");
    constexpr int SRS = OC * 2 + 16, VPS = OC / 8, PPI = 64 / VPS, NI = 32 / PPI;
    static_assert(!ZF || LASTMASK == (1u << NSLAB) - 1u, "fused head: every slab is a finished block of channels");
    constexpr int RING = (NSLAB % 4 == 0) ? 4 : 3;        // slabs in flight in registers; divides NSLAB: static slots
    static_assert(NSLAB % RING == 0, "ring slots must be static per slab");
    static_assert(NT * VPP % 512 == 0 && NI + 1 <= KS && HV <= 2 * KS && NSLAB >= 2, "slab split / drain and prefetch slots");
    static_assert((LASTMASK >> (NSLAB - 1)) & 1u, "the last slab of an image ends a phase");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* halo = smem;                                     // [HP][RS]
    unsigned char* wbuf = smem + HALO_BYTES;                        // [2][NT][RS]
    float* ssl = reinterpret_cast<float*>(smem + HALO_BYTES + 2 * WBUF);   // shift[NT]
    int* tab = reinterpret_cast<int*>(smem + HALO_BYTES + 2 * WBUF + SSLN * 4 + 8 * 32 * SRS);   // [32][4] slabs, then [4][4] phases
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    unsigned char* stg = smem + HALO_BYTES + 2 * WBUF + SSLN * 4 + wave * (32 * SRS);

    const bf16* in = reinterpret_cast<const bf16*>(a.in);
    const bf16* wp = reinterpret_cast<const bf16*>(a.wp);
    bf16* outb = reinterpret_cast<bf16*>(a.out);
    if (ZF) {
        if (tid < SSLN) ssl[tid] = a.shift ? a.shift[tid] : 0.f;
    } else if (FULL) {
        if (tid < 2 * NT) {
            ssl[tid] = (tid < a.N && a.shift) ? a.shift[tid] : 0.f;
            ssl[2 * NT + tid] = tid < a.N ? (a.scale ? a.scale[tid] : 1.f) : 0.f;
        }
    } else if (tid < NT) {
        ssl[tid] = (tid < a.N && a.shift) ? a.shift[tid] : 0.f;
    }
    if (tid < 32) { tab[tid * 4 + 0] = a.slab_toff[tid] * RS; tab[tid * 4 + 1] = a.slab_k0[tid]; }
    if (tid < 4) { tab[128 + tid * 4] = a.phase_d0y[tid]; tab[128 + tid * 4 + 1] = a.phase_d0x[tid]; tab[128 + tid * 4 + 2] = a.phase_n0[tid]; }
    __syncthreads();

    // ---- halo: HV vectors per thread, border pixels (outside the image) are zeros ----
    u32x4 hreg[HV];
    auto halo_pos = [&](int k, int& hp, int& cv) { const int v = tid + 512 * k; hp = v / VPP; cv = v % VPP; };
    // byte offset of halo vector k from the tile's halo corner (pixel (-1, -1) of the tile): 32 bits and never negative,
    // the tile base is uniform and stays in scalar registers.  Bit k of m_top / m_bot / m_left / m_right: vector k sits
    // in that border of the halo (outside the map when the tile touches that edge); m_tail: past the halo's end.
    static_assert(512 % VPP == 0, "a thread's halo vectors all cover the same channels");
    unsigned hofs[HV];
    unsigned m_top = 0u, m_bot = 0u, m_left = 0u, m_right = 0u, m_tail = 0u;
#pragma unroll
    for (int k = 0; k < HV; ++k) {
        int hp, cv;
        halo_pos(k, hp, cv);
        const int hy = hp / HWD, hx = hp % HWD;
        if (hp >= HP) m_tail |= 1u << k;
        else {
            if (hy == 0) m_top |= 1u << k;
            if (hy == HWD - 1) m_bot |= 1u << k;
            if (hx == 0) m_left |= 1u << k;
            if (hx == HWD - 1) m_right |= 1u << k;
        }
        hofs[k] = (unsigned)((hy * a.Win + hx) * CIN + cv * 8) * 2u;
    }
    // border / tail vectors read the tile's first pixel and are zeroed on the way to LDS: no branch around the load
    const unsigned hsafe = (unsigned)((a.Win + 1) * CIN + (tid % VPP) * 8) * 2u;
    // what a workgroup needs to know of a tile (all uniform but `bad`)
    struct Tile {
        const unsigned char* ib;   // the tile's halo corner in the input
        long ob;                   // first output pixel (phase 0) of the tile, in pixels of the whole output tensor
        unsigned bad;              // bit k: halo vector k of this thread is outside the map
        unsigned img;              // image the tile belongs to
        int pix0;                  // the tile's first pixel inside its image
    };
    auto tile_at = [&](int t) {
        int img, ty, tx;
        a.g.split(t, img, ty, tx);
        Tile T;
        T.img = (unsigned)img;
        T.pix0 = (ty * a.Win + tx) * S;
        T.ib = reinterpret_cast<const unsigned char*>(in + ((long)img * (a.Hin * a.Win) + T.pix0 - (a.Win + 1)) * CIN);
        T.ob = (long)img * (a.Hout * a.Wout) + (long)(ty * a.Wout + tx) * (S * a.DS);
        T.bad = m_tail | (ty == 0 ? m_top : 0u) | ((ty + 1) * S == a.Hin ? m_bot : 0u) | (tx == 0 ? m_left : 0u) |
                ((tx + 1) * S == a.Win ? m_right : 0u);
        return T;
    };
    auto load_halo_vec = [&](const Tile& T, int k) {
        hreg[k] = *reinterpret_cast<const u32x4*>(T.ib + (((T.bad >> k) & 1u) ? hsafe : hofs[k]));
    };
    auto store_halo = [&](unsigned bad) {
#pragma unroll
        for (int k = 0; k < HV; ++k) {
            int hp, cv;
            halo_pos(k, hp, cv);
            if (hp >= HP) continue;
            const u32x4 z = {0u, 0u, 0u, 0u};
            *reinterpret_cast<u32x4*>(halo + hp * RS + cv * 16) = ((bad >> k) & 1u) ? z : hreg[k];
        }
    };
    // ---- weight slab q = (phase, tap): NT rows of CIN, WV vectors per thread, ring slot q % RING ----
    u32x4 wreg[RING][WV];
    // this thread's vectors of a slab: a per-thread byte offset that does not depend on the slab, plus a uniform slab
    // base (so a slab costs no vector address arithmetic and no address registers of its own)
    unsigned wofs[WV];
#pragma unroll
    for (int k = 0; k < WV; ++k) {
        const int v = tid + 512 * k, n = v / VPP, cv = v % VPP;
        // fused head: rows 0..31 of a slab are mu of 32 channels, rows 32..63 the same channels' log_var
        const int row = ZF ? (n >> 5) * a.zL + (n & 31) : n;
        wofs[k] = (unsigned)(row * a.Ktot + cv * 8) * 2u;
    }
    auto load_slab = [&](int slot, int q) {                // slot, q: compile-time at every call site
        const int k0 = __builtin_amdgcn_readfirstlane(tab[q * 4 + 1]);
        // (read from the LDS table, the slab base is not a loop invariant the compiler could hoist: it would keep a
        // 64-bit address per slab and vector in registers for the whole kernel)
        const unsigned char* sbase = reinterpret_cast<const unsigned char*>(wp) + (long)k0 * 2;
#pragma unroll
        for (int k = 0; k < WV; ++k) wreg[slot][k] = *reinterpret_cast<const u32x4*>(sbase + wofs[k]);
    };
    auto store_slab = [&](int slot, int buf) {
#pragma unroll
        for (int k = 0; k < WV; ++k) {
            const int v = tid + 512 * k, n = v / VPP, cv = v % VPP;
            *reinterpret_cast<u32x4*>(wbuf + buf * WBUF + n * RS + cv * 16) = wreg[slot][k];
        }
    };

    // wave w owns the tile's phase-grid pixels [32w, 32w + 32): rows 2w, 2w + 1; lane r is pixel 32w + r
    const int gy = 2 * wave + (r >> 4), gx = r & 15;
    const unsigned char* hrow = halo + ((gy + 1) * HWD + gx + 1) * RS + 16 * h;
    const unsigned char* wrow = wbuf + r * RS + 16 * h;
    const int sp = lane / VPS, sc8 = lane % VPS;           // coalesced side: lane = (pixel sp of a PPI group, 8-channel chunk sc8)
    const bool chunk_live = ZF || FULL || sc8 * 8 < a.NS;
    // output pixel of this lane's 16-byte piece i, relative to the phase's first pixel; and each phase's first pixel
    // relative to the image (uniform, kept in scalar registers: the drain slots sit between MFMAs and must not wait on LDS)
    int lpix[NI], d0[4], pn0[4];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int q = i * PPI + sp;                        // pixel of the wave's 32
        lpix[i] = ((2 * wave + (q >> 4)) * a.Wout + (q & 15)) * a.DS;
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        d0[p] = __builtin_amdgcn_readfirstlane(tab[128 + p * 4]) * a.Wout + __builtin_amdgcn_readfirstlane(tab[128 + p * 4 + 1]);
        pn0[p] = FULL ? __builtin_amdgcn_readfirstlane(tab[128 + p * 4 + 2]) : 0;
    }
    auto out_off = [&](long ob, int p, int i) -> long {    // ob: the tile's Tile::ob; p, i: compile-time at every call site
        if (ZF) return (ob + lpix[i]) * a.zL + 32 * p + sc8 * 8;
        return (ob + d0[p] + lpix[i]) * a.NS + pn0[p] + sc8 * 8;
    };
    const int lanepix = gy * a.Win + gx;                   // this lane's pixel, relative to the tile's first
    const bf16* resb = reinterpret_cast<const bf16*>(a.res);
    u32x4 radd[NI];
    auto wave_lds_fence = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    auto phase_of = [](int q) { return __builtin_popcount(LASTMASK & ((1u << q) - 1u)); };

    const int G = gridDim.x, ntiles = a.B * a.g.tpi;
    int b = blockIdx.x;                                    // tile
    if (b >= ntiles) return;                               // whole workgroup: grid <= tiles, never taken
    Tile cur = tile_at(b);
#pragma unroll
    for (int k = 0; k < HV; ++k) load_halo_vec(cur, k);
#pragma unroll
    for (int i = 0; i < RING; ++i) load_slab(i, i % NSLAB);
    store_slab(0, 0);
    bool have_prev = false;                                // an earlier tile's last phase is parked in the patch
    long pob = cur.ob;
    int par = 0;                                           // LDS buffer of an image's slab 0 (alternates when NSLAB is odd)
    u32x4 dreg;
    f32x16 acc[TN];
    for (; b < ntiles; b += G) {
        store_halo(cur.bad);
        __syncthreads();                                   // halo (and, first time, slab 0 and ssl) visible
        const Tile nxt = tile_at(b + G < ntiles ? b + G : b);   // next tile, or this one again: the prefetch is unconditional
#pragma unroll
        for (int q = 0; q < NSLAB; ++q) {
            const int buf = (q & 1) ^ par;
            const bool first_tap = q == 0 || ((LASTMASK >> (q - 1)) & 1u);
            const bool last_tap = (LASTMASK >> q) & 1u;
            if (first_tap) {                               // a new phase: the accumulators start from the bias
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 sh = *reinterpret_cast<const f32x4*>(ssl + (ZF ? j * a.zL + 32 * q : j * 32) + 8 * g + 4 * h);
#pragma unroll
                        for (int k = 0; k < 4; ++k) acc[j][4 * g + k] = FULL ? 0.f : sh[k];   // FULL: scale, then shift, at the end
                    }
            }
            // this slab's ring slot was emptied into LDS a step ago: refill it with the slab RING steps ahead
            load_slab(q % RING, (q + RING) % NSLAB);
            const int toff = __builtin_amdgcn_readfirstlane(tab[q * 4]);
            const unsigned char* hp_ = hrow + toff;
            const unsigned char* wp_ = wrow + buf * WBUF;
            // what is parked in the patch: the previous slab's phase if that slab ended one
            const bool drain_static = q > 0 && ((LASTMASK >> (q - 1)) & 1u);
            const int dp = q > 0 ? phase_of(q - 1) : phase_of(NSLAB - 1);
            const long dob = q > 0 ? cur.ob : pob;
            bf16x8 pv[2], wv[2][TN];
            pv[0] = *reinterpret_cast<const bf16x8*>(hp_);
#pragma unroll
            for (int j = 0; j < TN; ++j) wv[0][j] = *reinterpret_cast<const bf16x8*>(wp_ + j * 32 * RS);
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                if (s + 1 < KS) {
                    pv[(s + 1) & 1] = *reinterpret_cast<const bf16x8*>(hp_ + 32 * (s + 1));
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        wv[(s + 1) & 1][j] = *reinterpret_cast<const bf16x8*>(wp_ + j * 32 * RS + 32 * (s + 1));
                }
                if (s <= NI && (drain_static || (q == 0 && have_prev))) {   // drain slot s of NI + 1: patch -> HBM
                    if (s > 0 && chunk_live) *reinterpret_cast<u32x4*>(outb + out_off(dob, dp, s - 1)) = dreg;
                    if (s < NI) dreg = *reinterpret_cast<const u32x4*>(stg + (s * PPI + sp) * SRS + sc8 * 16);
                }
                // the residual rows of a phase, in the second half of its last slab
                if (FULL && last_tap && s >= KS - NI) {
                    const int i = s - (KS - NI);
                    const u32x4 z = {0u, 0u, 0u, 0u};
                    radd[i] = a.res ? *reinterpret_cast<const u32x4*>(resb + out_off(cur.ob, phase_of(q), i)) : z;
                }
                // the next image's halo: in the image's first two slabs, or (fused head: its epilogue wants the registers,
                // and its slabs are long enough to cover the round trip) in the last two
                constexpr int HQ = ZF ? NSLAB - 2 : 0;
                if (q == HQ && s < HV) load_halo_vec(nxt, s);
                if (q == HQ + 1 && KS + s < HV) load_halo_vec(nxt, KS + s);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wv[s & 1][j], pv[s & 1], acc[j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (last_tap && ZF) {
                // ---- mu (acc[0]) and log_var (acc[1]) of 32 latent channels: draw eps, park z in the patch ----
                const uint64_t img = a.zkey.index ? (uint64_t)a.zkey.index[cur.img] : (uint64_t)(a.zkey.first + cur.img);
                unsigned e0 = (unsigned)(cur.pix0 + lanepix) * (unsigned)a.zL + 32u * q;   // first element of this lane pair's 32
                // the counter does not depend on the image, so the first Philox round of every (slab, quad) is a loop
                // invariant: hoisted, it would sit in 64 registers for the whole kernel
                asm volatile("" : "+v"(e0));
                if (a.zkey.stream_id & PHILOX_STREAM_16) {
                    f32x4 e[4];
                    philox_lane_quads(a.zkey, img, e0, h, e);
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        bf16x4 o;
#pragma unroll
                        for (int k = 0; k < 4; ++k) o[k] = (bf16)(e[g][k] * __expf(0.5f * acc[1][4 * g + k]) + acc[0][4 * g + k]);
                        *reinterpret_cast<bf16x4*>(stg + r * SRS + (8 * g + 4 * h) * 2) = o;
                    }
                } else {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 e = philox_normal4(a.zkey, img, (e0 + 8u * g + 4u * h) >> 2);
                        bf16x4 o;
#pragma unroll
                        for (int k = 0; k < 4; ++k) o[k] = (bf16)(e[k] * __expf(0.5f * acc[1][4 * g + k]) + acc[0][4 * g + k]);
                        *reinterpret_cast<bf16x4*>(stg + r * SRS + (8 * g + 4 * h) * 2) = o;
                        __builtin_amdgcn_sched_barrier(0);     // one Philox chain at a time: registers, not latency, are short
                    }
                }
                wave_lds_fence();
            }
            if (last_tap && FULL) {
                // ---- scale / shift, then the residual: through the patch (coalesced rows in, register layout out) ----
                const int n0 = phase_of(q) * NT;
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 sh = *reinterpret_cast<const f32x4*>(ssl + n0 + j * 32 + 8 * g + 4 * h);
                        const f32x4 sc = *reinterpret_cast<const f32x4*>(ssl + 2 * NT + n0 + j * 32 + 8 * g + 4 * h);
#pragma unroll
                        for (int k = 0; k < 4; ++k) acc[j][4 * g + k] = acc[j][4 * g + k] * sc[k] + sh[k];
                    }
                if (a.res) {
#pragma unroll
                    for (int i = 0; i < NI; ++i) *reinterpret_cast<u32x4*>(stg + (i * PPI + sp) * SRS + sc8 * 16) = radd[i];
                    wave_lds_fence();
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const bf16x4 rq = *reinterpret_cast<const bf16x4*>(stg + r * SRS + (j * 32 + 8 * g + 4 * h) * 2);
#pragma unroll
                            for (int k = 0; k < 4; ++k) acc[j][4 * g + k] += (float)rq[k];
                        }
                    wave_lds_fence();
                }
            }
            if (last_tap && !ZF) {
                // ---- the phase is finished: ReLU, bf16, park in the patch (drained inside the next slab) ----
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        bf16x4 o;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            float v = acc[j][4 * g + k];
                            if (a.relu) v = fmaxf(v, 0.f);
                            o[k] = (bf16)v;
                        }
                        *reinterpret_cast<bf16x4*>(stg + r * SRS + (j * 32 + 8 * g + 4 * h) * 2) = o;
                    }
                wave_lds_fence();
            }
            store_slab((q + 1) % RING, buf ^ 1);           // slab q + 1: that buffer was last read before the previous barrier
            __syncthreads();
        }
        have_prev = true; pob = cur.ob;
        cur = nxt;
        if (NSLAB & 1) par ^= 1;
    }
    if (chunk_live) {                                      // the last phase of the last tile
#pragma unroll
        for (int i = 0; i < NI; ++i)
            *reinterpret_cast<u32x4*>(outb + out_off(pob, phase_of(NSLAB - 1), i)) =
                *reinterpret_cast<const u32x4*>(stg + (i * PPI + sp) * SRS + sc8 * 16);
    }
}

template <int CIN, int NT>
static constexpr int image_lds_bytes() {
    return 18 * 18 * (CIN * 2 + 16) + 2 * NT * (CIN * 2 + 16) + NT * 4 + 8 * 32 * (NT * 2 + 16) + (32 * 4 + 4 * 4) * 4;
}
template <int CIN, int NT>
static constexpr int image_full_lds_bytes() { return image_lds_bytes<CIN, NT>() + 3 * NT * 4; }
template <int CIN, int NT, int NSLAB>
static constexpr int image_zf_lds_bytes() {
    return 18 * 18 * (CIN * 2 + 16) + 2 * NT * (CIN * 2 + 16) + 64 * NSLAB * 4 + 8 * 32 * (NT + 16) + (32 * 4 + 4 * 4) * 4;
}

// The input map of a plan cut into 16x16 tiles; false if its sides are not multiples of 16 or the tile index
// arithmetic would not be exact for this batch.
static bool image_tiles(const GemmPlan& P, ImageConvArgs& a) {
    if (P.Hin < 16 || P.Win < 16 || P.Hin % 16 || P.Win % 16) return false;
    a.Hin = P.Hin; a.Win = P.Win;
    return make_tile_grid(P.Win / 16, P.Hin / 16, ((long)P.B + 1) * (P.Hin / 16) * (P.Win / 16), a.g);
}
static dim3 image_grid(const ImageConvArgs& a) {
    const long nt = (long)a.B * a.g.tpi;
    return dim3((unsigned)(nt < 256 ? nt : 256));
}

// Slab structure of a plan this kernel is built for: a transposed gather over phase grids of the input's size with 128
// gathered and <= 64 produced channels whose (slabs per tile, last-tap mask) is one of the compiled pairs.  0 slabs: not
// applicable.
struct ImageGeom { int nslab; unsigned lastmask; };
static ImageGeom image_geom(const GemmPlan& P) {
    const ImageGeom none = {0, 0u};
    ImageConvArgs probe;
    if (P.CgS != 128 || P.Npad != 64 || !image_tiles(P, probe) || P.nphase < 1) return none;
    ImageGeom g = {0, 0u};
    for (int i = 0; i < P.nphase; ++i) {
        const Phase& ph = P.ph[i];
        if (ph.SY != 1 || ph.SX != 1 || ph.Hg != P.Hin || ph.Wg != P.Win || ph.ntaps < 1) return none;
        if (ph.DSY != P.ph[0].DSY || ph.DSX != P.ph[0].DSY) return none;
        for (int t = 0; t < ph.ntaps; ++t)
            if (ph.dy[t] < -1 || ph.dy[t] > 1 || ph.dx[t] < -1 || ph.dx[t] > 1) return none;
        g.nslab += ph.ntaps;
        if (g.nslab > 16) return none;
        g.lastmask |= 1u << (g.nslab - 1);
    }
    const bool built = (g.nslab == 16 && g.lastmask == 0x8888u) ||     // ConvTranspose2d k4 s2 p1 forward
                       (g.nslab == 9 && g.lastmask == 0x115u);         // Conv2d k3 s2 p1 input-gradient
    return built ? g : none;
}
static bool image_applicable(const GemmPlan& P) { return image_geom(P).nslab != 0; }

// 3x3 unit-stride conv 128 -> 128 (a residual block's convs in layer2): two blocks of 64 output
// channels x 9 taps = 18 slabs, with the scale / shift / residual / ReLU epilogue.
static bool image_full_applicable(const GemmPlan& P) {
    ImageConvArgs probe;
    if (P.nphase != 1 || P.CgS != 128 || P.N != 128 || P.NS != 128 || !image_tiles(P, probe) || P.Hout != P.Hin || P.Wout != P.Win) return false;
    const Phase& ph = P.ph[0];
    if (ph.ntaps != 9 || ph.SY != 1 || ph.SX != 1 || ph.DSY != 1 || ph.DSX != 1 || ph.D0Y || ph.D0X || ph.Hg != P.Hin || ph.Wg != P.Win) return false;
    for (int t = 0; t < 9; ++t)
        if (ph.dy[t] < -1 || ph.dy[t] > 1 || ph.dx[t] < -1 || ph.dx[t] > 1) return false;
    return true;
}

static bool try_launch_image_full(const GemmPlan& P, const void* in, const void* wp, void* out, const float* scale,
                                  const float* shift, const void* res, int relu, hipStream_t st) {
    if (!image_full_applicable(P)) return false;
    ImageConvArgs a;
    memset(&a, 0, sizeof(a));
    a.in = in; a.wp = wp; a.out = out; a.shift = shift; a.scale = scale; a.res = res; a.relu = relu;
    a.B = P.B; a.Hout = P.Hout; a.Wout = P.Wout; a.N = P.N; a.NS = P.NS; a.Ktot = P.Ktot;
    image_tiles(P, a);
    a.nslab = 18; a.DS = 1;
    const Phase& ph = P.ph[0];
    for (int half = 0; half < 2; ++half) {
        a.phase_n0[half] = 64 * half;
        for (int t = 0; t < 9; ++t) {
            const int q = half * 9 + t;
            a.slab_toff[q] = (int)ph.dy[t] * 18 + (int)ph.dx[t];
            a.slab_k0[q] = ph.koff + t * 128 + 64 * half * P.Ktot;       // weight rows 64 half .. of tap t
            a.slab_phase[q] = half << 1 | (t == 8 ? 1 : 0);
        }
    }
    constexpr int lds = image_full_lds_bytes<128, 64>();
    static_assert(lds <= 160 * 1024, "LDS budget");
    static LdsOptIn lds_opt;
    lds_opt.ensure(reinterpret_cast<const void*>(&conv_image_bf16_kernel<128, 64, 18, 0x20100u, false, true>), lds);
    dim3 grid = image_grid(a), block(512);
    hipLaunchKernelGGL((conv_image_bf16_kernel<128, 64, 18, 0x20100u, false, true>), grid, block, lds, st, a);
    return true;
}

static bool try_launch_image_bf16(const GemmPlan& P, const void* in, const void* wp, void* out, const float* scale,
                                  const float* shift, const void* res, int relu, int accumulate, int out_f32,
                                  hipStream_t st) {
    if (!accumulate && !out_f32 && try_launch_image_full(P, in, wp, out, scale, shift, res, relu, st)) return true;
    const ImageGeom g = image_geom(P);
    if (scale || res || accumulate || out_f32 || !g.nslab) return false;
    ImageConvArgs a;
    memset(&a, 0, sizeof(a));
    a.in = in; a.wp = wp; a.out = out; a.shift = shift; a.relu = relu;
    a.B = P.B; a.Hout = P.Hout; a.Wout = P.Wout; a.N = P.N; a.NS = P.NS; a.Ktot = P.Ktot;
    image_tiles(P, a);
    memset(a.slab_toff, 0, sizeof(a.slab_toff)); memset(a.slab_k0, 0, sizeof(a.slab_k0));
    memset(a.slab_phase, 0, sizeof(a.slab_phase)); memset(a.phase_d0y, 0, sizeof(a.phase_d0y)); memset(a.phase_d0x, 0, sizeof(a.phase_d0x));
    a.nslab = 0; a.DS = P.ph[0].DSY;
    for (int i = 0; i < P.nphase; ++i) {
        const Phase& ph = P.ph[i];
        a.phase_d0y[i] = ph.D0Y; a.phase_d0x[i] = ph.D0X;
        for (int t = 0; t < ph.ntaps; ++t, ++a.nslab) {
            a.slab_toff[a.nslab] = (int)ph.dy[t] * 18 + (int)ph.dx[t];
            a.slab_k0[a.nslab] = ph.koff + t * 128;
            a.slab_phase[a.nslab] = i << 1 | (t + 1 == ph.ntaps ? 1 : 0);
        }
    }
    constexpr int lds = image_lds_bytes<128, 64>();
    static_assert(lds <= 160 * 1024, "LDS budget");
    dim3 grid = image_grid(a), block(512);
#define IMAGE_CASE(NS_, MASK_)                                                                                      \
    if (g.nslab == NS_ && g.lastmask == MASK_) {                                                                    \
        static LdsOptIn lds_opt;                                                                                    \
        lds_opt.ensure(reinterpret_cast<const void*>(&conv_image_bf16_kernel<128, 64, NS_, MASK_>), lds);           \
        hipLaunchKernelGGL((conv_image_bf16_kernel<128, 64, NS_, MASK_>), grid, block, lds, st, a);                 \
        return true;                                                                                                \
    }
    IMAGE_CASE(16, 0x8888u) IMAGE_CASE(9, 0x115u)
#undef IMAGE_CASE
    return false;
}

// Fused encoder head on the tile-resident kernel: 1x1 conv 128 -> 2 L, L = 32 * (4 or 8).
static bool image_zf_applicable(const GemmPlan& P, int L) {
    ImageConvArgs probe;
    if (P.nphase != 1 || P.CgS != 128 || !image_tiles(P, probe) || P.Hout != P.Hin || P.Wout != P.Win || P.N != 2 * L) return false;
    if ((long)P.Hin * P.Win * L >= (1l << 32)) return false;      // the noise element index inside an image is 32 bits
    const Phase& ph = P.ph[0];
    if (ph.ntaps != 1 || ph.dy[0] || ph.dx[0] || ph.SY != 1 || ph.SX != 1 || ph.DSY != 1 || ph.DSX != 1 || ph.D0Y || ph.D0X) return false;
    return L == 128 || L == 256;
}

static bool try_launch_image_zf(const GemmPlan& P, const void* in, const void* wp, const float* bias, void* z, int L,
                                const PhiloxKey& key, hipStream_t st) {
    if (!image_zf_applicable(P, L)) return false;
    ImageConvArgs a;
    memset(&a, 0, sizeof(a));
    a.in = in; a.wp = wp; a.out = z; a.shift = bias; a.relu = 0;
    a.B = P.B; a.Hout = P.Hout; a.Wout = P.Wout; a.N = L; a.NS = L; a.Ktot = P.Ktot;
    image_tiles(P, a);
    a.nslab = L / 32; a.DS = 1;
    for (int q = 0; q < 16; ++q) a.slab_k0[q] = P.ph[0].koff + 32 * q * P.Ktot;   // slab q: weight rows 32 q .. of mu and of log_var
    a.zL = L; a.zkey = key;
    dim3 grid = image_grid(a), block(512);
#define IMAGE_ZF_CASE(NS_)                                                                                              \
    if (a.nslab == NS_) {                                                                                               \
        constexpr int lds = image_zf_lds_bytes<128, 64, NS_>();                                                         \
        static_assert(lds <= 160 * 1024, "LDS budget");                                                                 \
        static LdsOptIn lds_opt;                                                                                        \
        lds_opt.ensure(reinterpret_cast<const void*>(&conv_image_bf16_kernel<128, 64, NS_, (1u << NS_) - 1u, true>), lds); \
        hipLaunchKernelGGL((conv_image_bf16_kernel<128, 64, NS_, (1u << NS_) - 1u, true>), grid, block, lds, st, a);    \
        return true;                                                                                                    \
    }
    IMAGE_ZF_CASE(8) IMAGE_ZF_CASE(4)
#undef IMAGE_ZF_CASE
    return false;
}

}  // namespace dvae
