// 3x3 unit-stride conv 128 -> 128 channels (layer2's second conv and its input gradient; the 128-channel blocks of the
// classifier trunk), on maps whose width is a multiple of 16 and whose height is a multiple of 4, weights in registers.
//
// A workgroup is four waves, one per SIMD; wave w produces output channels 32 w .. 32 w + 31 and keeps their whole weight
// slab -- 9 taps x 128 input channels = 72 A fragments = 288 registers -- for its life: 64 fragments in the accumulator file
// (inline-assembly MFMAs with an "a" operand, as conv_wreg.hpp), the last tap's eight in architectural VGPRs.  A tile is
// 4 rows x 16 columns of output; a lane owns column ox of rows (orow * 2 + u), u = 0, 1: two accumulator sets ("blocks").
// Block 1's pixel lies one row under block 0's, so the pixel fragment block 1 needs for vertical tap dy is the one block 0
// needs for dy + 1: per (horizontal tap, 16-channel slab) FOUR fragment reads serve SIX MFMAs (0.67 KB of LDS per MFMA;
// the image-resident kernel of conv_image.hpp streams the weights through LDS as well and reads 1.5 KB).  Each
// accumulator still sees its taps in the plan's order when that order is row-major (the forward conv): then the sums are
// those of the image-resident kernel bit for bit.
//
// Between the MFMAs of a tile (fully unrolled, scheduling barriers keep the placement) a wave runs the PREVIOUS tile's
// epilogue (two pairs of accumulator sets alternate): scale / shift / residual / ReLU into an LDS patch of whole 256-byte
// pixels.  The residual arrives through the same patch: a thread fetches the 16-byte pieces it will later drain, coalesced,
// and parks them in the patch a tile ahead; the epilogue reads its four channels from there, adds in fp32 and writes the
// result back in place.  Each of a thread's seven halo vectors is stored for tile + 1 and at once re-loaded for tile + 2
// (through a buffer descriptor: lanes outside the map read zeros).  No memory operation of the tile loop is conditional.
// One barrier per tile, then the drain of the previous patch.  Persistent over a contiguous run of tiles.
#pragma once
#include <type_traits>

#include "conv_wreg.hpp"

namespace dvae {

struct MidArgs {
    const void* in;          // bf16 [B][H][W][128]
    const void* wp;          // packed weights [128][Ktot], K = (tap, channel)
    void* out;               // bf16 [B][H][W][128]
    const void* res;         // residual, laid out like out, or null (only read by the RES instantiation)
    const float* scale;      // per output channel, or null (1)
    const float* shift;      // per output channel, or null (0)
    int B, H, W, Ktot, relu;
    int tiles_per_wg;
    TileGrid g;              // tiles of 4 x 16 pixels: W / 16 across, H / 4 down
    int kcol[9];             // first K column of the tap at (dy, dx) = (c / 3 - 1, c % 3 - 1)
};

static constexpr int MID_PIX = 256 + 16;                       // bytes per staged input pixel
static constexpr int MID_ROW = 5120;                           // bytes per staged row (18 pixels, padded to a multiple of 256)
static constexpr int MID_HALO_BYTES = 6 * MID_ROW;
static constexpr int MID_OPIX = 256 + 8;                       // bytes per pixel of the output patch
static constexpr int MID_PATCH_BYTES = 64 * MID_OPIX + 128 + 128;   // 128-byte gap after pixel 31 (bank spread)
static constexpr int MID_WROW = 256 + 16;                      // bytes per row of a wave's weight window (one tap: 128 columns)
static constexpr int MID_TILE_LDS = 2 * MID_HALO_BYTES + 2 * MID_PATCH_BYTES + 2 * 128 * 4;
static constexpr int MID_LDS_BYTES = 4 * 32 * MID_WROW > MID_TILE_LDS ? 4 * 32 * MID_WROW : MID_TILE_LDS;

template <bool RES>
__global__ __launch_bounds__(256) void conv_mid_bf16_kernel(const MidArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* patch = smem + 2 * MID_HALO_BYTES;
    float* ssl = reinterpret_cast<float*>(smem + 2 * MID_HALO_BYTES + 2 * MID_PATCH_BYTES);    // scale[128], shift[128]
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, r = lane & 31, h = lane >> 5, orow = r >> 4, ox = r & 15;
    const bf16* wp = reinterpret_cast<const bf16*>(a.wp);
    const int ntiles = a.B * a.g.tpi;
    const int t0 = blockIdx.x * a.tiles_per_wg;
    const int t1 = t0 + a.tiles_per_wg < ntiles ? t0 + a.tiles_per_wg : ntiles;
    if (t0 >= t1) return;

    // ---- weights: a wave moves its own 32 rows through a private LDS window, a tap at a time (32 rows x 128 columns,
    // copied as contiguous 256-byte row pieces), and takes its fragments from there: lane (r, h) supplies row 32 w + r,
    // K columns 16 s + 8 h .. of the tap.  Straight-line and unconditional, so that the fragments are plain definitions
    // the "a" operands can pin.  Fragment index = 8 c + s, c = 3 (dy + 1) + (dx + 1).
    u32x4 wreg[72];                                        // [0, 64): accumulator file; [64, 72): architectural VGPRs
    {
        unsigned char* win = smem + wave * (32 * MID_WROW);
        const unsigned char* src = reinterpret_cast<const unsigned char*>(wp) + (long)(32 * wave) * a.Ktot * 2;   // uniform
        const int wrow = lane >> 4, wcol = lane & 15;
#pragma unroll
        for (int c = 0; c < 9; ++c) {
            u32x4 wt[8];
#pragma unroll
            for (int i = 0; i < 8; ++i)
                wt[i] = *reinterpret_cast<const u32x4*>(src + (unsigned)(((4 * i + wrow) * a.Ktot + a.kcol[c]) * 2 + wcol * 16));
#pragma unroll
            for (int i = 0; i < 8; ++i)
                *reinterpret_cast<u32x4*>(win + (4 * i + wrow) * MID_WROW + wcol * 16) = wt[i];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int s = 0; s < 8; ++s)
                wreg[8 * c + s] = *reinterpret_cast<const u32x4*>(win + r * MID_WROW + 32 * s + 16 * h);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
    __syncthreads();                                       // the windows are dead: the halo buffers and patches take their place
    if (tid < 128) {
        ssl[tid] = a.scale ? a.scale[tid] : 1.f;
        ssl[128 + tid] = a.shift ? a.shift[tid] : 0.f;
    }

    // ---- staging: vector v = tid + 256 k of the 6 x 18 halo pixels (16 chunks each): pixel (tid >> 4) + 16 k, chunk tid & 15.
    // A vector's address is descriptor base + hoff[k] (per lane, fixed for the thread's life) + the tile's origin (uniform).
    // The descriptor's base lies one map row and one pixel in front of the tensor, so that hoff[k] >= 0; its range check
    // looks at the per-lane part only, so a lane that must not read (halo pixel outside the map, vector beyond the halo)
    // sets bit 31 of it and gets zeros.
    constexpr int HV = 7;
    const int chunk = tid & 15;
    const unsigned hbias = (unsigned)((a.W + 1) * 256);
    unsigned hoff[HV], hst[HV];
    unsigned m_left = 0, m_right = 0, m_top = 0, m_bot = 0;  // bit k: vector k lies in that border of the halo
#pragma unroll
    for (int k = 0; k < HV; ++k) {
        const int p = (tid >> 4) + 16 * k;
        const int hr = p / 18, hc = p - hr * 18;
        if (p < 108) {
            hoff[k] = (unsigned)((hr * a.W + hc) * 256 + chunk * 16);
            hst[k] = (unsigned)(hr * MID_ROW + hc * MID_PIX + chunk * 16);
            if (hc == 0) m_left |= 1u << k;
            if (hc == 17) m_right |= 1u << k;
            if (hr == 0) m_top |= 1u << k;
            if (hr == 5) m_bot |= 1u << k;
        } else {
            hoff[k] = 0x80000000u;
            hst[k] = 18 * MID_PIX;                         // a row's padding: written, never read
        }
    }
    const auto in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(a.in)) - hbias, 0, 0x80000000u, 0x00020000);
    u32x4 hreg[HV];
    // origin (byte offset of pixel (4 ty, 16 tx) of its image) and border mask of a tile; both uniform, branch-free
    auto tile_org = [&](int tile, unsigned& org, unsigned& bad) {
        int b, ty, tx;
        a.g.split(tile, b, ty, tx);
        org = (unsigned)(((b * a.H + 4 * ty) * a.W + 16 * tx) * 256);
        bad = (m_left & (0u - (unsigned)(tx == 0))) | (m_right & (0u - (unsigned)(tx == a.g.ntx - 1))) |
              (m_top & (0u - (unsigned)(ty == 0))) | (m_bot & (0u - (unsigned)(4 * ty + 4 == a.H)));
    };
    auto load_vec = [&](unsigned org, unsigned bad, int k) {
        const unsigned off = (__builtin_amdgcn_ubfe(bad, k, 1) << 31) | hoff[k];
        hreg[k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, off, org, 0));
    };
    auto store_vec = [&](int buf, int k) {
        *reinterpret_cast<u32x4*>(smem + buf * MID_HALO_BYTES + hst[k]) = hreg[k];
    };

    // ---- pixel fragment (f, dx, s): chunk 2 s + h of halo pixel (f + 2 orow, ox + dx + 1); halo pixel (0, 0) is (-1, -1).
    // Block u (pixel row u + 2 orow) uses it for vertical tap dy = f - u - 1.
    const unsigned char* fb = smem + 2 * orow * MID_ROW + ox * MID_PIX + 16 * h;
    auto frag = [&](int cur, int i) -> u32x4 {             // flattened step i = (f, dx + 1, s)
        const int f = i / 24, dxi = (i % 24) / 8, s = i % 8;
        return *reinterpret_cast<const u32x4*>(fb + cur * MID_HALO_BYTES + f * MID_ROW + dxi * MID_PIX + 32 * s);
    };

    // ---- the patch: pixel p = 16 * (row of the tile) + column at p * 264 + (p >> 5) * 128
    // epilogue group e = (u, g): channels 32 w + 8 g + 4 h .. of pixel (u + 2 orow, ox)
    const int poff = (16 * 2 * orow + ox) * MID_OPIX + orow * 128 + (32 * wave + 4 * h) * 2;
    const short rl = (short)(a.relu ? 0 : -32768);
    typedef short i16x2_t __attribute__((ext_vector_type(2)));
    typedef bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    const i16x2_t relu_lo = {rl, rl};
    f32x4 esc, esh;
    bf16x4 erq;
    auto epi_load = [&](int pbuf, int e) {
        const int u = e >> 2, g = e & 3;
        const int nl = 32 * wave + 8 * g + 4 * h;
        esc = *reinterpret_cast<const f32x4*>(ssl + nl);
        esh = *reinterpret_cast<const f32x4*>(ssl + 128 + nl);
        if (RES) erq = *reinterpret_cast<const bf16x4*>(patch + pbuf * MID_PATCH_BYTES + poff + u * (16 * MID_OPIX) + 16 * g);
    };
    auto epi_store = [&](const f32x16 (&acc)[2], int pbuf, int e) {
        const int u = e >> 2, g = e & 3;
        unsigned o[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            f32x2_t f = {acc[u][4 * g + 2 * k] * esc[2 * k] + esh[2 * k], acc[u][4 * g + 2 * k + 1] * esc[2 * k + 1] + esh[2 * k + 1]};
            if (RES) { f[0] += (float)erq[2 * k]; f[1] += (float)erq[2 * k + 1]; }
            const bf16x2_t q = __builtin_convertvector(f, bf16x2_t);
            // ReLU on the rounded pair: a bf16 read as a signed 16-bit integer is negative exactly when the number is
            o[k] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(i16x2_t, q), relu_lo));
        }
        *reinterpret_cast<u32x2*>(patch + pbuf * MID_PATCH_BYTES + poff + u * (16 * MID_OPIX) + 16 * g) = u32x2{o[0], o[1]};
    };
    // the coalesced side: row i of the tile is 256 vectors = 16 pixels x 16 chunks, one per thread
    const unsigned orow_bytes = (unsigned)a.W * 256u;      // one map row
    auto patch_at = [&](int pbuf, int i) -> unsigned char* {
        return patch + pbuf * MID_PATCH_BYTES + (16 * i + (tid >> 4)) * MID_OPIX + (i >> 1) * 128 + chunk * 16;
    };
    auto drain = [&](int tile, int pbuf) {
        unsigned org, bad;
        tile_org(tile, org, bad);
        unsigned char* ob = reinterpret_cast<unsigned char*>(a.out) + org;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned char* s = patch_at(pbuf, i);
            const u32x2 lo = *reinterpret_cast<const u32x2*>(s), hi = *reinterpret_cast<const u32x2*>(s + 8);
            *reinterpret_cast<u32x4*>(ob + i * orow_bytes + tid * 16) = u32x4{lo[0], lo[1], hi[0], hi[1]};
        }
    };
    u32x4 rreg[4];
    auto res_load = [&](unsigned org, int i) {
        rreg[i] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(a.res) + org + i * orow_bytes + tid * 16);
    };
    auto res_park = [&](int pbuf, int i) {
        unsigned char* s = patch_at(pbuf, i);
        *reinterpret_cast<u32x2*>(s) = u32x2{rreg[i][0], rreg[i][1]};
        *reinterpret_cast<u32x2*>(s + 8) = u32x2{rreg[i][2], rreg[i][3]};
    };

    // One tile: 144 MFMAs into accN over 96 fragment steps; between them the epilogue of accO (tile - 1), this tile's own
    // residual rows (fetched early, parked late, into the patch this tile's epilogue will use) and, for each halo vector,
    // its store (tile + 1) and reload (tile + 2).  Nothing is conditional: the first tile's "previous" epilogue works on
    // an undefined accumulator and is drained to the first tile's own place, which the real drain overwrites one barrier
    // later (same thread, same addresses); the last tiles re-load the last tile and store a halo nobody reads.
    // (cur = parity of the tile's position in the run, a compile-time constant of each of the loop body's two calls)
    auto phase = [&](auto CUR, f32x16 (&accN)[2], const f32x16 (&accO)[2], int tile, bool first) {
        constexpr int cur = decltype(CUR)::value;
        const int tile2 = tile + 2 < t1 ? tile + 2 : t1 - 1;
        unsigned org2, bad2, org0, bad0;
        tile_org(tile2, org2, bad2);
        tile_org(tile, org0, bad0);
        constexpr int PF = 2;                              // fragments are read PF steps ahead of their MFMAs
        u32x4 pv[PF + 1];
#pragma unroll
        for (int i = 0; i < PF; ++i) pv[i] = frag(cur, i);
#pragma unroll
        for (int i = 0; i < 96; ++i) {
            const int f = i / 24, dxi = (i % 24) / 8, s = i % 8;
            if (i + PF < 96) pv[(i + PF) % (PF + 1)] = frag(cur, i + PF);
            const u32x4& x = pv[i % (PF + 1)];
            if (f <= 2) {                                  // block 0: vertical tap f
                const int w = 8 * (3 * f + dxi) + s;
                if (i == 0) mfma_first_a(accN[0], wreg[w < 64 ? w : 0], x);
                else if (w < 64) mfma_acc_a(accN[0], wreg[w < 64 ? w : 0], x);
                else mfma_acc_v(accN[0], wreg[w], x);
            }
            if (f >= 1) {                                  // block 1: vertical tap f - 1
                const int w = 8 * (3 * (f - 1) + dxi) + s;
                if (i == 24) mfma_first_a(accN[1], wreg[w < 64 ? w : 0], x);
                else if (w < 64) mfma_acc_a(accN[1], wreg[w < 64 ? w : 0], x);
                else mfma_acc_v(accN[1], wreg[w], x);
            }
            // side work of this step
            if (i % 12 == 1) epi_load(cur ^ 1, i / 12);
            if (i % 12 == 7) epi_store(accO, cur ^ 1, i / 12);
            if (i % 13 == 2 && i / 13 < HV) {
                store_vec(cur ^ 1, i / 13);
                load_vec(org2, bad2, i / 13);
            }
            if (RES) {
                if (i % 12 == 4 && i / 12 < 4) res_load(org0, i / 12);
                if (i % 10 == 0 && i >= 50 && i < 90) res_park(cur, (i - 50) / 10);
            }
            __builtin_amdgcn_sched_barrier(0);             // keep reads and side work where they were placed
        }
        __syncthreads();                                   // patch and next halo complete; everyone is done with this halo
        drain(first ? tile : tile - 1, cur ^ 1);
    };

    // first tile's halo, and the registers of the second
    {
        unsigned org, bad;
        tile_org(t0, org, bad);
#pragma unroll
        for (int k = 0; k < HV; ++k) load_vec(org, bad, k);
        tile_org(t0 + 1 < t1 ? t0 + 1 : t0, org, bad);
#pragma unroll
        for (int k = 0; k < HV; ++k) {
            store_vec(0, k);
            load_vec(org, bad, k);
        }
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
    // the last tile's epilogue and drain (patch of iteration it - 1: its residual rows are parked there)
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");   // the last MFMAs' results -> the VALU reads below
    {
        const int pbuf = (it - 1) & 1;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            epi_load(pbuf, e);
            if (last_a) epi_store(accA, pbuf, e); else epi_store(accB, pbuf, e);
        }
        __syncthreads();
        drain(t1 - 1, pbuf);
    }
}

// true if the plan is a launch of the kernel above; kcol: the K column of each (dy, dx)
static bool mid_applicable(const GemmPlan& P, int (&kcol)[9]) {
    if (P.nphase != 1 || P.CgS != 128 || P.Cg != 128 || P.N != 128 || P.NS != 128 || P.Ktot != 1152) return false;
    if (P.Hin < 4 || P.Hin % 4 || P.Win < 16 || P.Win % 16 || P.Hout != P.Hin || P.Wout != P.Win) return false;
    if (((long)P.B * P.Hin + 2) * P.Win * 256 >= 0x7fffffffL) return false;   // byte offsets into the tensors are 32-bit
    const Phase& ph = P.ph[0];
    if (ph.ntaps != 9 || ph.SY != 1 || ph.SX != 1 || ph.DSY != 1 || ph.DSX != 1 || ph.D0Y || ph.D0X || ph.Hg != P.Hin ||
        ph.Wg != P.Win)
        return false;
    for (int c = 0; c < 9; ++c) kcol[c] = -1;
    for (int t = 0; t < 9; ++t) {
        if (ph.dy[t] < -1 || ph.dy[t] > 1 || ph.dx[t] < -1 || ph.dx[t] > 1) return false;
        kcol[3 * (ph.dy[t] + 1) + ph.dx[t] + 1] = ph.koff + 128 * t;
    }
    for (int c = 0; c < 9; ++c)
        if (kcol[c] < 0) return false;
    TileGrid tg;
    return make_tile_grid(P.Win / 16, P.Hin / 4, ((long)P.B + 1) * (P.Win / 16) * (P.Hin / 4), tg);
}

static bool try_launch_mid_bf16(const GemmPlan& P, const void* in, const void* wp, void* out, const float* scale,
                                const float* shift, const void* res, int relu, int accumulate, int out_f32, hipStream_t st) {
    MidArgs a;
    if (accumulate || out_f32 || !mid_applicable(P, a.kcol)) return false;
    a.in = in; a.wp = wp; a.out = out; a.res = res; a.scale = scale; a.shift = shift;
    a.B = P.B; a.H = P.Hin; a.W = P.Win; a.Ktot = P.Ktot; a.relu = relu;
    make_tile_grid(P.Win / 16, P.Hin / 4, 0, a.g);
    const int ntiles = P.B * a.g.tpi;
    a.tiles_per_wg = (ntiles + 255) / 256;                 // one workgroup per CU (a wave per SIMD with all its registers)
    const int gx = (ntiles + a.tiles_per_wg - 1) / a.tiles_per_wg;
    static LdsOptIn lds_res, lds_plain;
    if (res) {
        lds_res.ensure(reinterpret_cast<const void*>(&conv_mid_bf16_kernel<true>), MID_LDS_BYTES);
        hipLaunchKernelGGL(conv_mid_bf16_kernel<true>, dim3(gx), dim3(256), MID_LDS_BYTES, st, a);
    } else {
        lds_plain.ensure(reinterpret_cast<const void*>(&conv_mid_bf16_kernel<false>), MID_LDS_BYTES);
        hipLaunchKernelGGL(conv_mid_bf16_kernel<false>, dim3(gx), dim3(256), MID_LDS_BYTES, st, a);
    }
    return true;
}

}  // namespace dvae
