// ConvTranspose2d k4 s2 p1, 128 -> 64 channels (the decoders' first layer), on maps whose width is a multiple of 16 and
// whose height is even (16x16 at 32x32 images, 32x32 at 64x64, 112x112 at 224x224), weights in registers.
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
// -- a tile's output is four output rows x 32 pixels: four 4 KB runs, 16 KB contiguous on a 16-wide map -- and the staging
// of the tiles ahead: each of a thread's five halo vectors is stored for tile + 1 and at once re-loaded for tile + 2
// (through a buffer descriptor: lanes outside the map read zeros).  No memory operation of the tile loop is conditional.  One barrier per tile, then the drain of the previous patch.
// Persistent over a contiguous run of tiles (an image's tiles row by row, images in order).
#pragma once
#include <type_traits>

#include "conv_wreg.hpp"

namespace dvae {

struct UpArgs {
    const void* in;          // bf16 [B][H][W][128]
    const void* wp;          // packed forward weights [64][Ktot], K = (phase, tap, channel)
    void* out;               // bf16 [B][2 H][2 W][64]
    const float* scale;      // per output channel, or null (1)
    const float* shift;      // per output channel, or null (0)
    int B, H, W, Ktot, relu;
    int tiles_per_wg;
    TileGrid g;              // tiles of 2 x 16 input positions: W / 16 across, H / 2 down (its divisions are branch-free)
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
    const int ntiles = a.B * a.g.tpi;
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

    // ---- staging: vector v = tid + 256 k of the 4 x 18 halo pixels (16 chunks each): pixel (tid >> 4) + 16 k, chunk tid & 15.
    // A vector's address is descriptor base + hoff[k] (per lane, fixed for the thread's life) + the tile's origin (uniform).
    // The descriptor's base lies one map row and one pixel in front of the tensor, so that hoff[k] >= 0; its range check
    // looks at the per-lane part only, so a lane that must not read (halo pixel outside the map, vector beyond the halo)
    // sets bit 31 of it and gets zeros: the zero padding costs two instructions per vector and nothing at the LDS store.
    constexpr int HV = 5;
    const int chunk = tid & 15;
    const unsigned hbias = (unsigned)((a.W + 1) * 256);
    unsigned hoff[HV], hst[HV];
    unsigned m_left = 0, m_right = 0, m_top = 0, m_bot = 0;  // bit k: vector k lies in that border of the halo
#pragma unroll
    for (int k = 0; k < HV; ++k) {
        const int p = (tid >> 4) + 16 * k;
        const int hr = p / 18, hc = p - hr * 18;
        if (p < 72) {
            hoff[k] = (unsigned)((hr * a.W + hc) * 256 + chunk * 16);
            hst[k] = (unsigned)(hr * UP_ROW + hc * UP_PIX + chunk * 16);
            if (hc == 0) m_left |= 1u << k;
            if (hc == 17) m_right |= 1u << k;
            if (hr == 0) m_top |= 1u << k;
            if (hr == 3) m_bot |= 1u << k;
        } else {
            hoff[k] = 0x80000000u;
            hst[k] = 18 * UP_PIX;                          // a row's padding: written, never read
        }
    }
    const auto in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(in)) - hbias, 0, 0x80000000u, 0x00020000);
    u32x4 hreg[HV];
    // origin (byte offset of input pixel (2 ty, 16 tx) of its image) and border mask of a tile; both uniform, branch-free
    auto tile_org = [&](int tile, unsigned& org, unsigned& bad) {
        int b, ty, tx;
        a.g.split(tile, b, ty, tx);
        org = (unsigned)(((b * a.H + 2 * ty) * a.W + 16 * tx) * 256);
        bad = (m_left & (0u - (unsigned)(tx == 0))) | (m_right & (0u - (unsigned)(tx == a.g.ntx - 1))) |
              (m_top & (0u - (unsigned)(ty == 0))) | (m_bot & (0u - (unsigned)(2 * ty + 2 == a.H)));
    };
    auto load_vec = [&](unsigned org, unsigned bad, int k) {
        const unsigned off = (__builtin_amdgcn_ubfe(bad, k, 1) << 31) | hoff[k];
        hreg[k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, off, org, 0));
    };
    auto store_vec = [&](int buf, int k) {
        *reinterpret_cast<u32x4*>(smem + buf * UP_HALO_BYTES + hst[k]) = hreg[k];
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
    const unsigned orow_bytes = (unsigned)(2 * a.W) * 128u;      // one output row
    auto drain = [&](int tile, int pbuf) {                 // the patch as whole pixels: 32-pixel runs of four output rows
        int b, ty, tx;
        a.g.split(tile, b, ty, tx);
        unsigned char* ob = reinterpret_cast<unsigned char*>(a.out) + ((long)b * 2 * a.H + 4 * ty) * orow_bytes + tx * (32 * 128);
        const unsigned char* ps = patch + pbuf * UP_PATCH_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {                      // (row i of the patch: 256 vectors = 32 pixels x 8 chunks)
            const int pix = tid >> 3, c = tid & 7;
            const unsigned char* s = ps + i * UP_OROW + pix * UP_OPIX + (pix >> 4) * 8 + c * 16;
            const u32x2 lo = *reinterpret_cast<const u32x2*>(s), hi = *reinterpret_cast<const u32x2*>(s + 8);
            *reinterpret_cast<u32x4*>(ob + i * orow_bytes + tid * 16) = u32x4{lo[0], lo[1], hi[0], hi[1]};
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
        unsigned org2, bad2;
        tile_org(tile2, org2, bad2);
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
                load_vec(org2, bad2, st / 6);
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
    if (P.Hin < 2 || P.Hin % 2 || P.Win < 16 || P.Win % 16 || P.Hout != 2 * P.Hin || P.Wout != 2 * P.Win || P.Ktot != 2048) return false;
    if (((long)P.B * P.Hin + 2) * P.Win * 256 >= 0x7fffffffL) return false;   // byte offsets into the input are 32-bit
    for (int i = 0; i < 4; ++i) {
        const Phase& ph = P.ph[i];
        if (ph.ntaps != 4 || ph.SY != 1 || ph.SX != 1 || ph.DSY != 2 || ph.DSX != 2 || ph.D0Y != (i >> 1) || ph.D0X != (i & 1) ||
            ph.Hg != P.Hin || ph.Wg != P.Win || ph.koff != 512 * i)
            return false;
        for (int t = 0; t < 4; ++t)
            if (ph.dy[t] < -1 || ph.dy[t] > 1 || ph.dx[t] < -1 || ph.dx[t] > 1) return false;
    }
    TileGrid tg;
    return make_tile_grid(P.Win / 16, P.Hin / 2, ((long)P.B + 1) * (P.Win / 16) * (P.Hin / 2), tg);
}

static bool try_launch_up_bf16(const GemmPlan& P, const void* in, const void* wp, void* out, const float* scale,
                               const float* shift, const void* res, int relu, int accumulate, int out_f32, hipStream_t st) {
    if (res || accumulate || out_f32 || !up_applicable(P)) return false;
    UpArgs a;
    a.in = in; a.wp = wp; a.out = out; a.scale = scale; a.shift = shift;
    a.B = P.B; a.H = P.Hin; a.W = P.Win; a.Ktot = P.Ktot; a.relu = relu;
    make_tile_grid(P.Win / 16, P.Hin / 2, 0, a.g);
    for (int i = 0; i < 4; ++i) {
        a.koff[i] = P.ph[i].koff;
        for (int t = 0; t < 4; ++t) { a.dy[i][t] = P.ph[i].dy[t]; a.dx[i][t] = P.ph[i].dx[t]; }
    }
    const int ntiles = P.B * a.g.tpi;
    a.tiles_per_wg = (ntiles + 255) / 256;                 // one workgroup per CU (a wave per SIMD with all its registers)
    const int gx = (ntiles + a.tiles_per_wg - 1) / a.tiles_per_wg;
    static LdsOptIn lds_opt;
    lds_opt.ensure(reinterpret_cast<const void*>(&conv_up_bf16_kernel), UP_LDS_BYTES);
    hipLaunchKernelGGL(conv_up_bf16_kernel, dim3(gx), dim3(256), UP_LDS_BYTES, st, a);
    return true;
}

}  // namespace dvae
