// 3x3 stride-2 bf16 convolution, 64 -> 128 channels (the first conv of a down-sampling residual block), weights in
// registers.
//
// A workgroup is four waves, one per SIMD, and owns tiles of 4 x 16 output pixels.  Wave (c, q) holds the weights of
// output channels [64 c, 64 c + 64) for all nine taps -- 36 K steps x 2 blocks of 32 channels = 72 MFMA A-operand
// fragments = 288 registers: 64 fragments in the accumulator file (inline-assembly MFMAs with an "a" operand, as
// conv_wreg.hpp), the last tap's eight in architectural VGPRs -- and computes them for output rows 2 q, 2 q + 1 of the tile:
// 36 K steps of two MFMAs that share one 1 KB pixel fragment (0.5 KB of LDS per MFMA; with a wave per 32 channels, the
// first form of this kernel, every MFMA read its own fragment, and at the full MFMA rate those reads alone are the LDS
// pipe: it ran at 43 % matrix-pipe and 43 % LDS occupancy).
//
// The 9 x 33 input pixels under a tile are staged in LDS once (zero outside the image: the loads go through a buffer
// descriptor whose range check returns zeros for the lanes that set bit 31 of their offset) and every tap re-addresses
// that copy: lane (output row o, column x) reads input pixel (2 o + ky, 2 x + kx).  Consecutive lanes are two pixels =
// 256 B apart, one whole bank row, so the image is swizzled twice: the eight 16-byte chunks of a pixel are XORed with
// (column >> 1) & 7, and rows 2-3 (mod 4) start one pixel late, which moves the second output row of a lane group to the
// other half of the bank row: the 16 lanes a ds_read_b128 serves together then hit 16 different slots.
//
// A wave's side work rides between its MFMAs, a piece every few steps of the fully unrolled 36-step loop (placed by hand,
// held in place by scheduling barriers): the PREVIOUS tile's epilogue (two pairs of accumulator sets alternate; scale /
// shift / ReLU, rounding, into an LDS patch so that a tile leaves as whole 256-byte pixels: four contiguous 4 KB rows),
// and the staging of the tiles ahead -- each of a thread's ten halo vectors is written to the other halo buffer (next
// tile) and at once re-loaded from memory for the tile after, so a load has a whole tile period to land.  Nothing in the
// tile loop is conditional and nothing in it branches.  One barrier per tile, followed by the drain of the patch.
// Workgroups are persistent over a contiguous run of tiles (whole images in order, so the input row two tiles share is
// found in L2).
#pragma once
#include <type_traits>

#include "conv_wreg.hpp"

namespace dvae {

struct DownArgs {
    const void* in;          // bf16 [B][Hin][Win][64]
    const void* wp;          // packed forward weights [128][Ktot], K = (tap, channel)
    void* out;               // bf16 [B][Hout][Wout][128]
    const float* scale;      // per output channel, or null (1)
    const float* shift;      // per output channel, or null (0)
    int B, Hin, Win, Hout, Wout, Ktot, relu;
    int tiles_per_wg;
    TileGrid g;              // tiles of an image: Wout / 16 per tile row, Hout / 4 tile rows
};

static constexpr int DOWN_RP = 34;                              // pixels per staged row (33 + the one-pixel shift)
static constexpr int DOWN_HALO_BYTES = 9 * DOWN_RP * 128;
static constexpr int DOWN_OPIX = 256 + 8;                       // bytes per pixel of the output patch
static constexpr int DOWN_PATCH_BYTES = 64 * DOWN_OPIX + 128 + 128;   // a 128-byte gap after pixel 31 (bank spread)
static constexpr int DOWN_WROW = 512 + 16;                      // bytes per row of a wave's weight window (256 columns)
static constexpr int DOWN_TILE_LDS = 2 * DOWN_HALO_BYTES + 2 * DOWN_PATCH_BYTES + 2 * 128 * 4;
static constexpr int DOWN_LDS_BYTES = 4 * 32 * DOWN_WROW > DOWN_TILE_LDS ? 4 * 32 * DOWN_WROW : DOWN_TILE_LDS;

__global__ __launch_bounds__(256) void conv_down_bf16_kernel(const DownArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* patch = smem + 2 * DOWN_HALO_BYTES;
    float* ssl = reinterpret_cast<float*>(smem + 2 * DOWN_HALO_BYTES + 2 * DOWN_PATCH_BYTES);   // scale[128], shift[128]
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, r = lane & 31, h = lane >> 5, orow = r >> 4, ox = r & 15;
    const int c = wave >> 1, q = wave & 1;
    const bf16* wp = reinterpret_cast<const bf16*>(a.wp);
    const int ntiles = a.B * a.g.tpi;
    const int t0 = blockIdx.x * a.tiles_per_wg;
    const int t1 = t0 + a.tiles_per_wg < ntiles ? t0 + a.tiles_per_wg : ntiles;
    if (t0 >= t1) return;

    // ---- weights: a wave moves its own 64 rows through a private LDS window, 32 rows x 256 columns at a time (copied as
    // contiguous row pieces; the last piece of a row is 64 columns), and takes its fragments from there: lane (r, h)
    // supplies row 64 c + 32 j + r, K columns 16 st + 8 h .. .  Straight-line and unconditional, so that the fragments are
    // plain definitions the "a" operands can pin.  Fragment index 2 st + j.
    u32x4 wreg[72];                                        // [0, 64): accumulator file; [64, 72): architectural VGPRs
    {
        unsigned char* win = smem + wave * (32 * DOWN_WROW);
        const unsigned char* src = reinterpret_cast<const unsigned char*>(wp) + (long)(64 * c) * a.Ktot * 2;   // uniform
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int kq = 0; kq < 2; ++kq) {               // columns 256 kq .. 256 kq + 255: 32 rows x 512 bytes
                const int wrow = lane >> 5, wcol = lane & 31;
                u32x4 wt[16];
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    wt[i] = *reinterpret_cast<const u32x4*>(src + (unsigned)((32 * j + 2 * i + wrow) * a.Ktot * 2 + 512 * kq + wcol * 16));
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    *reinterpret_cast<u32x4*>(win + (2 * i + wrow) * DOWN_WROW + wcol * 16) = wt[i];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                for (int s16 = 0; s16 < 16; ++s16)
                    wreg[2 * (16 * kq + s16) + j] = *reinterpret_cast<const u32x4*>(win + r * DOWN_WROW + 32 * s16 + 16 * h);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
            {                                              // columns 512 .. 575: 32 rows x 128 bytes
                const int wrow = lane >> 3, wcol = lane & 7;
                u32x4 wt[4];
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    wt[i] = *reinterpret_cast<const u32x4*>(src + (unsigned)((32 * j + 8 * i + wrow) * a.Ktot * 2 + 1024 + wcol * 16));
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    *reinterpret_cast<u32x4*>(win + (8 * i + wrow) * DOWN_WROW + wcol * 16) = wt[i];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                for (int s16 = 0; s16 < 4; ++s16)
                    wreg[2 * (32 + s16) + j] = *reinterpret_cast<const u32x4*>(win + r * DOWN_WROW + 32 * s16 + 16 * h);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
    }
    __syncthreads();                                       // the windows are dead: the halo buffers and patches take their place
    if (tid < 128) {
        ssl[tid] = a.scale ? a.scale[tid] : 1.f;
        ssl[128 + tid] = a.shift ? a.shift[tid] : 0.f;
    }

    // ---- staging: the 9 x 33 halo pixels (8 chunks each); halo pixel (0, 0) is input pixel (-1, -1) of the tile's origin.
    // Thread (tid >> 3, chunk tid & 7) takes its vector of halo column (tid >> 3) + 1 of each of the nine rows (k = 0 .. 8),
    // and the first 72 threads also take column 0 of row tid >> 3 (k = 9).  A vector's address is descriptor base +
    // hoff[k] (per lane) + the tile's origin (uniform); the base lies one image row and one pixel in front of the tensor,
    // so that hoff[k] >= 0; a lane that must not read (row -1, column -1, no tenth vector) sets bit 31 of its offset and
    // gets zeros.  (The far edges are inside the image: Hin = 2 Hout, Win = 2 Wout.)
    constexpr int HV = 10;
    const int chunk = tid & 7, scol = (tid >> 3) + 1;
    const unsigned hbias = (unsigned)((a.Win + 1) * 128);
    auto lds_at = [&](int row, int col) -> unsigned {
        return (unsigned)((row * DOWN_RP + ((row >> 1) & 1) + col) * 128 + ((chunk ^ ((col >> 1) & 7)) * 16));
    };
    unsigned hoff9, hst9;                                  // the tenth vector
    {
        const int erow = tid >> 3;
        hoff9 = tid < 72 ? (unsigned)(erow * a.Win * 128 + chunk * 16) : 0x80000000u;
        hst9 = tid < 72 ? lds_at(erow, 0) : (unsigned)(33 * 128);   // (row 0's spare pixel: written, never read)
    }
    const unsigned hoff0 = (unsigned)(scol * 128 + chunk * 16);      // + k * Win * 128
    const unsigned hst0 = lds_at(0, scol);                             // + (k * RP + ((k >> 1) & 1)) * 128
    const unsigned rowb = (unsigned)a.Win * 128u;
    const auto in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(a.in)) - hbias, 0, 0x80000000u, 0x00020000);
    u32x4 hreg[HV];
    // origin (byte offset of input pixel (8 ty, 32 tx) of its image) and the two border flags of a tile, as bit 31 or 0
    auto tile_org = [&](int tile, unsigned& org, unsigned& top, unsigned& left) {
        int b, ty, tx;
        a.g.split(tile, b, ty, tx);
        org = (unsigned)(((b * a.Hin + 8 * ty) * a.Win + 32 * tx) * 128);
        top = (unsigned)(ty == 0) << 31;
        left = (unsigned)(tx == 0) << 31;
    };
    auto load_vec = [&](unsigned org, unsigned top, unsigned left, int k) {
        const unsigned off = k == 0 ? (hoff0 | top) : k < 9 ? hoff0 + k * rowb : (hoff9 | left | (tid < 8 ? top : 0u));
        hreg[k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, off, org, 0));
    };
    auto store_vec = [&](int buf, int k) {
        const unsigned at = k < 9 ? hst0 + (unsigned)((k * DOWN_RP + ((k >> 1) & 1)) * 128) : hst9;
        *reinterpret_cast<u32x4*>(smem + buf * DOWN_HALO_BYTES + at) = hreg[k];
    };

    // ---- pixel fragment of tap (ky, kx), slab s: chunk 2 s + h of halo pixel (4 q + 2 orow + ky, 2 ox + kx)
    // = base[kx == 2] ^ (ky == 2 ? 128 : 0) ^ (32 s), + (ky * RP + kx) * 128.  The chunk swizzle depends on
    // (ox + (kx >> 1)) & 7: two lane-dependent bases.  The row shift (orow + (ky >> 1)) & 1 is bit 7 of the address (the
    // other terms of the pixel index are even), flipped for ky = 2; the slab's 32 s falls on bits 5-6, where the bases
    // hold the swizzled chunk; everything else is immediate.
    int base[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
        base[j] = ((4 * q + 2 * orow) * DOWN_RP + (orow & 1) + 2 * ox) * 128 + ((h ^ ((ox + j) & 7)) * 16);
    auto frag = [&](int cur, int st) -> u32x4 {
        const int t = st >> 2, ky = t / 3, kx = t % 3;
        int bs = base[kx >> 1];
        asm volatile("" : "+v"(bs));                       // the XOR is made here: hoisted addresses cost the prefetch its registers
        return *reinterpret_cast<const u32x4*>(smem + cur * DOWN_HALO_BYTES + (bs ^ ((st & 3) * 32 + (ky >> 1) * 128)) +
                                               (ky * DOWN_RP + kx) * 128);
    };

    // ---- the patch: pixel p = 16 * (row of the tile) + column at p * 264 + (p >> 5) * 128
    // epilogue group e = (j, g): channels 64 c + 32 j + 8 g + 4 h .. of pixel (2 q + orow, ox)
    const int poff = (32 * q + r) * DOWN_OPIX + q * 128 + (64 * c + 4 * h) * 2;
    const short rl = (short)(a.relu ? 0 : -32768);
    typedef short i16x2_t __attribute__((ext_vector_type(2)));
    typedef bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    const i16x2_t relu_lo = {rl, rl};
    f32x4 esc, esh;
    auto epi_load = [&](int e) {
        const int nl = 64 * c + 32 * (e >> 2) + 8 * (e & 3) + 4 * h;
        esc = *reinterpret_cast<const f32x4*>(ssl + nl);
        esh = *reinterpret_cast<const f32x4*>(ssl + 128 + nl);
    };
    auto epi_store = [&](const f32x16 (&acc)[2], int pbuf, int e) {
        const int j = e >> 2, g = e & 3;
        unsigned o[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const f32x2_t f = {acc[j][4 * g + 2 * k] * esc[2 * k] + esh[2 * k], acc[j][4 * g + 2 * k + 1] * esc[2 * k + 1] + esh[2 * k + 1]};
            const bf16x2_t v = __builtin_convertvector(f, bf16x2_t);
            // ReLU on the rounded pair: a bf16 read as a signed 16-bit integer is negative exactly when the number is
            o[k] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(i16x2_t, v), relu_lo));
        }
        *reinterpret_cast<u32x2*>(patch + pbuf * DOWN_PATCH_BYTES + poff + (32 * j + 8 * g) * 2) = u32x2{o[0], o[1]};
    };
    // the patch as whole pixels: row i of the tile is 256 vectors = 16 pixels x 16 chunks, one per thread
    const unsigned orow_bytes = (unsigned)a.Wout * 256u;
    auto drain = [&](int tile, int pbuf) {
        int b, ty, tx;
        a.g.split(tile, b, ty, tx);
        unsigned char* ob = reinterpret_cast<unsigned char*>(a.out) + (((long)b * a.Hout + 4 * ty) * a.Wout + 16 * tx) * 256;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned char* s = patch + pbuf * DOWN_PATCH_BYTES + (16 * i + (tid >> 4)) * DOWN_OPIX + (i >> 1) * 128 + (tid & 15) * 16;
            const u32x2 lo = *reinterpret_cast<const u32x2*>(s), hi = *reinterpret_cast<const u32x2*>(s + 8);
            *reinterpret_cast<u32x4*>(ob + i * orow_bytes + tid * 16) = u32x4{lo[0], lo[1], hi[0], hi[1]};
        }
    };

    // One tile: 72 MFMAs into accN; between them the epilogue of accO (tile - 1) and, for each halo vector, its store
    // (tile + 1) and reload (tile + 2).  Nothing is conditional.  The first tile has no predecessor: its "previous"
    // epilogue works on an undefined accumulator and is drained to the first tile's own place, which the real drain
    // overwrites one barrier later (same thread, same addresses).  The last tiles re-load the last tile and store a halo
    // nobody reads.
    // (cur = parity of the tile's position in the run, a compile-time constant of each of the loop body's two calls)
    auto phase = [&](auto CUR, f32x16 (&accN)[2], const f32x16 (&accO)[2], int tile, bool first) {
        constexpr int cur = decltype(CUR)::value;
        const int tile2 = tile + 2 < t1 ? tile + 2 : t1 - 1;
        unsigned org2, top2, left2;
        tile_org(tile2, org2, top2, left2);
        constexpr int PF = 2;                              // fragments are read PF steps ahead of their MFMAs
        u32x4 pv[PF + 1];
#pragma unroll
        for (int st = 0; st < PF; ++st) pv[st] = frag(cur, st);
#pragma unroll
        for (int st = 0; st < 36; ++st) {
            if (st + PF < 36) pv[(st + PF) % (PF + 1)] = frag(cur, st + PF);
            const u32x4& x = pv[st % (PF + 1)];
            if (st == 0) {
                mfma_first_a(accN[0], wreg[0], x);
                mfma_first_a(accN[1], wreg[1], x);
            } else if (st < 32) {
                mfma_acc_a(accN[0], wreg[st < 32 ? 2 * st : 0], x);
                mfma_acc_a(accN[1], wreg[st < 32 ? 2 * st + 1 : 1], x);
            } else {
                mfma_acc_v(accN[0], wreg[2 * st], x);
                mfma_acc_v(accN[1], wreg[2 * st + 1], x);
            }
            // side work of this step
            if (st % 4 == 1 && st / 4 < 8) epi_load(st / 4);
            if (st % 4 == 3 && st / 4 < 8) epi_store(accO, cur ^ 1, st / 4);
            if (st % 4 == 0 || st == 34) {
                const int k = st == 34 ? 9 : st / 4;
                store_vec(cur ^ 1, k);
                load_vec(org2, top2, left2, k);
            }
            __builtin_amdgcn_sched_barrier(0);             // keep reads and side work where they were placed
        }
        __syncthreads();                                   // patch and next halo complete; everyone is done with this halo
        drain(first ? tile : tile - 1, cur ^ 1);
    };

    // first tile's halo, and the registers of the second
    {
        unsigned org, top, left;
        tile_org(t0, org, top, left);
#pragma unroll
        for (int k = 0; k < HV; ++k) load_vec(org, top, left, k);
        tile_org(t0 + 1 < t1 ? t0 + 1 : t0, org, top, left);
#pragma unroll
        for (int k = 0; k < HV; ++k) {
            store_vec(0, k);
            load_vec(org, top, left, k);
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
static bool down_applicable(const GemmPlan& P) {
    if (P.nphase != 1 || P.CgS != 64 || P.Cg != 64 || P.N != 128 || P.NS != 128 || P.Ktot != 576) return false;
    const Phase& ph = P.ph[0];
    if (ph.SY != 2 || ph.SX != 2 || ph.DSY != 1 || ph.DSX != 1 || ph.D0Y || ph.D0X || ph.ntaps != 9) return false;
    if (ph.Hg != P.Hout || ph.Wg != P.Wout || P.Hin != 2 * P.Hout || P.Win != 2 * P.Wout) return false;
    if (P.Wout % 16 || P.Hout % 4 || ph.koff != 0) return false;
    for (int t = 0; t < 9; ++t)
        if (ph.dy[t] != t / 3 - 1 || ph.dx[t] != t % 3 - 1) return false;
    if (((long)P.B * P.Hin + 2) * P.Win * 128 >= 0x7fffffffL) return false;    // byte offsets into the input are 32-bit
    TileGrid tg;
    return make_tile_grid(P.Wout / 16, P.Hout / 4, ((long)P.B + 1) * (P.Wout / 16) * (P.Hout / 4), tg);
}

static bool try_launch_down_bf16(const GemmPlan& P, const void* in, const void* wp, void* out, const float* scale,
                                 const float* shift, const void* res, int relu, int accumulate, int out_f32, hipStream_t st) {
    if (res || accumulate || out_f32 || !down_applicable(P)) return false;
    DownArgs a;
    a.in = in; a.wp = wp; a.out = out; a.scale = scale; a.shift = shift;
    a.B = P.B; a.Hin = P.Hin; a.Win = P.Win; a.Hout = P.Hout; a.Wout = P.Wout; a.Ktot = P.Ktot; a.relu = relu;
    make_tile_grid(P.Wout / 16, P.Hout / 4, 0, a.g);
    const int ntiles = P.B * a.g.tpi;
    a.tiles_per_wg = (ntiles + 255) / 256;                 // one workgroup per CU (a wave per SIMD with all its registers)
    const int gx = (ntiles + a.tiles_per_wg - 1) / a.tiles_per_wg;
    static LdsOptIn lds_opt;
    lds_opt.ensure(reinterpret_cast<const void*>(&conv_down_bf16_kernel), DOWN_LDS_BYTES);
    hipLaunchKernelGGL(conv_down_bf16_kernel, dim3(gx), dim3(256), DOWN_LDS_BYTES, st, a);
    return true;
}

}  // namespace dvae
