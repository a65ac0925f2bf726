// 3x3 (radius-1) stride-1 bf16 convolution with the input tile resident in LDS.
//
// The gather GEMM of igemm.hpp re-reads every input pixel once per tap (9x for a 3x3 conv) through
// L2; at bf16 MFMA rates that traffic, not the matrix pipe, bounds it.  Here one workgroup owns a
// TH x TW patch of output pixels (256 pixels) of one image, stages the (TH+2) x (TW+2) input halo in
// LDS once (zero-filled outside the image), and forms the A fragment of every tap by re-addressing
// that tile: lane r of a 32-pixel MFMA row block reads halo pixel (py+dy, px+dx) -- consecutive
// lanes hit consecutive 144-byte (64 ch) / 80-byte (32 ch) pixel rows, conflict-free for
// ds_read_b128.  Workgroups are persistent (one per CU, 8 waves): the weights of every tap stay in LDS for the
// workgroup's lifetime, tiles are strided over the grid, and the next tile's halo is fetched into
// registers while the current tile's MFMAs run, so a tile costs two barriers and no weight traffic.
// The MFMA is issued as D[channel][pixel] = W * X^T: each lane then owns one pixel and register
// quads of 4 consecutive channels, so the epilogue (scale/shift, residual, ReLU, accumulate) reads and
// writes 8-byte (bf16) / 16-byte (fp32) channel runs straight from registers, no LDS staging.
// Serves every single-phase radius-1 unit-stride plan: Conv2d 3x3 s1 forward and input-gradient, ConvTranspose2d 3x3 s1 forward and input-gradient (the plan supplies the taps).
#pragma once
#include "igemm.hpp"

namespace dvae {

struct HaloArgs {
    const void* in;
    const void* wp;
    void* out;
    const float* scale;
    const float* shift;
    const void* res;
    const float* in_ss;      // optional input normalisation [B][2][CIN]: x <- relu?(x * scale + shift) while staging
    int in_relu;
    int B, H, W, N, NS, Ktot;
    int relu, accumulate, out_f32;
    int lx, ly;              // log2 of the tile counts per image row / column (both powers of two)
    signed char dy[16], dx[16];
    unsigned long long* stamps;   // profiling aid: [grid.x][64 tiles][8] cycle-counter samples of wave 0, or null
};

// One cycle-counter sample per phase boundary of a tile (wave 0, lane 0 of each workgroup; first 64 tiles).
#define HALO_STAMP(i)                                                                               \
    do {                                                                                            \
        if (a.stamps && tid == 0 && it < 64)                                                        \
            a.stamps[((long)blockIdx.x * 64 + it) * 8 + (i)] = __builtin_readcyclecounter();        \
    } while (0)

template <int CIN, int NT, int TW>
__global__ __launch_bounds__(512) void conv_halo_bf16_kernel(const HaloArgs a) {
    constexpr int TH = 256 / TW, HWD = TW + 2, HP = (TH + 2) * HWD;
    constexpr int RS = CIN * 2 + 16;          // bytes per staged pixel / weight row (16-byte pad spreads banks)
    constexpr int VPP = CIN * 2 / 16;         // 16-byte vectors per pixel
    constexpr int HALO_BYTES = HP * RS;
    constexpr int WTAP = NT * RS;             // one tap's weight slab
    constexpr int TN = NT / 32, KS = CIN / 16, NSTEP = 9 * KS;
    constexpr int HV = (HP * VPP + 511) / 512;
    // persistent workgroup: all taps' weights stay in LDS, one halo tile at a time, next tile's halo in registers
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* wlds = smem;                                  // [9][NT][RS]
    unsigned char* halo = smem + 9 * WTAP;                       // [HP][RS]
    float* ssl = reinterpret_cast<float*>(smem + 9 * WTAP + HALO_BYTES);   // scale[NT], shift[NT]

    const int tid = threadIdx.x;
    const int n0 = blockIdx.y * NT;
    const bf16* in = reinterpret_cast<const bf16*>(a.in);
    const bf16* wp = reinterpret_cast<const bf16*>(a.wp);
    const int ntiles = a.B << (a.lx + a.ly);
    const int mx = (1 << a.lx) - 1, my = (1 << a.ly) - 1;

    // ---- weights (every tap) and the epilogue's per-channel scale/shift -> LDS, once -----------------
    for (int v = tid; v < 9 * NT * VPP; v += 512) {
        const int t = v / (NT * VPP), rem = v % (NT * VPP), n = rem / VPP, cv = rem % VPP;
        *reinterpret_cast<u32x4*>(wlds + t * WTAP + n * RS + cv * 16) =
            *reinterpret_cast<const u32x4*>(wp + (long)(n0 + n) * a.Ktot + t * CIN + cv * 8);
    }
    if (tid < NT) {
        const int n = n0 + tid;
        const bool live = n < a.N;
        ssl[tid] = (live && a.scale) ? a.scale[n] : (live ? 1.f : 0.f);
        ssl[NT + tid] = (live && a.shift) ? a.shift[n] : 0.f;
    }
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
    auto load_halo = [&](int tile) {
        const int tx = tile & mx, ty = (tile >> a.lx) & my, b = tile >> (a.lx + a.ly);
        const int y0 = ty * TH, x0 = tx * TW;
        const bf16* org = in + ((long)(b * a.H + y0) * a.W + x0) * CIN;
        inimg = 0;
        if (a.in_ss) {
            const float* p = a.in_ss + (long)b * 2 * CIN + (tid % VPP) * 8;
            isc[0] = *reinterpret_cast<const f32x4*>(p); isc[1] = *reinterpret_cast<const f32x4*>(p + 4);
            ish[0] = *reinterpret_cast<const f32x4*>(p + CIN); ish[1] = *reinterpret_cast<const f32x4*>(p + CIN + 4);
        }
#pragma unroll
        for (int k = 0; k < HV; ++k) {
            u32x4 val = {0u, 0u, 0u, 0u};
            const int iy = y0 - 1 + (hyx[k] >> 16), ix = x0 - 1 + (hyx[k] & 0xffff);
            if (hyx[k] >= 0 && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W) {
                val = *reinterpret_cast<const u32x4*>(org + hrel[k]);
                inimg |= 1u << k;
            }
            hreg[k] = val;
        }
    };
    auto store_halo = [&]() {
#pragma unroll
        for (int k = 0; k < HV; ++k) {
            const int v = tid + 512 * k;
            if (v >= HP * VPP) continue;
            u32x4 val = hreg[k];
            if (a.in_ss && (inimg >> k & 1u)) {           // zero padding stays zero: it pads the normalised tensor
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
    const unsigned char* wrow = wlds + r * RS + 16 * h;
    // this wave's epilogue patch: 32 pixels x 32 channels (bf16, one MFMA column block at a time), 80-byte rows
    constexpr int SRS = 32 * 2 + 16;
    unsigned char* stg = smem + 9 * WTAP + HALO_BYTES + 2 * NT * 4 + wave * (32 * SRS);
    const int sp = lane >> 2, sc8 = lane & 3;             // coalesced side: lane = (pixel sp of 16, 8-channel chunk sc8 of 4)
    auto out_off = [&](int tile, int j, int i) -> long {  // element offset of this lane's 16-byte piece of the output tile
        const int tx = tile & mx, ty = (tile >> a.lx) & my, b = tile >> (a.lx + a.ly);
        const int p = wave * 32 + i * 16 + sp;
        return ((long)(b * a.H + ty * TH + p / TW) * a.W + tx * TW + p % TW) * a.NS + n0 + j * 32 + sc8 * 8;
    };
    auto wave_lds_fence = [&]() {                          // LDS hand-over between lanes of one wave: order, no barrier
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    // operand fragments of flattened step st = (tap, 16-channel slab): one pixel vector, TN weight vectors
    auto frag = [&](int st, bf16x8& pv, bf16x8 (&wv)[TN]) {
        const int t = st / KS, sl = st % KS;
        pv = *reinterpret_cast<const bf16x8*>(hrow + toff[t] + 32 * sl);
#pragma unroll
        for (int j = 0; j < TN; ++j) wv[j] = *reinterpret_cast<const bf16x8*>(wrow + t * WTAP + j * 32 * RS + 32 * sl);
    };

    int tile = blockIdx.x;
    if (tile < ntiles) load_halo(tile);
    for (int it = 0; tile < ntiles; tile += gridDim.x, ++it) {
        HALO_STAMP(0);
        store_halo();
        HALO_STAMP(1);
        __syncthreads();                                   // halo (and, first time, weights) visible
        HALO_STAMP(2);
        const int next = tile + gridDim.x;
        if (next < ntiles) load_halo(next);                // in flight under the MFMAs below
        u32x4 rres[TN][2], racc[TN][2];
        if (!a.out_f32 && (a.res || a.accumulate)) {       // coalesced, and in flight under the MFMAs too
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const bool live = n0 + j * 32 + sc8 * 8 < a.NS;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const long o = out_off(tile, j, i);
                    const u32x4 z = {0u, 0u, 0u, 0u};
                    rres[j][i] = (a.res && live) ? *reinterpret_cast<const u32x4*>(reinterpret_cast<const bf16*>(a.res) + o) : z;
                    racc[j][i] = (a.accumulate && live) ? *reinterpret_cast<const u32x4*>(reinterpret_cast<const bf16*>(a.out) + o) : z;
                }
            }
        }
        HALO_STAMP(3);

        f32x16 acc[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
        {
            // fully unrolled over the 9 taps x KS slabs; the next step's fragments are read while this step's MFMAs issue
            bf16x8 pv[2], wv[2][TN];
            frag(0, pv[0], wv[0]);
#pragma unroll
            for (int st = 0; st < NSTEP; ++st) {
                if (st + 1 < NSTEP) frag(st + 1, pv[(st + 1) & 1], wv[(st + 1) & 1]);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    // D[i = out channel][j = pixel]: each lane ends up with 4 consecutive channels per register quad
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wv[st & 1][j], pv[st & 1], acc[j], 0, 0, 0);
            }
        }

        HALO_STAMP(4);
        if (a.out_f32) {
            const int tx = tile & mx, ty = (tile >> a.lx) & my, b = tile >> (a.lx + a.ly);
            // ---- fp32 output (narrow decoder heads): straight from registers, lane = pixel, quad g = channels 8g+4h.. ----
            const long pix = (long)(b * a.H + ty * TH + pl / TW) * a.W + tx * TW + pl % TW;
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
                    if (a.res) {
                        const f32x4 rv = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(a.res) + o);
#pragma unroll
                        for (int k = 0; k < 4; ++k) v[k] += rv[k];
                    }
                    if (a.relu) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], 0.f);
                    }
                    if (a.accumulate) {
                        const f32x4 ov = *reinterpret_cast<const f32x4*>(op + o);
#pragma unroll
                        for (int k = 0; k < 4; ++k) v[k] += ov[k];
                    }
                    f32x4 st4 = {v[0], v[1], v[2], v[3]};
                    *reinterpret_cast<f32x4*>(op + o) = st4;
                }
            }
        } else {
            // ---- bf16 output through this wave's private LDS patch: registers hold (pixel = lane, 4-channel quads), global
            // memory wants whole pixel rows.  Residual / old output arrive as coalesced 16-byte vectors (fetched before the
            // MFMAs), are re-read from the patch in register layout and added in fp32, so rounding happens once. ----
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
            bf16* op = reinterpret_cast<bf16*>(a.out);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                auto add_staged = [&](const u32x4 (&src)[2]) {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        *reinterpret_cast<u32x4*>(stg + (i * 16 + sp) * SRS + sc8 * 16) = src[i];
                    wave_lds_fence();
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const bf16x4 q = *reinterpret_cast<const bf16x4*>(stg + r * SRS + (8 * g + 4 * h) * 2);
#pragma unroll
                        for (int k = 0; k < 4; ++k) acc[j][4 * g + k] += (float)q[k];
                    }
                    wave_lds_fence();
                };
                if (a.res) add_staged(rres[j]);
                if (a.relu) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[j][e] = fmaxf(acc[j][e], 0.f);
                }
                if (a.accumulate) add_staged(racc[j]);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    bf16x4 q;
#pragma unroll
                    for (int k = 0; k < 4; ++k) q[k] = (bf16)acc[j][4 * g + k];
                    *reinterpret_cast<bf16x4*>(stg + r * SRS + (8 * g + 4 * h) * 2) = q;
                }
                wave_lds_fence();
                if (n0 + j * 32 + sc8 * 8 < a.NS) {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        *reinterpret_cast<u32x4*>(op + out_off(tile, j, i)) =
                            *reinterpret_cast<const u32x4*>(stg + (i * 16 + sp) * SRS + sc8 * 16);
                }
                wave_lds_fence();
            }
        }
        HALO_STAMP(5);
        __syncthreads();                                   // every wave is done reading this halo
        HALO_STAMP(6);
    }
}
#undef HALO_STAMP

static constexpr int halo_lds_bytes_of(int CIN, int NT, int TW) {
    // weights of 9 taps + one halo tile + epilogue scale/shift + the 8 waves' output patches
    return 9 * NT * (CIN * 2 + 16) + (256 / TW + 2) * (TW + 2) * (CIN * 2 + 16) + 2 * NT * 4 + 8 * 32 * 80;
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
    if (ilog2_exact(P.Wout / TW) < 0 || ilog2_exact(P.Hout / (256 / TW)) < 0) return none;
    const int NT = (P.Npad % 64 == 0) ? 64 : 32;
    if (halo_lds_bytes_of(P.CgS, NT, TW) > HALO_LDS_MAX) return none;
    const HaloGeom g = {TW, NT};
    return g;
}

// true if the plan runs on the halo kernel
static bool halo_applicable(const GemmPlan& P) { return halo_geom(P).TW != 0; }

static bool try_launch_halo_bf16(const GemmPlan& P, const void* in, const void* wp, void* out, const float* scale,
                                 const float* shift, const void* res, int relu, int accumulate, int out_f32,
                                 hipStream_t st, const float* in_ss = nullptr, int in_relu = 0,
                                 unsigned long long* stamps = nullptr) {
    const HaloGeom hg = halo_geom(P);
    if (!hg.TW) return false;
    const Phase& ph = P.ph[0];
    const int TW = hg.TW, NT = hg.NT;
    HaloArgs a;
    a.in = in; a.wp = wp; a.out = out; a.scale = scale; a.shift = shift; a.res = res;
    a.in_ss = in_ss; a.in_relu = in_relu; a.stamps = stamps;
    a.B = P.B; a.H = P.Hout; a.W = P.Wout; a.N = P.N; a.NS = out_f32 ? act_channels(P.N) : P.NS; a.Ktot = P.Ktot;
    a.relu = relu; a.accumulate = accumulate; a.out_f32 = out_f32;
    a.lx = ilog2_exact(P.Wout / TW); a.ly = ilog2_exact(P.Hout / (256 / TW));
    memcpy(a.dy, ph.dy, 16); memcpy(a.dx, ph.dx, 16);
    const int ntiles = P.B << (a.lx + a.ly);
    // one persistent workgroup per CU (its LDS footprint allows no more); tiles are strided over the grid
    dim3 grid(ntiles < 256 ? ntiles : 256, P.Npad / NT), block(512);
#define HALO_CASE(C, N, T)                                                                                   \
    if (P.CgS == C && NT == N && TW == T) {                                                                  \
        constexpr int lds = halo_lds_bytes<C, N, T>();                                                       \
        if (lds > HALO_LDS_MAX) return false;                                                                \
        static bool attr_set = false;                                                                        \
        if (!attr_set) {                                                                                     \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_halo_bf16_kernel<C, N, T>),        \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds);                      \
            attr_set = true;                                                                                 \
        }                                                                                                    \
        hipLaunchKernelGGL((conv_halo_bf16_kernel<C, N, T>), grid, block, lds, st, a);                       \
        return true;                                                                                         \
    }
    HALO_CASE(64, 64, 32) HALO_CASE(64, 32, 32) HALO_CASE(32, 64, 32) HALO_CASE(32, 32, 32)
    HALO_CASE(64, 64, 16) HALO_CASE(64, 32, 16) HALO_CASE(32, 64, 16) HALO_CASE(32, 32, 16)
#undef HALO_CASE
    return false;
}

}  // namespace dvae
