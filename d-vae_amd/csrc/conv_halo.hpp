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
    int relu, accumulate, out_f32, ntaps;
    int tiles_x, tiles_y;
    signed char dy[16], dx[16];
};

template <int CIN, int NT, int TW>
__global__ __launch_bounds__(512) void conv_halo_bf16_kernel(const HaloArgs a) {
    constexpr int TH = 256 / TW, HWD = TW + 2, HP = (TH + 2) * HWD;
    constexpr int RS = CIN * 2 + 16;          // bytes per staged pixel / weight row (16-byte pad spreads banks)
    constexpr int VPP = CIN * 2 / 16;         // 16-byte vectors per pixel
    constexpr int HALO_BYTES = HP * RS;
    constexpr int WTAP = NT * RS;             // one tap's weight slab
    constexpr int TN = NT / 32, KS = CIN / 16;
    constexpr int HV = (HP * VPP + 511) / 512;
    // persistent workgroup: all taps' weights stay in LDS, one halo tile at a time, next tile's halo in registers
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* wlds = smem;                                  // [ntaps][NT][RS]
    unsigned char* halo = smem + 9 * WTAP;                       // [HP][RS]
    float* ssl = reinterpret_cast<float*>(smem + 9 * WTAP + HALO_BYTES);   // scale[NT], shift[NT]

    const int tid = threadIdx.x;
    const int n0 = blockIdx.y * NT;
    const bf16* in = reinterpret_cast<const bf16*>(a.in);
    const bf16* wp = reinterpret_cast<const bf16*>(a.wp);
    const int tiles_per_img = a.tiles_x * a.tiles_y;
    const int ntiles = tiles_per_img * a.B;

    // ---- weights (every tap) and the epilogue's per-channel scale/shift -> LDS, once -----------------
    for (int v = tid; v < a.ntaps * NT * VPP; v += 512) {
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

    u32x4 hreg[HV];
    unsigned inimg = 0;                                   // bit k: hreg[k] is a pixel inside the image
    f32x4 isc[2], ish[2];                                 // input normalisation of this thread's 8 channels
    static_assert(512 % VPP == 0, "a thread keeps one channel group across its halo vectors");
    auto load_halo = [&](int tile) {
        const int tx = tile % a.tiles_x, ty = (tile / a.tiles_x) % a.tiles_y, b = tile / tiles_per_img;
        const int y0 = ty * TH, x0 = tx * TW;
        inimg = 0;
        if (a.in_ss) {
            const float* p = a.in_ss + (long)b * 2 * CIN + (tid % VPP) * 8;
            isc[0] = *reinterpret_cast<const f32x4*>(p); isc[1] = *reinterpret_cast<const f32x4*>(p + 4);
            ish[0] = *reinterpret_cast<const f32x4*>(p + CIN); ish[1] = *reinterpret_cast<const f32x4*>(p + CIN + 4);
        }
#pragma unroll
        for (int k = 0; k < HV; ++k) {
            const int v = tid + 512 * k;
            u32x4 val = {0u, 0u, 0u, 0u};
            if (v < HP * VPP) {
                const int hp = v / VPP, cv = v % VPP;
                const int iy = y0 - 1 + hp / HWD, ix = x0 - 1 + hp % HWD;
                if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W) {
                    val = *reinterpret_cast<const u32x4*>(in + ((long)(b * a.H + iy) * a.W + ix) * CIN + cv * 8);
                    inimg |= 1u << k;
                }
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
    const int hp0 = ((pl / TW + 1) * HWD + (pl % TW) + 1) * RS + 16 * h;
    const unsigned char* wrow = wlds + r * RS + 16 * h;

    int tile = blockIdx.x;
    if (tile < ntiles) load_halo(tile);
    for (; tile < ntiles; tile += gridDim.x) {
        store_halo();
        __syncthreads();                                   // halo (and, first time, weights) visible
        const int next = tile + gridDim.x;
        if (next < ntiles) load_halo(next);                // in flight under the MFMAs below

        f32x16 acc[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
        for (int t = 0; t < a.ntaps; ++t) {
            const int toff = ((int)a.dy[t] * HWD + (int)a.dx[t]) * RS;
            const unsigned char* wt = wrow + t * WTAP;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const bf16x8 pv = *reinterpret_cast<const bf16x8*>(halo + hp0 + toff + 32 * s);
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const bf16x8 wv = *reinterpret_cast<const bf16x8*>(wt + j * 32 * RS + 32 * s);
                    // D[i = out channel][j = pixel]: each lane ends up with 4 consecutive channels per register quad
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wv, pv, acc[j], 0, 0, 0);
                }
            }
        }

        // ---- epilogue straight from registers: lane = pixel, register quad g = channels 8g + 4h .. +3 ----
        const int tx = tile % a.tiles_x, ty = (tile / a.tiles_x) % a.tiles_y, b = tile / tiles_per_img;
        const long pix = (long)(b * a.H + ty * TH + pl / TW) * a.W + tx * TW + pl % TW;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nl = j * 32 + 8 * g + 4 * h;         // first of 4 consecutive local channels
                const int nn = n0 + nl;
                if (nn >= a.NS) continue;
                const f32x4 sc = *reinterpret_cast<const f32x4*>(ssl + nl);
                const f32x4 sh = *reinterpret_cast<const f32x4*>(ssl + NT + nl);
                float v[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = acc[j][4 * g + k] * sc[k] + sh[k];
                const long o = pix * a.NS + nn;
                if (a.out_f32) {
                    float* op = reinterpret_cast<float*>(a.out);
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
                } else {
                    bf16* op = reinterpret_cast<bf16*>(a.out);
                    if (a.res) {
                        float rv[4];
                        Act<bf16>::load4(reinterpret_cast<const bf16*>(a.res) + o, rv);
#pragma unroll
                        for (int k = 0; k < 4; ++k) v[k] += rv[k];
                    }
                    if (a.relu) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], 0.f);
                    }
                    if (a.accumulate) {
                        float ov[4];
                        Act<bf16>::load4(op + o, ov);
#pragma unroll
                        for (int k = 0; k < 4; ++k) v[k] += ov[k];
                    }
                    Act<bf16>::store4(op + o, v);
                }
            }
        }
        __syncthreads();                                   // every wave is done reading this halo
    }
}

template <int CIN, int NT, int TW>
static constexpr int halo_lds_bytes() {
    return 9 * NT * (CIN * 2 + 16) + (256 / TW + 2) * (TW + 2) * (CIN * 2 + 16) + 2 * NT * 4;
}

// true if the plan was launched on the halo kernel
static bool halo_applicable(const GemmPlan& P) {
    if (P.nphase != 1) return false;
    const Phase& ph = P.ph[0];
    if (ph.SY != 1 || ph.SX != 1 || ph.DSY != 1 || ph.DSX != 1 || ph.D0Y || ph.D0X) return false;
    if (P.Hin != P.Hout || P.Win != P.Wout || ph.Hg != P.Hout || ph.Wg != P.Wout) return false;
    if (ph.ntaps < 1 || ph.ntaps > 9) return false;
    for (int t = 0; t < ph.ntaps; ++t)
        if (ph.dy[t] < -1 || ph.dy[t] > 1 || ph.dx[t] < -1 || ph.dx[t] > 1) return false;
    if (P.CgS != 32 && P.CgS != 64) return false;
    const int TW = (P.Wout % 32 == 0) ? 32 : ((P.Wout % 16 == 0) ? 16 : 0);
    return TW && P.Hout % (256 / TW) == 0;
}

static bool try_launch_halo_bf16(const GemmPlan& P, const void* in, const void* wp, void* out, const float* scale,
                                 const float* shift, const void* res, int relu, int accumulate, int out_f32,
                                 hipStream_t st, const float* in_ss = nullptr, int in_relu = 0) {
    if (P.nphase != 1) return false;
    const Phase& ph = P.ph[0];
    if (ph.SY != 1 || ph.SX != 1 || ph.DSY != 1 || ph.DSX != 1 || ph.D0Y || ph.D0X) return false;
    if (P.Hin != P.Hout || P.Win != P.Wout || ph.Hg != P.Hout || ph.Wg != P.Wout) return false;
    if (ph.ntaps < 1 || ph.ntaps > 9) return false;
    for (int t = 0; t < ph.ntaps; ++t)
        if (ph.dy[t] < -1 || ph.dy[t] > 1 || ph.dx[t] < -1 || ph.dx[t] > 1) return false;
    if (P.CgS != 32 && P.CgS != 64) return false;
    const int TW = (P.Wout % 32 == 0) ? 32 : ((P.Wout % 16 == 0) ? 16 : 0);
    if (!TW || P.Hout % (256 / TW)) return false;
    HaloArgs a;
    a.in = in; a.wp = wp; a.out = out; a.scale = scale; a.shift = shift; a.res = res;
    a.in_ss = in_ss; a.in_relu = in_relu;
    a.B = P.B; a.H = P.Hout; a.W = P.Wout; a.N = P.N; a.NS = out_f32 ? act_channels(P.N) : P.NS; a.Ktot = P.Ktot;
    a.relu = relu; a.accumulate = accumulate; a.out_f32 = out_f32; a.ntaps = ph.ntaps;
    a.tiles_x = P.Wout / TW; a.tiles_y = P.Hout / (256 / TW);
    memcpy(a.dy, ph.dy, 16); memcpy(a.dx, ph.dx, 16);
    const int NT = (P.Npad % 64 == 0) ? 64 : 32;
    const int ntiles = a.tiles_x * a.tiles_y * P.B;
    // one persistent workgroup per CU (its LDS footprint allows no more); tiles are strided over the grid
    dim3 grid(ntiles < 256 ? ntiles : 256, P.Npad / NT), block(512);
#define HALO_CASE(C, N, T)                                                                                   \
    if (P.CgS == C && NT == N && TW == T) {                                                                  \
        constexpr int lds = halo_lds_bytes<C, N, T>();                                                       \
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
