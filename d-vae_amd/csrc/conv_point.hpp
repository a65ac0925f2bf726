// 1x1 convolutions to 128 output channels in bf16, weights resident in LDS: post_conv (256 -> 128, unit stride) and
// layer2's shortcut (64 -> 128, stride 2).
//
// The gather GEMM treats a 1x1 conv like any other: every 128 x 128 tile re-stages its weights and pays two
// barriers per 64-deep K slab.  Here a persistent 512-thread workgroup keeps the whole 128 x CIN weight matrix
// in LDS and streams 128-pixel tiles of the input through a second buffer: the next tile's rows are
// fetched into registers while the current tile's MFMAs run, so a tile costs two barriers for CIN-deep
// work.  A wave owns 64 pixels x 32 output channels (two MFMAs per K step share one weight fragment: 1.5 KB of LDS
// reads per MFMA); results go through the wave's LDS patch and leave as 64-byte channel runs.  With 256 input channels
// the two buffers fill the LDS (one workgroup per CU); with 64 they take 58 KB and two workgroups share a CU.  The
// strided variant reads every other pixel of every other row: whole 128-byte pixels, so no fetched byte is wasted.
#pragma once
#include "igemm.hpp"

namespace dvae {

struct PointArgs {
    const bf16* in;            // [M][CIN] (unit stride) or [B][SD ho][SD wo][CIN]
    const bf16* wp;            // packed rows [128][Ktot]
    bf16* out;                 // [M][128]
    const float* scale;        // per output channel or NULL (1)
    const float* shift;        // per output channel or NULL (0)
    int relu, ntiles, Ktot;    // tiles of 128 output pixels
    // optional second output: out2 = out + emb[labels[image]][pixel] (class embedding in activation layout, hw pixels per
    // image; emb_px = 0: one row per class, the same for every pixel)
    int emb_px;              // table rows per image: hw (spatial table) or 0
    const bf16* emb;
    const int64_t* labels;
    bf16* out2;
    int hw;
    int ho, wo, lho, lwo;      // strided variant: output height / width (powers of two) and their logs
};

constexpr int PT_N = 128;
constexpr int PT_SRS = 32 * 2 + 16;                                                     // patch row: 32 channels + pad
template <int CIN>
static constexpr int point_lds_bytes() { return 2 * 128 * (CIN * 2 + 16) + 8 * 32 * PT_SRS + 2 * PT_N * 4; }

// SD: spatial stride of the gather (1, or 2 with Hout = Hin / 2)
template <int CIN, int SD>
__global__ __launch_bounds__(512) void conv_point_bf16_kernel(const PointArgs a) {
    constexpr int PT_CIN = CIN, PT_RS = CIN * 2 + 16, PT_VPP = CIN / 8;   // padded row; 16-byte vectors per row
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* wlds = smem;                          // [128 channels][PT_RS]
    unsigned char* xlds = smem + 128 * PT_RS;            // [128 pixels][PT_RS]
    unsigned char* patches = smem + 2 * 128 * PT_RS;
    float* ssl = reinterpret_cast<float*>(patches + 8 * 32 * PT_SRS);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    constexpr int XV = 128 * PT_VPP / 512;               // vectors per thread of a tile (and of the weights)

    {   // (every load issued before the first LDS write: a load-store loop waits out one memory latency per round)
        u32x4 wtmp[XV];
#pragma unroll
        for (int i = 0; i < XV; ++i) {
            const int v = tid + 512 * i, n = v / PT_VPP, cv = v % PT_VPP;
            wtmp[i] = *reinterpret_cast<const u32x4*>(a.wp + (long)n * a.Ktot + cv * 8);
        }
#pragma unroll
        for (int i = 0; i < XV; ++i) {
            const int v = tid + 512 * i, n = v / PT_VPP, cv = v % PT_VPP;
            *reinterpret_cast<u32x4*>(wlds + n * PT_RS + cv * 16) = wtmp[i];
        }
    }
    if (tid < PT_N) {
        ssl[tid] = a.scale ? a.scale[tid] : 1.f;
        ssl[PT_N + tid] = a.shift ? a.shift[tid] : 0.f;
    }
    u32x4 xr[XV];
    auto fetch = [&](int tile) {
        if (SD == 1) {                                    // a tile is 128 consecutive pixels: contiguous
            const u32x4* src = reinterpret_cast<const u32x4*>(a.in + (long)tile * 128 * PT_CIN);
#pragma unroll
            for (int k = 0; k < XV; ++k) xr[k] = src[tid + 512 * k];
        } else {                                          // output pixel (b, oy, ox) reads input pixel (b, SD oy, SD ox)
#pragma unroll
            for (int k = 0; k < XV; ++k) {
                const int v = tid + 512 * k, m = tile * 128 + v / PT_VPP, cv = v % PT_VPP;
                const int ox = m & (a.wo - 1), q = m >> a.lwo, oy = q & (a.ho - 1), b = q >> a.lho;
                xr[k] = *reinterpret_cast<const u32x4*>(a.in + ((long)(b * a.ho * SD + oy * SD) * (a.wo * SD) + ox * SD) * PT_CIN + cv * 8);
            }
        }
    };
    auto wave_lds_fence = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    // wave = (pixel half ph: 64 pixels, channel quarter cq: 32 channels)
    const int ph = wave >> 2, cq = wave & 3;
    const unsigned char* xrow = xlds + (ph * 64 + r) * PT_RS + 16 * h;
    const unsigned char* wrow = wlds + (cq * 32 + r) * PT_RS + 16 * h;
    unsigned char* stg = patches + wave * (32 * PT_SRS);
    const int sp = lane >> 2, sc8 = lane & 3;             // coalesced side: 16 pixels x 4 sixteen-byte chunks per instruction

    if ((int)blockIdx.x < a.ntiles) fetch(blockIdx.x);
    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        __syncthreads();                                  // the previous tile has been read by every wave
#pragma unroll
        for (int k = 0; k < XV; ++k) {
            const int v = tid + 512 * k;
            *reinterpret_cast<u32x4*>(xlds + (v / PT_VPP) * PT_RS + (v % PT_VPP) * 16) = xr[k];
        }
        __syncthreads();                                  // tile (and, first time, weights and scale / shift) visible
        if (tile + (int)gridDim.x < a.ntiles) fetch(tile + gridDim.x);

        f32x16 acc[2];
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[p][e] = 0.f;
#pragma unroll
        for (int s = 0; s < PT_CIN / 16; ++s) {
            const bf16x8 wv = *reinterpret_cast<const bf16x8*>(wrow + 32 * s);
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const bf16x8 pv = *reinterpret_cast<const bf16x8*>(xrow + p * 32 * PT_RS + 32 * s);
                // D[i = out channel][j = pixel]: a lane ends up with 4 consecutive channels per register quad
                acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wv, pv, acc[p], 0, 0, 0);
            }
        }
        bf16* ob = a.out + ((long)tile * 128 + ph * 64) * PT_N + cq * 32;
        // second output: a tile lies inside one image (128 divides hw), so its class row of the table is uniform; the
        // embedding is added on the coalesced side, to the 16-byte pieces on their way to `out`
        bf16* ob2 = nullptr;
        const bf16* eb = nullptr;
        if (a.out2) {
            const long m0 = (long)tile * 128;
            const int b = (int)(m0 / a.hw), px0 = (int)(m0 - (long)b * a.hw) + ph * 64;
            ob2 = a.out2 + (m0 + ph * 64) * PT_N + cq * 32;
            eb = a.emb + (a.emb_px ? (long)a.labels[b] * a.hw + px0 : (long)a.labels[b]) * PT_N + cq * 32;
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            u32x4 ev[2];
            if (a.out2) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    ev[i] = *reinterpret_cast<const u32x4*>(eb + (a.emb_px ? (long)(p * 32 + i * 16 + sp) * PT_N : 0l) + sc8 * 8);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nl = 8 * g + 4 * h;
                const f32x4 sc = *reinterpret_cast<const f32x4*>(ssl + cq * 32 + nl);
                const f32x4 sh = *reinterpret_cast<const f32x4*>(ssl + PT_N + cq * 32 + nl);
                bf16x4 o;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float v = acc[p][4 * g + k] * sc[k] + sh[k];
                    if (a.relu) v = fmaxf(v, 0.f);
                    o[k] = (bf16)v;
                }
                *reinterpret_cast<bf16x4*>(stg + r * PT_SRS + nl * 2) = o;
            }
            wave_lds_fence();
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int px = i * 16 + sp;
                const u32x4 v = *reinterpret_cast<const u32x4*>(stg + px * PT_SRS + sc8 * 16);
                *reinterpret_cast<u32x4*>(ob + (long)(p * 32 + px) * PT_N + sc8 * 8) = v;
                if (a.out2) {
                    const bf16x8 x = *reinterpret_cast<const bf16x8*>(&v), e = *reinterpret_cast<const bf16x8*>(&ev[i]);
                    bf16x8 o2;
#pragma unroll
                    for (int c = 0; c < 8; ++c) o2[c] = (bf16)((float)x[c] + (float)e[c]);   // the stored value + the embedding
                    *reinterpret_cast<bf16x8*>(ob2 + (long)(p * 32 + px) * PT_N + sc8 * 8) = o2;
                }
            }
            wave_lds_fence();
        }
    }
}

// Geometry the kernel is built for: CIN 256 with unit stride, or CIN 64 with stride 2.  0: not applicable.
static int point_cin(const GemmPlan& P) {
    if (P.nphase != 1 || P.N != PT_N || P.NS != PT_N) return 0;
    const Phase& ph = P.ph[0];
    if (ph.ntaps != 1 || ph.dy[0] || ph.dx[0] || ph.DSY != 1 || ph.DSX != 1 || ph.D0Y || ph.D0X) return 0;
    if (ph.Hg != P.Hout || ph.Wg != P.Wout || ph.SY != ph.SX) return 0;
    const long M = (long)P.B * P.Hout * P.Wout;
    if (M % 128 || M * 256 > 0x7fffffffL) return 0;
    if (P.CgS == 256 && ph.SY == 1 && P.Hin == P.Hout && P.Win == P.Wout) return 256;
    if (P.CgS == 64 && ph.SY == 2 && P.Hin == 2 * P.Hout && P.Win == 2 * P.Wout && ilog2_exact(P.Hout) >= 0 &&
        ilog2_exact(P.Wout) >= 0)
        return 64;
    return 0;
}
static bool point_applicable(const GemmPlan& P) { return point_cin(P) != 0; }

static bool try_launch_point_bf16(const GemmPlan& P, const void* in, const void* wp, void* out, const float* scale,
                                  const float* shift, const void* res, int relu, int accumulate, int out_f32,
                                  hipStream_t st, const void* emb = nullptr, const int64_t* labels = nullptr,
                                  void* out2 = nullptr, int emb_plain = 0) {
    const int cin = point_cin(P);
    if (res || accumulate || out_f32 || !cin) return false;
    if (out2 && (cin != 256 || (P.Hout * P.Wout) % 128)) return false;   // the second output wants a tile inside one image
    PointArgs a;
    a.emb = (const bf16*)emb; a.labels = labels; a.out2 = (bf16*)out2; a.hw = P.Hout * P.Wout;
    a.emb_px = emb_plain ? 0 : a.hw;
    a.in = (const bf16*)in; a.wp = (const bf16*)wp + P.ph[0].koff; a.out = (bf16*)out; a.scale = scale; a.shift = shift;
    a.relu = relu; a.ntiles = (int)((long)P.B * P.Hout * P.Wout / 128); a.Ktot = P.Ktot;
    a.ho = P.Hout; a.wo = P.Wout; a.lho = ilog2_exact(P.Hout); a.lwo = ilog2_exact(P.Wout);
    if (cin == 256) {
        static LdsOptIn lds_opt;
        lds_opt.ensure(reinterpret_cast<const void*>(&conv_point_bf16_kernel<256, 1>), point_lds_bytes<256>());
        const int grid = a.ntiles < 256 ? a.ntiles : 256;
        hipLaunchKernelGGL((conv_point_bf16_kernel<256, 1>), dim3(grid), dim3(512), point_lds_bytes<256>(), st, a);
    } else {
        static LdsOptIn lds_opt;
        lds_opt.ensure(reinterpret_cast<const void*>(&conv_point_bf16_kernel<64, 2>), point_lds_bytes<64>());
        const int grid = a.ntiles < 512 ? a.ntiles : 512;   // two workgroups per CU
        hipLaunchKernelGGL((conv_point_bf16_kernel<64, 2>), dim3(grid), dim3(512), point_lds_bytes<64>(), st, a);
    }
    return true;
}

}  // namespace dvae
