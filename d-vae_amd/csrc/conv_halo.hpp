// 3x3 (radius-1) stride-1 bf16 convolution with the input tile resident in LDS.
//
// The gather GEMM of igemm.hpp re-reads every input pixel once per tap (9x for a 3x3 conv) through
// L2; at bf16 MFMA rates that traffic, not the matrix pipe, bounds it.  Here one workgroup owns a
// TH x TW patch of output pixels (256 pixels) of one image, stages the (TH+2) x (TW+2) input halo in
// LDS once (zero-filled outside the image), and forms the A fragment of every tap by re-addressing
// that tile: lane r of a 32-pixel MFMA row block reads halo pixel (py+dy, px+dx) -- consecutive
// lanes hit consecutive 144-byte (64 ch) / 80-byte (32 ch) pixel rows, conflict-free for
// ds_read_b128.  Only the per-tap weight slab ([NT][CIN], 4-8 KB, L2-resident) streams through a
// double-buffered LDS ring, one barrier per tap.  The output tile is staged in LDS (overlaying the
// halo) so that global stores, the residual read and the accumulate read are 16-byte vectors over
// whole pixel rows.  Serves every single-phase radius-1 unit-stride plan: Conv2d 3x3 s1 forward and
// input-gradient, ConvTranspose2d 3x3 s1 forward and input-gradient (the plan supplies the taps).
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
    int B, H, W, N, NS, Ktot;
    int relu, accumulate, out_f32, ntaps;
    int tiles_x, tiles_y;
    signed char dy[16], dx[16];
};

template <int CIN, int NT, int TW>
__global__ __launch_bounds__(256) void conv_halo_bf16_kernel(const HaloArgs a) {
    constexpr int TH = 256 / TW, HWD = TW + 2, HP = (TH + 2) * HWD;
    constexpr int RS = CIN * 2 + 16;          // bytes per staged pixel / weight row (16-byte pad spreads banks)
    constexpr int VPP = CIN * 2 / 16;         // 16-byte vectors per pixel
    constexpr int HALO_BYTES = HP * RS, STAGE_BYTES = 256 * NT * 2;
    constexpr int A_BYTES = HALO_BYTES > STAGE_BYTES ? HALO_BYTES : STAGE_BYTES;
    constexpr int WB = NT * RS;               // one weight slab
    constexpr int WV = (NT * VPP + 255) / 256;
    constexpr int TM = 2, TN = NT / 32, KS = CIN / 16;
    __shared__ __attribute__((aligned(16))) unsigned char smem[A_BYTES + 2 * WB];
    unsigned char* halo = smem;
    unsigned char* wbuf = smem + A_BYTES;

    const int tid = threadIdx.x;
    const int bid = blockIdx.x;
    const int tx = bid % a.tiles_x, ty = (bid / a.tiles_x) % a.tiles_y, b = bid / (a.tiles_x * a.tiles_y);
    const int y0 = ty * TH, x0 = tx * TW;
    const int n0 = blockIdx.y * NT;
    const bf16* in = reinterpret_cast<const bf16*>(a.in);
    const bf16* wp = reinterpret_cast<const bf16*>(a.wp);

    u32x4 wreg[WV];
    auto load_w = [&](int t) {
#pragma unroll
        for (int k = 0; k < WV; ++k) {
            const int v = tid + 256 * k;
            if (NT * VPP % 256 == 0 || v < NT * VPP) {
                const int n = v / VPP, cv = v % VPP;
                wreg[k] = *reinterpret_cast<const u32x4*>(wp + (long)(n0 + n) * a.Ktot + t * CIN + cv * 8);
            }
        }
    };
    auto store_w = [&](int buf) {
#pragma unroll
        for (int k = 0; k < WV; ++k) {
            const int v = tid + 256 * k;
            if (NT * VPP % 256 == 0 || v < NT * VPP) {
                const int n = v / VPP, cv = v % VPP;
                *reinterpret_cast<u32x4*>(wbuf + buf * WB + n * RS + cv * 16) = wreg[k];
            }
        }
    };

    load_w(0);
    // ---- input halo -> LDS: every global load is issued before the first LDS store ---------------
    constexpr int HV = (HP * VPP + 255) / 256;
    u32x4 hreg[HV];
#pragma unroll
    for (int k = 0; k < HV; ++k) {
        const int v = tid + 256 * k;
        u32x4 val = {0u, 0u, 0u, 0u};
        if (v < HP * VPP) {
            const int hp = v / VPP, cv = v % VPP;
            const int iy = y0 - 1 + hp / HWD, ix = x0 - 1 + hp % HWD;
            if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
                val = *reinterpret_cast<const u32x4*>(in + ((long)(b * a.H + iy) * a.W + ix) * CIN + cv * 8);
        }
        hreg[k] = val;
    }
#pragma unroll
    for (int k = 0; k < HV; ++k) {
        const int v = tid + 256 * k;
        if (v < HP * VPP) *reinterpret_cast<u32x4*>(halo + (v / VPP) * RS + (v % VPP) * 16) = hreg[k];
    }
    store_w(0);
    __syncthreads();

    const int wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    int hp0[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int pl = (wave * TM + i) * 32 + r;
        hp0[i] = ((pl / TW + 1) * HWD + (pl % TW) + 1) * RS + 16 * h;
    }
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    for (int t = 0; t < a.ntaps; ++t) {
        if (t + 1 < a.ntaps) load_w(t + 1);
        const int toff = ((int)a.dy[t] * HWD + (int)a.dx[t]) * RS;
        const unsigned char* wb = wbuf + (t & 1) * WB + r * RS + 16 * h;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            bf16x8 av[TM], bv[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) av[i] = *reinterpret_cast<const bf16x8*>(halo + hp0[i] + toff + 32 * s);
#pragma unroll
            for (int j = 0; j < TN; ++j) bv[j] = *reinterpret_cast<const bf16x8*>(wb + j * 32 * RS + 32 * s);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
        if (t + 1 < a.ntaps) store_w((t + 1) & 1);
        __syncthreads();
    }

    // ---- epilogue ------------------------------------------------------------------------------
    if (a.out_f32) {
        // narrow fp32 outputs (the 3-channel decoder heads): direct stores
        float* op = reinterpret_cast<float*>(a.out);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + j * 32 + r;
            if (n >= a.NS) continue;
            const bool live = n < a.N;
            const float sc = (live && a.scale) ? a.scale[n] : 1.f;
            const float sh = (live && a.shift) ? a.shift[n] : 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int pl = (wave * TM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    const long o = ((long)(b * a.H + y0 + pl / TW) * a.W + x0 + pl % TW) * a.NS + n;
                    float v = acc[i][j][e] * sc + sh;
                    if (a.res) v += reinterpret_cast<const float*>(a.res)[o];
                    if (a.relu) v = fmaxf(v, 0.f);
                    if (!live) v = 0.f;
                    if (a.accumulate) v += op[o];
                    op[o] = v;
                }
        }
        return;
    }
    bf16* stage = reinterpret_cast<bf16*>(smem);     // [256 pixels][NT]; the halo is dead after the last barrier
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + j * 32 + r;
        const bool live = n < a.N;
        const float sc = (live && a.scale) ? a.scale[n] : 1.f;
        const float sh = (live && a.shift) ? a.shift[n] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int pl = (wave * TM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                stage[pl * NT + j * 32 + r] = (bf16)(live ? acc[i][j][e] * sc + sh : 0.f);
            }
    }
    __syncthreads();
    constexpr int OV = NT / 8;                       // 16-byte vectors per staged pixel
    bf16* out = reinterpret_cast<bf16*>(a.out);
    const bf16* res = reinterpret_cast<const bf16*>(a.res);
    for (int v = tid; v < 256 * OV; v += 256) {
        const int pl = v / OV, ov = v % OV;
        const int nn = n0 + ov * 8;
        if (nn >= a.NS) continue;
        const long o = ((long)(b * a.H + y0 + pl / TW) * a.W + x0 + pl % TW) * a.NS + nn;
        bf16x8 sv = *reinterpret_cast<const bf16x8*>(stage + pl * NT + ov * 8);
        if (res || a.relu || a.accumulate) {
            float f[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) f[k] = (float)sv[k];
            if (res) {
                const bf16x8 rv = *reinterpret_cast<const bf16x8*>(res + o);
#pragma unroll
                for (int k = 0; k < 8; ++k) f[k] += (float)rv[k];
            }
            if (a.relu) {
#pragma unroll
                for (int k = 0; k < 8; ++k) f[k] = fmaxf(f[k], 0.f);
            }
            if (a.accumulate) {
                const bf16x8 ov8 = *reinterpret_cast<const bf16x8*>(out + o);
#pragma unroll
                for (int k = 0; k < 8; ++k) f[k] += (float)ov8[k];
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) sv[k] = (bf16)f[k];
        }
        *reinterpret_cast<bf16x8*>(out + o) = sv;
    }
}

// true if the plan was launched on the halo kernel
static bool try_launch_halo_bf16(const GemmPlan& P, const void* in, const void* wp, void* out, const float* scale,
                                 const float* shift, const void* res, int relu, int accumulate, int out_f32,
                                 hipStream_t st) {
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
    if (out_f32 && (res || P.NS > 32)) return false;
    HaloArgs a;
    a.in = in; a.wp = wp; a.out = out; a.scale = scale; a.shift = shift; a.res = res;
    a.B = P.B; a.H = P.Hout; a.W = P.Wout; a.N = P.N; a.NS = out_f32 ? act_channels(P.N) : P.NS; a.Ktot = P.Ktot;
    a.relu = relu; a.accumulate = accumulate; a.out_f32 = out_f32; a.ntaps = ph.ntaps;
    a.tiles_x = P.Wout / TW; a.tiles_y = P.Hout / (256 / TW);
    memcpy(a.dy, ph.dy, 16); memcpy(a.dx, ph.dx, 16);
    const int NT = (P.Npad % 64 == 0) ? 64 : 32;
    dim3 grid(a.tiles_x * a.tiles_y * P.B, P.Npad / NT), block(256);
#define HALO_CASE(C, N, T)                                                                                   \
    if (P.CgS == C && NT == N && TW == T) {                                                                  \
        hipLaunchKernelGGL((conv_halo_bf16_kernel<C, N, T>), grid, block, 0, st, a);                         \
        return true;                                                                                         \
    }
    HALO_CASE(64, 64, 32) HALO_CASE(64, 32, 32) HALO_CASE(32, 64, 32) HALO_CASE(32, 32, 32)
    HALO_CASE(64, 64, 16) HALO_CASE(64, 32, 16) HALO_CASE(32, 64, 16) HALO_CASE(32, 32, 16)
#undef HALO_CASE
    return false;
}

}  // namespace dvae
