// Implicit-GEMM convolution family for gfx950 (MFMA, LDS-staged gathered tiles).
//
// One "gather GEMM" covers Conv2d forward, Conv2d input-gradient, ConvTranspose2d forward and
// ConvTranspose2d input-gradient:
//
//     out[dst(m), n] = sum_{t < ntaps} sum_{c < Cg}  in[src(m, t), c] * W[n, t, c]
//
// m enumerates (b, y, x) of an output sub-grid ("phase"); src(m,t) = (b, y*SY + dy[t], x*SX + dx[t])
// (zero outside the tensor); dst(m) = (b, y*DSY + D0Y, x*DSX + D0X).  A stride-s transposed gather
// (Conv2d dgrad, ConvTranspose2d forward) is split into s*s phases by output parity so that no MFMA
// work is spent on taps that never hit.  Activations are NHWC, so the GEMM K index (t, c) is
// contiguous in memory per tap and a 16-byte vector never straddles taps.
//
// The weight-gradient kernel contracts over pixels instead:
//     dW[t, cx, cd] = sum_m  X[src(m, t), cx] * D[m, cd]
// with per-chunk partial slabs in a caller-provided workspace that a second kernel sums in a fixed
// order and scatters into the reference's weight layout (deterministic, no float atomics).
#pragma once
#include "common.hpp"
#include "philox.hpp"

namespace dvae {

struct Phase {
    int Hg, Wg;           // sub-grid of output positions handled by this phase
    int SY, SX;           // source step per grid step
    int DSY, DSX, D0Y, D0X;
    int ntaps;
    int koff;             // first K column of this phase in the packed weight rows
    signed char dy[16], dx[16];
};

struct PackPhase {
    int ntaps, koff, kcols;   // kcols = K columns of this phase, padded to a multiple of BK
    int woff[16];             // ky*kw + kx of each tap in the reference weight
};

struct GemmPlan {
    int B;
    int Hin, Win, Cg, CgS, cg_log2;   // gathered tensor (logical / padded channels)
    int Hout, Wout, N, NS, Npad;      // output tensor; N logical, NS act channel stride
    int nphase;
    Phase ph[4];
    PackPhase pk[4];
    int Ktot;
    int sN, sC;                       // reference weight index = n*sN + c*sC + woff
};

static inline int act_channels(int C) {
    int p = 4;
    while (p < C) p <<= 1;
    return p;
}

// Build the gather plan of op (DVAE_OP_FWD / DVAE_OP_DGRAD) for a conv descriptor.
static int build_plan(const dvae_conv_desc& d, int op, int dtype, GemmPlan& P) {
    if (d.B <= 0 || d.H <= 0 || d.W <= 0 || d.Cin <= 0 || d.Cout <= 0 || d.kh <= 0 || d.kw <= 0 ||
        d.stride <= 0 || d.pad < 0 || d.kh * d.kw > 16 || d.stride > 2)
        DVAE_FAIL(DVAE_EINVAL, "conv desc out of range (kh*kw <= 16, stride <= 2)");
    if (dtype != DVAE_F32 && dtype != DVAE_BF16) DVAE_FAIL(DVAE_EDTYPE, "unknown dtype %d", dtype);
    const int BK = dtype == DVAE_F32 ? 32 : 64;
    const int s = d.stride, p = d.pad;
    int Ho, Wo;
    if (!d.transposed) {
        Ho = (d.H + 2 * p - d.kh) / s + 1;
        Wo = (d.W + 2 * p - d.kw) / s + 1;
    } else {
        Ho = (d.H - 1) * s - 2 * p + d.kh;
        Wo = (d.W - 1) * s - 2 * p + d.kw;
    }
    if (Ho <= 0 || Wo <= 0) DVAE_FAIL(DVAE_EINVAL, "empty conv output");
    memset(&P, 0, sizeof(P));
    P.B = d.B;
    // direct gather: Conv2d fwd, ConvT dgrad.  transposed gather: Conv2d dgrad, ConvT fwd.
    const bool direct = (!d.transposed && op == DVAE_OP_FWD) || (d.transposed && op == DVAE_OP_DGRAD);
    const int kk = d.kh * d.kw;
    if (!d.transposed && op == DVAE_OP_FWD) {        // x -> y
        P.Hin = d.H; P.Win = d.W; P.Cg = d.Cin; P.Hout = Ho; P.Wout = Wo; P.N = d.Cout;
        P.sN = d.Cin * kk; P.sC = kk;
    } else if (!d.transposed) {                      // dy -> dx
        P.Hin = Ho; P.Win = Wo; P.Cg = d.Cout; P.Hout = d.H; P.Wout = d.W; P.N = d.Cin;
        P.sN = kk; P.sC = d.Cin * kk;
    } else if (op == DVAE_OP_FWD) {                  // convT: x -> y   (weight [Cin][Cout][kh][kw])
        P.Hin = d.H; P.Win = d.W; P.Cg = d.Cin; P.Hout = Ho; P.Wout = Wo; P.N = d.Cout;
        P.sN = kk; P.sC = d.Cout * kk;
    } else {                                         // convT: dy -> dx
        P.Hin = Ho; P.Win = Wo; P.Cg = d.Cout; P.Hout = d.H; P.Wout = d.W; P.N = d.Cin;
        P.sN = d.Cout * kk; P.sC = kk;
    }
    P.CgS = act_channels(P.Cg);
    if (dtype == DVAE_BF16 && P.CgS < 8) P.CgS = 8;
    P.cg_log2 = ilog2_exact(P.CgS);
    P.NS = act_channels(P.N);
    if (dtype == DVAE_BF16 && P.NS < 8) P.NS = 8;
    P.Npad = round_up(P.NS, 32);
    int koff = 0;
    if (direct) {
        P.nphase = 1;
        Phase& ph = P.ph[0];
        ph.Hg = P.Hout; ph.Wg = P.Wout; ph.SY = s; ph.SX = s; ph.DSY = 1; ph.DSX = 1; ph.D0Y = 0; ph.D0X = 0;
        ph.ntaps = kk; ph.koff = 0;
        for (int ky = 0; ky < d.kh; ++ky)
            for (int kx = 0; kx < d.kw; ++kx) {
                int t = ky * d.kw + kx;
                ph.dy[t] = (signed char)(ky - p); ph.dx[t] = (signed char)(kx - p);
                P.pk[0].woff[t] = t;
            }
        P.pk[0].ntaps = kk; P.pk[0].koff = 0; P.pk[0].kcols = round_up(kk * P.CgS, BK);
        koff = P.pk[0].kcols;
    } else {
        P.nphase = s * s;
        for (int py = 0; py < s; ++py)
            for (int px = 0; px < s; ++px) {
                int i = py * s + px;
                Phase& ph = P.ph[i];
                ph.Hg = (P.Hout - py + s - 1) / s; ph.Wg = (P.Wout - px + s - 1) / s;
                ph.SY = 1; ph.SX = 1; ph.DSY = s; ph.DSX = s; ph.D0Y = py; ph.D0X = px;
                int nt = 0;
                for (int ky = 0; ky < d.kh; ++ky) {
                    int vy = py + p - ky;
                    if (((vy % s) + s) % s) continue;
                    for (int kx = 0; kx < d.kw; ++kx) {
                        int vx = px + p - kx;
                        if (((vx % s) + s) % s) continue;
                        ph.dy[nt] = (signed char)(vy / s); ph.dx[nt] = (signed char)(vx / s);
                        P.pk[i].woff[nt] = ky * d.kw + kx;
                        ++nt;
                    }
                }
                ph.ntaps = nt; ph.koff = koff;
                P.pk[i].ntaps = nt; P.pk[i].koff = koff; P.pk[i].kcols = round_up(nt * P.CgS, BK);
                koff += P.pk[i].kcols;
            }
    }
    P.Ktot = koff > 0 ? koff : BK;
    return DVAE_OK;
}

// ------------------------------------------------------------------------------------------------
// weight packing: reference layout fp32 -> [Npad][Ktot] operand image (K contiguous), dtype T
// ------------------------------------------------------------------------------------------------
struct PackArgs {
    const float* w;
    void* out;
    int N, Npad, Cg, cg_log2, Ktot, sN, sC, nphase;
    PackPhase pk[4];
};

template <typename T>
__global__ void pack_weight_kernel(const PackArgs a) {
    const long total = (long)a.Npad * a.Ktot;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int n = (int)(e / a.Ktot), k = (int)(e % a.Ktot);
        float v = 0.f;
        if (n < a.N) {
            for (int i = 0; i < a.nphase; ++i) {
                const PackPhase& pk = a.pk[i];
                if (k >= pk.koff && k < pk.koff + pk.kcols) {
                    const int kk = k - pk.koff;
                    const int tap = kk >> a.cg_log2, c = kk & ((1 << a.cg_log2) - 1);
                    if (tap < pk.ntaps && c < a.Cg) v = a.w[(long)n * a.sN + (long)c * a.sC + pk.woff[tap]];
                }
            }
        }
        reinterpret_cast<T*>(a.out)[e] = (T)v;
    }
}

// All conv weights of a model in one launch: blockIdx.y selects a job (a PackArgs record in device
// memory), blockIdx.x grid-strides over that job's elements.
template <typename T>
__global__ void pack_weight_batch_kernel(const PackArgs* jobs) {
    const PackArgs& a = jobs[blockIdx.y];
    const long total = (long)a.Npad * a.Ktot;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int n = (int)(e / a.Ktot), k = (int)(e % a.Ktot);
        float v = 0.f;
        if (n < a.N) {
            for (int i = 0; i < a.nphase; ++i) {
                const PackPhase& pk = a.pk[i];
                if (k >= pk.koff && k < pk.koff + pk.kcols) {
                    const int kk = k - pk.koff;
                    const int tap = kk >> a.cg_log2, c = kk & ((1 << a.cg_log2) - 1);
                    if (tap < pk.ntaps && c < a.Cg) v = a.w[(long)n * a.sN + (long)c * a.sC + pk.woff[tap]];
                }
            }
        }
        reinterpret_cast<T*>(a.out)[e] = (T)v;
    }
}

// ------------------------------------------------------------------------------------------------
// gather GEMM
// ------------------------------------------------------------------------------------------------
struct GemmArgs {
    const void* in;
    const void* wp;
    void* out;
    const float* scale;
    const float* shift;
    const void* res;
    int B, Hin, Win, cg_log2;
    int Hout, Wout, N, NS;
    int Ktot;
    int relu, accumulate, out_f32;
    Phase ph[4];
    // fused encoder head (ZF kernels only): out = z [rows][zL] = eps * exp(lv / 2) + mu with (mu | lv) = this GEMM's 2*zL columns
    int zL, zrpi;            // latent channels; GEMM rows per image (the Philox counter restarts per image)
    PhiloxKey zkey;
};

constexpr int LDS_ROW = 144;   // bytes per staged tile row: 128 B of K data + 16 B pad (bank spread)

template <typename T> struct Mfma;
template <> struct Mfma<float> {
    // 128-byte K slab = 32 floats.  v_mfma_f32_32x32x2_f32: lane (r = l&31, h = l>>5) supplies
    // A[r][k=h] and B[k=h][r].  The weight fragment goes in as A and the pixel fragment as B, so the
    // result tile is D[channel][pixel]: lane r owns one pixel and each register quad 4 consecutive channels.  Each lane reads 4 consecutive floats at column 8q+4h, so MFMA e of
    // group q contracts columns {8q+e, 8q+4+e}; A and B use the same map, the sum is unchanged.
    __device__ static inline void run(const unsigned char* arow, const unsigned char* brow, int h, f32x16& acc) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 a = *reinterpret_cast<const f32x4*>(arow + 32 * q + 16 * h);
            f32x4 b = *reinterpret_cast<const f32x4*>(brow + 32 * q + 16 * h);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b[e], a[e], acc, 0, 0, 0);
        }
    }
};

template <> struct Mfma<bf16> {
    // 128-byte K slab = 64 bf16.  v_mfma_f32_32x32x16_bf16: lane (r, h) supplies A[r][k = 8h + j] and
    // B[k = 8h + j][r], j = 0..7: one 16-byte LDS read per operand per MFMA, four MFMAs per slab.
    __device__ static inline void run(const unsigned char* arow, const unsigned char* brow, int h, f32x16& acc) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            bf16x8 a = *reinterpret_cast<const bf16x8*>(arow + 32 * s + 16 * h);
            bf16x8 b = *reinterpret_cast<const bf16x8*>(brow + 32 * s + 16 * h);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc, 0, 0, 0);
        }
    }
};

// ZF (fused encoder head, 1x1 GEMM with N = 2*zL columns [mu | log_var]): the weight rows are taken in an order that
// gives every wave 16*TN latent channels as TN/2 column blocks of mu followed by the same channels' log_var, so a
// lane holds mu and log_var of the same channel quad; the epilogue draws eps, forms z and writes only z.
// Column cl of N-tile t (BN columns = BN/2 latent channels) reads weight row zrow(t, cl).
template <int BN, int WGN, int TN>
__device__ __forceinline__ int zf_row(int t, int cl, int zL) {
    constexpr int WC = BN / WGN;                 // GEMM columns per wave
    const int w = cl / WC, jj = (cl % WC) / (WC / 2), c = cl % (WC / 2);
    return jj * zL + t * (BN / 2) + w * (WC / 2) + c;
}

template <typename T, int TM, int TN, int WGM, int WGN, bool ZF = false>
__global__ __launch_bounds__(256) void igemm_kernel(const GemmArgs a) {
    constexpr int BM = WGM * 32 * TM, BN = WGN * 32 * TN;
    constexpr int VEC = 16 / sizeof(T);
    constexpr int BK = 8 * VEC;
    constexpr int AP = BM / 32, BP = BN / 32;
    __shared__ __attribute__((aligned(16))) unsigned char smem[(BM + BN) * LDS_ROW];
    __shared__ int rowdst[BM];
    __shared__ int taptab[16];
    __shared__ __attribute__((aligned(16))) float ssl[2 * BN];   // epilogue scale[BN], shift[BN] (0, 0 on pad channels)
    unsigned char* As = smem;
    unsigned char* Bs = smem + BM * LDS_ROW;

    const Phase& ph = a.ph[blockIdx.z];
    const int M = a.B * ph.Hg * ph.Wg;
    const int m0 = blockIdx.x * BM;
    if (m0 >= M) return;
    const int n0 = blockIdx.y * BN;
    const int tid = threadIdx.x;
    const int kvec = tid & 7, lrow = tid >> 3;

    if (tid < 16) taptab[tid] = ((int)ph.dy[tid] << 16) | ((int)ph.dx[tid] & 0xffff);
    for (int c = tid; c < BN; c += 256) {
        const int n = ZF ? zf_row<BN, WGN, TN>(blockIdx.y, c, a.zL) : n0 + c;
        const bool live = n < a.N;
        ssl[c] = live ? (a.scale ? a.scale[n] : 1.f) : 0.f;
        ssl[BN + c] = (live && a.shift) ? a.shift[n] : 0.f;
    }
    if (tid < BM) {
        const int m = m0 + tid;
        int dst = -1;
        if (m < M) {
            const int x = m % ph.Wg, t = m / ph.Wg, y = t % ph.Hg, b = t / ph.Hg;
            dst = (b * a.Hout + y * ph.DSY + ph.D0Y) * a.Wout + x * ph.DSX + ph.D0X;
        }
        rowdst[tid] = dst;
    }
    int pixbase[AP], sy[AP], sx[AP];
#pragma unroll
    for (int i = 0; i < AP; ++i) {
        const int m = m0 + lrow + 32 * i;
        if (m < M) {
            const int x = m % ph.Wg, t = m / ph.Wg, y = t % ph.Hg, b = t / ph.Hg;
            pixbase[i] = b * a.Hin * a.Win; sy[i] = y * ph.SY; sx[i] = x * ph.SX;
        } else {
            pixbase[i] = 0; sy[i] = -(1 << 20); sx[i] = 0;
        }
    }
    __syncthreads();

    const int cmask = (1 << a.cg_log2) - 1;
    const int nk = ((ph.ntaps << a.cg_log2) + BK - 1) / BK;
    const T* in = reinterpret_cast<const T*>(a.in);
    const T* wp = reinterpret_cast<const T*>(a.wp) + ph.koff;

    u32x4 areg[AP], breg[BP];
    auto load_tile = [&](int kt) {
        const int k = kt * BK + kvec * VEC;
        const int tap = k >> a.cg_log2, c = k & cmask;
        const bool tv = tap < ph.ntaps;
        const int tt = taptab[tap & 15];
        const int dy = tt >> 16, dx = (int)(short)(tt & 0xffff);
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            const int iy = sy[i] + dy, ix = sx[i] + dx;
            const bool ok = tv && (unsigned)iy < (unsigned)a.Hin && (unsigned)ix < (unsigned)a.Win;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (ok) {
                const long off = ((long)(pixbase[i] + iy * a.Win + ix) << a.cg_log2) + c;
                v = *reinterpret_cast<const u32x4*>(in + off);
            }
            areg[i] = v;
        }
#pragma unroll
        for (int i = 0; i < BP; ++i) {
            const int n = ZF ? zf_row<BN, WGN, TN>(blockIdx.y, lrow + 32 * i, a.zL) : n0 + lrow + 32 * i;
            breg[i] = *reinterpret_cast<const u32x4*>(wp + (long)n * a.Ktot + k);
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < AP; ++i)
            *reinterpret_cast<u32x4*>(As + (lrow + 32 * i) * LDS_ROW + kvec * 16) = areg[i];
#pragma unroll
        for (int i = 0; i < BP; ++i)
            *reinterpret_cast<u32x4*>(Bs + (lrow + 32 * i) * LDS_ROW + kvec * 16) = breg[i];
    };

    const int wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int wm0 = (wave / WGN) * 32 * TM, wn0 = (wave % WGN) * 32 * TN;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    if (nk > 0) load_tile(0);
    for (int kt = 0; kt < nk; ++kt) {
        store_tile();
        __syncthreads();
        if (kt + 1 < nk) load_tile(kt + 1);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                Mfma<T>::run(As + (wm0 + 32 * i + r) * LDS_ROW, Bs + (wn0 + 32 * j + r) * LDS_ROW, h, acc[i][j]);
        __syncthreads();
    }

    if constexpr (ZF) {
        // ---- fused encoder head: lane (pixel r, quad g of column block j < TN/2) holds mu in acc[i][j] and log_var in
        // acc[i][j + TN/2]; z goes through the wave's LDS patch so that global memory sees whole channel runs ----
        static_assert(sizeof(T) == 2 && TN % 2 == 0, "fused encoder head: bf16, mu and log_var column blocks");
        constexpr int ZC = 16 * TN;                          // latent channels per wave
        constexpr int SRS = ZC * 2 + 16, VPS = ZC / 8, PPI = 64 / VPS, NI = 32 / PPI;
        static_assert(4 * 32 * SRS <= (BM + BN) * LDS_ROW, "epilogue patches must fit in the tile buffers");
        unsigned char* stg = smem + wave * (32 * SRS);
        const int sp = lane / VPS, sc8 = lane % VPS;
        const int zc0 = blockIdx.y * (BN / 2) + (wave % WGN) * ZC;   // first latent channel of this wave
        T* outb = reinterpret_cast<T*>(a.out);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int row = rowdst[wm0 + 32 * i + r];        // 1x1 unit-stride GEMM: output pixel == GEMM row
            const int rr = row < 0 ? 0 : row;
            const int b = rr / a.zrpi;
            const uint64_t img = a.zkey.index ? (uint64_t)a.zkey.index[b] : (uint64_t)(a.zkey.first + b);
            const long in_img = (long)(rr % a.zrpi) * a.zL;
#pragma unroll
            for (int j = 0; j < TN / 2; ++j) {
                f32x4 e4[4];                                 // (zc0 and in_img are multiples of 8: zL % 128 == 0)
                philox_lane_quads(a.zkey, img, (uint32_t)(in_img + zc0 + 32 * j), h, e4);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int cl = 32 * j + 8 * g + 4 * h;   // channel within the wave's ZC
                    const int nl = wn0 + cl;                 // tile column of mu; log_var sits ZC columns further
                    const f32x4 bm = *reinterpret_cast<const f32x4*>(ssl + BN + nl);
                    const f32x4 bl = *reinterpret_cast<const f32x4*>(ssl + BN + nl + ZC);
                    const f32x4 e = e4[g];
                    bf16x4 qd;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float mu = acc[i][j][4 * g + k] + bm[k], lv = acc[i][j + TN / 2][4 * g + k] + bl[k];
                        qd[k] = (bf16)(e[k] * __expf(0.5f * lv) + mu);
                    }
                    *reinterpret_cast<bf16x4*>(stg + r * SRS + cl * 2) = qd;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int q = 0; q < NI; ++q) {
                const int dp = rowdst[wm0 + 32 * i + q * PPI + sp];
                const u32x4 val = *reinterpret_cast<const u32x4*>(stg + (q * PPI + sp) * SRS + sc8 * 16);
                if (dp >= 0) *reinterpret_cast<u32x4*>(outb + (long)dp * a.zL + zc0 + sc8 * 8) = val;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        return;
    }
    if constexpr (sizeof(T) == 2) {
        if (!a.out_f32) {
            // ---- bf16 output through this wave's private LDS patch (the K-loop's tiles are dead after its last
            // barrier): the MFMA leaves a lane with 4-channel quads of one pixel, global memory wants whole
            // channel runs.  32 pixels x 32*TN channels at a time; residual / old output come in as coalesced
            // 16-byte vectors, are re-read in register layout and added in fp32, so the result is rounded once.
            constexpr int SRS = TN * 64 + 16, VPS = TN * 4, PPI = 64 / VPS, NI = 32 / PPI;
            static_assert(4 * 32 * SRS <= (BM + BN) * LDS_ROW, "epilogue patches must fit in the tile buffers");
            unsigned char* stg = smem + wave * (32 * SRS);
            const int sp = lane / VPS, sc8 = lane % VPS;
            const int nc = n0 + wn0 + sc8 * 8;               // first channel of this lane's 16-byte piece
            const T* resb = reinterpret_cast<const T*>(a.res);
            T* outb = reinterpret_cast<T*>(a.out);
            auto wave_lds_fence = [&]() {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            };
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                long orow[NI];                               // element offset of this lane's piece, or -1
#pragma unroll
                for (int q = 0; q < NI; ++q) {
                    const int dp = rowdst[wm0 + 32 * i + q * PPI + sp];
                    orow[q] = (dp >= 0 && nc < a.NS) ? (long)dp * a.NS + nc : -1;
                }
                u32x4 radd[NI], rold[NI];
                const u32x4 z = {0u, 0u, 0u, 0u};
                if (a.res) {
#pragma unroll
                    for (int q = 0; q < NI; ++q) radd[q] = orow[q] >= 0 ? *reinterpret_cast<const u32x4*>(resb + orow[q]) : z;
                }
                if (a.accumulate) {
#pragma unroll
                    for (int q = 0; q < NI; ++q) rold[q] = orow[q] >= 0 ? *reinterpret_cast<const u32x4*>(outb + orow[q]) : z;
                }
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int nl = wn0 + 32 * j + 8 * g + 4 * h;
                        const f32x4 sc = *reinterpret_cast<const f32x4*>(ssl + nl);
                        const f32x4 sh = *reinterpret_cast<const f32x4*>(ssl + BN + nl);
#pragma unroll
                        for (int k = 0; k < 4; ++k) acc[i][j][4 * g + k] = acc[i][j][4 * g + k] * sc[k] + sh[k];
                    }
                auto add_staged = [&](const u32x4 (&src)[NI]) {
#pragma unroll
                    for (int q = 0; q < NI; ++q)
                        *reinterpret_cast<u32x4*>(stg + (q * PPI + sp) * SRS + sc8 * 16) = src[q];
                    wave_lds_fence();
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const bf16x4 qd = *reinterpret_cast<const bf16x4*>(stg + r * SRS + (32 * j + 8 * g + 4 * h) * 2);
#pragma unroll
                            for (int k = 0; k < 4; ++k) acc[i][j][4 * g + k] += (float)qd[k];
                        }
                    wave_lds_fence();
                };
                if (a.res) add_staged(radd);
                if (a.relu) {
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int e = 0; e < 16; ++e) acc[i][j][e] = fmaxf(acc[i][j][e], 0.f);
                }
                if (a.accumulate) add_staged(rold);
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int n = n0 + wn0 + 32 * j + 8 * g + 4 * h;
                        bf16x4 qd;
#pragma unroll
                        for (int k = 0; k < 4; ++k) qd[k] = (bf16)((n + k < a.N) ? acc[i][j][4 * g + k] : 0.f);   // pad channels stay zero
                        *reinterpret_cast<bf16x4*>(stg + r * SRS + (32 * j + 8 * g + 4 * h) * 2) = qd;
                    }
                wave_lds_fence();
#pragma unroll
                for (int q = 0; q < NI; ++q) {
                    const u32x4 val = *reinterpret_cast<const u32x4*>(stg + (q * PPI + sp) * SRS + sc8 * 16);
                    if (orow[q] >= 0) *reinterpret_cast<u32x4*>(outb + orow[q]) = val;
                }
                wave_lds_fence();
            }
            return;
        }
    }
    // epilogue: lane r owns pixel wm0 + 32i + r; register quad g of tile j holds channels
    // n0 + wn0 + 32j + 8g + 4h .. +3 -> 16-byte (fp32) / 8-byte (bf16) channel runs per store
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int dp = rowdst[wm0 + 32 * i + r];
        if (dp < 0) continue;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + wn0 + 32 * j + 8 * g + 4 * h;
                if (n >= a.NS) continue;
                const long o = (long)dp * a.NS + n;
                float v[4];
                const f32x4 sc = *reinterpret_cast<const f32x4*>(ssl + (n - n0));
                const f32x4 sh = *reinterpret_cast<const f32x4*>(ssl + BN + (n - n0));
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = acc[i][j][4 * g + k] * sc[k] + sh[k];
                if (a.res) {
                    float rv[4];
                    Act<T>::load4(reinterpret_cast<const T*>(a.res) + o, rv);
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] += rv[k];
                }
                if (a.relu) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], 0.f);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) if (n + k >= a.N) v[k] = 0.f;
                if (a.out_f32) {
                    float* op = reinterpret_cast<float*>(a.out);
                    if (a.accumulate) {
                        float ov[4];
                        Act<float>::load4(op + o, ov);
#pragma unroll
                        for (int k = 0; k < 4; ++k) v[k] += ov[k];
                    }
                    Act<float>::store4(op + o, v);
                } else {
                    T* op = reinterpret_cast<T*>(a.out);
                    if (a.accumulate) {
                        float ov[4];
                        Act<T>::load4(op + o, ov);
#pragma unroll
                        for (int k = 0; k < 4; ++k) v[k] += ov[k];
                    }
                    Act<T>::store4(op + o, v);
                }
            }
        }
    }
}

template <typename T>
static int launch_igemm(const GemmPlan& P, const void* in, const void* wp, void* out, const float* scale,
                        const float* shift, const void* res, int relu, int accumulate, int out_f32,
                        hipStream_t st) {
    GemmArgs a;
    a.in = in; a.wp = wp; a.out = out; a.scale = scale; a.shift = shift; a.res = res;
    a.B = P.B; a.Hin = P.Hin; a.Win = P.Win; a.cg_log2 = P.cg_log2;
    // an fp32 output keeps the fp32 channel padding (>= 4) whatever the activation dtype is
    a.Hout = P.Hout; a.Wout = P.Wout; a.N = P.N; a.NS = out_f32 ? act_channels(P.N) : P.NS; a.Ktot = P.Ktot;
    a.relu = relu; a.accumulate = accumulate; a.out_f32 = out_f32;
    a.zL = 0; a.zrpi = 1; a.zkey = PhiloxKey{nullptr, 0, 0u, 0u, 0u};
    int maxM = 0;
    for (int i = 0; i < 4; ++i) a.ph[i] = P.ph[i];
    for (int i = 0; i < P.nphase; ++i) {
        long M = (long)P.B * P.ph[i].Hg * P.ph[i].Wg;
        if (M > 0x7fffffffL / 2) DVAE_FAIL(DVAE_EINVAL, "too many output pixels for 32-bit indexing");
        if ((int)M > maxM) maxM = (int)M;
    }
    if ((long)P.B * P.Hin * P.Win * P.CgS > 0x7fffffffL || (long)P.B * P.Hout * P.Wout * P.NS > 0x7fffffffL)
        DVAE_FAIL(DVAE_EINVAL, "tensor exceeds 2^31 elements");
    if (maxM == 0) return DVAE_OK;
    constexpr int BM = 128;
    dim3 block(256);
    if (P.Npad % 256 == 0 && sizeof(T) == 2) {
        // wide outputs (mu|log_var: N = 2L): a 128 x 256 tile halves the re-gathering of the A rows
        dim3 grid((maxM + BM - 1) / BM, P.Npad / 256, P.nphase);
        hipLaunchKernelGGL((igemm_kernel<T, 2, 4, 2, 2>), grid, block, 0, st, a);
    } else if (P.Npad % 128 == 0) {
        dim3 grid((maxM + BM - 1) / BM, P.Npad / 128, P.nphase);
        hipLaunchKernelGGL((igemm_kernel<T, 2, 2, 2, 2>), grid, block, 0, st, a);
    } else if (P.Npad % 64 == 0) {
        dim3 grid((maxM + BM - 1) / BM, P.Npad / 64, P.nphase);
        hipLaunchKernelGGL((igemm_kernel<T, 2, 1, 2, 2>), grid, block, 0, st, a);
    } else {
        dim3 grid((maxM + BM - 1) / BM, P.Npad / 32, P.nphase);
        hipLaunchKernelGGL((igemm_kernel<T, 1, 1, 4, 1>), grid, block, 0, st, a);
    }
    DVAE_CHECK_LAUNCH("igemm");
    return DVAE_OK;
}

// Fused encoder head: z = eps * exp(log_var / 2) + mu for a 1x1 unit-stride conv whose 2L outputs are [mu | log_var].
static bool zf_applicable(const GemmPlan& P, int L) {
    if (P.nphase != 1 || P.N != 2 * L || L % 128) return false;
    const Phase& ph = P.ph[0];
    return ph.ntaps == 1 && ph.SY == 1 && ph.SX == 1 && ph.DSY == 1 && ph.DSX == 1 && !ph.D0Y && !ph.D0X &&
           ph.dy[0] == 0 && ph.dx[0] == 0 && P.Hin == P.Hout && P.Win == P.Wout && ph.Hg == P.Hout && ph.Wg == P.Wout;
}

static int launch_igemm_zf(const GemmPlan& P, const void* in, const void* wp, const float* bias, void* z, int L,
                           const PhiloxKey& key, hipStream_t st) {
    if (!zf_applicable(P, L)) DVAE_FAIL(DVAE_EINVAL, "fc_reparam: needs a 1x1 unit-stride conv with 2L outputs, L a multiple of 128");
    GemmArgs a;
    a.in = in; a.wp = wp; a.out = z; a.scale = nullptr; a.shift = bias; a.res = nullptr;
    a.B = P.B; a.Hin = P.Hin; a.Win = P.Win; a.cg_log2 = P.cg_log2;
    a.Hout = P.Hout; a.Wout = P.Wout; a.N = P.N; a.NS = P.NS; a.Ktot = P.Ktot;
    a.relu = 0; a.accumulate = 0; a.out_f32 = 0;
    for (int i = 0; i < 4; ++i) a.ph[i] = P.ph[i];
    a.zL = L; a.zrpi = P.Hout * P.Wout; a.zkey = key;
    const long M = (long)P.B * P.Hout * P.Wout;
    if (M > 0x7fffffffL / 2 || M * P.CgS > 0x7fffffffL || M * L > 0x7fffffffL) DVAE_FAIL(DVAE_EINVAL, "tensor exceeds 2^31 elements");
    if (M == 0) return DVAE_OK;
    // 128 x 128 tiles (64 latent channels per workgroup): 2 workgroups per CU, so one's Philox epilogue overlaps the
    // other's loads and MFMAs; the 128 x 256 tile needs > 256 registers and runs alone
    dim3 grid((unsigned)((M + 127) / 128), P.N / 128, 1), block(256);
    hipLaunchKernelGGL((igemm_kernel<bf16, 2, 2, 2, 2, true>), grid, block, 0, st, a);
    DVAE_CHECK_LAUNCH("igemm_zf");
    return DVAE_OK;
}

// ------------------------------------------------------------------------------------------------
// weight gradient
// ------------------------------------------------------------------------------------------------
struct WgradPlan {
    int B;
    int Hx, Wx, Cx, CxS, cx_log2;   // gathered tensor
    int Hg, Wg, Cd, CdS;            // dense tensor (rows m)
    int SY, SX, ntaps;
    signed char dy[16], dx[16];
    int woff[16];
    int sX, sD;                     // reference weight index = cd*sD + cx*sX + woff
    int M, S, chunk_rows, Kpad, CdPad;
    int bias_from_gathered;         // bias gradient = column sums of the gathered (1) or dense (0) tensor
    unsigned bias_taps;             // gathered case: the taps that together read every pixel exactly once (0: none known)
    int halo, halo_tiles, halo_per; // all-taps-per-tile kernel (wgrad_halo.hpp): S workgroups of halo_per 4 x 32 tiles
    int big;                        // large-tile kernel (wgrad_big.hpp): every other bf16 launch
};

// tile of the large-tile kernel for a slab of Kpad x CdPad: {K width, Cd width}, always 8 waves of 64 x 64
struct WgradBigTile { int tk, tc; };
static inline WgradBigTile wgrad_big_tile(int Kpad, int CdPad) {
    (void)Kpad; (void)CdPad;
    return {256, 128};              // (512 x 64 and 128 x 256 were built too: narrow slabs are not bandwidth-bound, see below)
}

static int build_wgrad_plan(const dvae_conv_desc& d, int dtype, WgradPlan& W) {
    GemmPlan P;
    int rc = build_plan(d, DVAE_OP_FWD, dtype, P);
    if (rc) return rc;
    memset(&W, 0, sizeof(W));
    const int kk = d.kh * d.kw, p = d.pad, s = d.stride;
    const int Ho = d.transposed ? P.Hout : P.Hout, Wo = P.Wout;   // module output dims
    W.B = d.B;
    if (!d.transposed) {   // X = x (gathered with stride s), D = dy
        W.Hx = d.H; W.Wx = d.W; W.Cx = d.Cin; W.Hg = Ho; W.Wg = Wo; W.Cd = d.Cout;
        W.sX = kk; W.sD = d.Cin * kk; W.bias_from_gathered = 0;
    } else {               // X = dy (gathered with stride s), D = x; weight [Cin][Cout][kh][kw]
        W.Hx = Ho; W.Wx = Wo; W.Cx = d.Cout; W.Hg = d.H; W.Wg = d.W; W.Cd = d.Cin;
        W.sX = kk; W.sD = d.Cout * kk; W.bias_from_gathered = 1;
    }
    const int minc = dtype == DVAE_BF16 ? 8 : 4;
    W.CxS = act_channels(W.Cx); if (W.CxS < minc) W.CxS = minc;
    W.CdS = act_channels(W.Cd); if (W.CdS < minc) W.CdS = minc;
    W.cx_log2 = ilog2_exact(W.CxS);
    W.SY = s; W.SX = s; W.ntaps = kk;
    for (int ky = 0; ky < d.kh; ++ky)
        for (int kx = 0; kx < d.kw; ++kx) {
            int t = ky * d.kw + kx;
            W.dy[t] = (signed char)(ky - p); W.dx[t] = (signed char)(kx - p); W.woff[t] = t;
        }
    W.M = d.B * W.Hg * W.Wg;
    if (W.Hx == W.Hg * s && W.Wx == W.Wg * s) {
        int n = 0;
        for (int t = 0; t < kk; ++t)
            if (W.dy[t] >= 0 && W.dy[t] < s && W.dx[t] >= 0 && W.dx[t] < s) { W.bias_taps |= 1u << t; ++n; }
        if (n != s * s) W.bias_taps = 0;
    }
    W.Kpad = round_up(kk * W.CxS, 64);
    W.CdPad = round_up(W.CdS, 64);
    const int tiles = (W.Kpad / 64) * (W.CdPad / 64);
    int S = 768 / tiles; if (S < 1) S = 1;
    int chunk = round_up((W.M + S - 1) / S, 64); if (chunk < 64) chunk = 64;
    // Large tiles (wgrad_big.hpp) where the slab is wide both ways -- the 128-channel 3x3 / 4x4 shapes: 380 -> 233, 204 ->
    // 152 and 275 -> 254 us per 1024 images.  Narrow slabs read little to begin with and are bound by the per-row
    // address arithmetic, which the large-tile kernel spreads over fewer waves: they stay on the 64 x 64 tiles.  So do
    // small launches (batch 128 of 16 x 16 maps is 32 768 rows: 16 stages per large tile, 40 against 28 us).
    if (dtype == DVAE_BF16 && W.Kpad >= 512 && W.CdPad >= 128 && W.M >= 100000) {
        // one workgroup per CU where the pixels allow it, at least 16 stages of 64 rows each (a workgroup leaves a
        // whole tile per chunk for the reduction to read back)
        const WgradBigTile t = wgrad_big_tile(W.Kpad, W.CdPad);
        const int cols = ((W.Kpad + t.tk - 1) / t.tk) * ((W.CdPad + t.tc - 1) / t.tc);
        S = 256 / cols;
        if (S > W.M / 1024) S = W.M / 1024;
        if (S < 1) S = 1;
        chunk = round_up((W.M + S - 1) / S, 64);
        W.big = 1;
    }
    W.chunk_rows = chunk;
    W.S = (W.M + chunk - 1) / chunk;
    // 3x3 unit-stride in bf16 with 64 x 64, 32 x 64, 8 x 64 or 8 x 32 stored channels (gathered x dense): one staged tile
    // serves all nine taps (wgrad_halo.hpp).  Each workgroup leaves whole slabs for the reduction to read back (147 KB for
    // 64 x 64), so at least 8 tiles each once there are enough (batch 128: one workgroup per two CUs), at most one
    // workgroup per CU.
    const bool halo_cc = (W.CxS == 64 && W.CdS == 64) || (W.CxS == 32 && W.CdS == 64) || (W.CxS == 8 && W.CdS == 64) ||
                         (W.CxS == 8 && W.CdS == 32);
    bool halo = dtype == DVAE_BF16 && kk == 9 && s == 1 && p == 1 && halo_cc && W.Hx == W.Hg && W.Wx == W.Wg &&
                W.Wg % 32 == 0 && W.Hg % 4 == 0 && (long)W.M * 64 <= 0x7fffffffL;
    if (halo) {
        W.halo = 1; W.big = 0;
        W.halo_tiles = d.B * (W.Hg / 4) * (W.Wg / 32);
        int g = W.halo_tiles / 8;
        if (g > 256) g = 256;
        if (g < 1) g = 1;
        W.halo_per = (W.halo_tiles + g - 1) / g;
        W.halo_per += W.halo_per & 1;                  // the kernel walks its tiles two at a time
        W.S = (W.halo_tiles + W.halo_per - 1) / W.halo_per;
    }
    return DVAE_OK;
}

constexpr int COLSUM_BLOCKS = 128;

// the bias gradient rides on the weight-gradient launches (column sums of the tiles they stage anyway) unless the
// bias comes from the gathered tensor and no set of taps reads each of its pixels exactly once
static bool wgrad_bias_fused(const WgradPlan& W) {
    return !W.bias_from_gathered || (W.bias_taps != 0 && W.SY * W.SX <= 4);
}
static int wgrad_bias_width(const WgradPlan& W) { return W.bias_from_gathered ? W.Kpad : W.CdPad; }

static size_t wgrad_workspace_floats(const WgradPlan& W) {
    const int cs = W.bias_from_gathered ? W.CxS : W.CdS;
    const size_t fused = (size_t)W.S * wgrad_bias_width(W), split = (size_t)COLSUM_BLOCKS * cs;
    return (size_t)W.S * W.Kpad * W.CdPad + (fused > split ? fused : split);
}

struct WgradArgs {
    const void* xg;
    const void* d;
    float* ws;
    int B, Hx, Wx, cx_log2, Hg, Wg, SY, SX, CdS, ntaps;
    signed char dy[16], dx[16];
    int M, chunk_rows, Kpad, CdPad;
    int wg_log2, hg_log2;          // log2 of Wg / Hg when both are powers of two, else -1
    // bias gradient: 0 none; 1 column sums of d (by the workgroups of K slab 0); 2 column sums of the gathered
    // tensor over the taps in bias_taps (by the workgroups of Cd slab 0).  Partials go to bpart[chunk][width].
    int bias_mode;
    unsigned bias_taps;
    float* bpart;
};

// Column sums of the rows a workgroup staged, from the per-thread sums bs[V] (V channels at column kv * V, RL row
// lanes): reduced over the row lanes in LDS in a fixed order and written as one 64-wide partial row.
template <int V, int RL>
__device__ inline void wgrad_bias_flush(const float* bs, float* red, int kv, int rl, float* out) {
#pragma unroll
    for (int i = 0; i < V; ++i) red[rl * 64 + kv * V + i] = bs[i];
    __syncthreads();
    if (threadIdx.x < 64) {
        float t = 0.f;
        for (int q = 0; q < RL; ++q) t += red[q * 64 + threadIdx.x];
        out[threadIdx.x] = t;
    }
}

// fp32: one 64(K) x 64(Cd) slab tile per workgroup, contraction over a chunk of pixels,
// 32 pixels per LDS stage, v_mfma_f32_32x32x2_f32 with A[i = K column][k = pixel], B[k = pixel][j = cd].
__global__ __launch_bounds__(256) void wgrad_kernel_f32(const WgradArgs a) {
    __shared__ __attribute__((aligned(16))) float As[32 * 64];
    __shared__ __attribute__((aligned(16))) float Ds[32 * 64];
    const int tid = threadIdx.x, kv = tid & 15, rl = tid >> 4;
    const int kbase = blockIdx.y * 64, cdbase = blockIdx.z * 64;
    const int k = kbase + kv * 4;
    const int tap = k >> a.cx_log2, c = k & ((1 << a.cx_log2) - 1);
    const bool tv = tap < a.ntaps;
    const int dy = a.dy[tap & 15], dx = a.dx[tap & 15];
    const int cd = cdbase + kv * 4;
    const bool cv = cd < a.CdS;
    const int mbeg = blockIdx.x * a.chunk_rows;
    const int mend = min(a.M, mbeg + a.chunk_rows);
    const float* xg = reinterpret_cast<const float*>(a.xg);
    const float* dd = reinterpret_cast<const float*>(a.d);

    f32x4 xr[2], dr[2];
    auto load = [&](int mb) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = mb + rl + 16 * i;
            f32x4 xv = {0.f, 0.f, 0.f, 0.f}, dv = {0.f, 0.f, 0.f, 0.f};
            if (m < mend) {
                int x, y, b;
                if (a.wg_log2 >= 0) {                 // power-of-two maps: no integer division per row
                    x = m & (a.Wg - 1); y = (m >> a.wg_log2) & (a.Hg - 1); b = m >> (a.wg_log2 + a.hg_log2);
                } else {
                    const int t = m / a.Wg;
                    x = m % a.Wg; y = t % a.Hg; b = t / a.Hg;
                }
                const int iy = y * a.SY + dy, ix = x * a.SX + dx;
                if (tv && (unsigned)iy < (unsigned)a.Hx && (unsigned)ix < (unsigned)a.Wx)
                    xv = *reinterpret_cast<const f32x4*>(xg + (((long)((b * a.Hx + iy) * a.Wx + ix)) << a.cx_log2) + c);
                if (cv) dv = *reinterpret_cast<const f32x4*>(dd + (long)m * a.CdS + cd);
            }
            xr[i] = xv; dr[i] = dv;
        }
    };
    const int wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int wi = wave >> 1, wj = wave & 1;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;

    const bool bias_d = a.bias_mode == 1 && blockIdx.y == 0, bias_x = a.bias_mode == 2 && blockIdx.z == 0;
    float bs[4] = {0.f, 0.f, 0.f, 0.f};
    if (mbeg < mend) load(mbeg);
    for (int mb = mbeg; mb < mend; mb += 32) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *reinterpret_cast<f32x4*>(&As[(rl + 16 * i) * 64 + kv * 4]) = xr[i];
            *reinterpret_cast<f32x4*>(&Ds[(rl + 16 * i) * 64 + kv * 4]) = dr[i];
            if (bias_d || bias_x) {
                const f32x4 v = bias_d ? dr[i] : xr[i];
#pragma unroll
                for (int c = 0; c < 4; ++c) bs[c] += v[c];
            }
        }
        __syncthreads();
        if (mb + 32 < mend) load(mb + 32);
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const float av = As[(2 * s + h) * 64 + wi * 32 + r];
            const float bv = Ds[(2 * s + h) * 64 + wj * 32 + r];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    float* ws = a.ws + ((long)blockIdx.x * a.Kpad + kbase + wi * 32) * a.CdPad + cdbase + wj * 32 + r;
#pragma unroll
    for (int e = 0; e < 16; ++e) ws[(long)((e & 3) + 8 * (e >> 2) + 4 * h) * a.CdPad] = acc[e];
    if (bias_d) wgrad_bias_flush<4, 16>(bs, As, kv, rl, a.bpart + (long)blockIdx.x * a.CdPad + cdbase);
    if (bias_x) wgrad_bias_flush<4, 16>(bs, As, kv, rl, a.bpart + (long)blockIdx.x * a.Kpad + kbase);
}

// bf16: same decomposition; 64 pixels per LDS stage in [pixel][channel] order (192-byte rows: the four
// rows of a transposed read land on disjoint bank quarters), operands fetched with ds_read_b64_tr_b16 so
// that each lane receives 4 consecutive pixels of one channel = the k-contiguous MFMA fragment layout.
__device__ inline bf16x8 tr_frag(const unsigned char* tile, int mrow0, int col0, int lane) {
    constexpr int RS = 192;
    const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3, h = g >> 1;
    const unsigned char* base = tile + (mrow0 + 8 * h + q) * RS + (col0 + 16 * (g & 1) + 4 * p) * 2;
    auto lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(base));
    auto hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(base + 4 * RS));
    bf16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return r;
}

__global__ __launch_bounds__(256) void wgrad_kernel_bf16(const WgradArgs a) {
    constexpr int RS = 192, MS = 64;
    __shared__ __attribute__((aligned(16))) unsigned char As[MS * RS];
    __shared__ __attribute__((aligned(16))) unsigned char Ds[MS * RS];
    const int tid = threadIdx.x, kv = tid & 7, rl = tid >> 3;      // 8 x 16-byte vectors per 64-channel row
    const int kbase = blockIdx.y * 64, cdbase = blockIdx.z * 64;
    const int k = kbase + kv * 8;
    const int tap = k >> a.cx_log2, c = k & ((1 << a.cx_log2) - 1);
    const bool tv = tap < a.ntaps;
    const int dy = a.dy[tap & 15], dx = a.dx[tap & 15];
    const int cd = cdbase + kv * 8;
    const bool cv = cd < a.CdS;
    const int mbeg = blockIdx.x * a.chunk_rows;
    const int mend = min(a.M, mbeg + a.chunk_rows);
    const bf16* xg = reinterpret_cast<const bf16*>(a.xg);
    const bf16* dd = reinterpret_cast<const bf16*>(a.d);
    u32x4 xr[2], dr[2];
    auto load = [&](int mb) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = mb + rl + 32 * i;
            u32x4 xv = {0u, 0u, 0u, 0u}, dv = {0u, 0u, 0u, 0u};
            if (m < mend) {
                int x, y, b;
                if (a.wg_log2 >= 0) {                 // power-of-two maps: no integer division per row
                    x = m & (a.Wg - 1); y = (m >> a.wg_log2) & (a.Hg - 1); b = m >> (a.wg_log2 + a.hg_log2);
                } else {
                    const int t = m / a.Wg;
                    x = m % a.Wg; y = t % a.Hg; b = t / a.Hg;
                }
                const int iy = y * a.SY + dy, ix = x * a.SX + dx;
                if (tv && (unsigned)iy < (unsigned)a.Hx && (unsigned)ix < (unsigned)a.Wx)
                    xv = *reinterpret_cast<const u32x4*>(xg + (((long)((b * a.Hx + iy) * a.Wx + ix)) << a.cx_log2) + c);
                if (cv) dv = *reinterpret_cast<const u32x4*>(dd + (long)m * a.CdS + cd);
            }
            xr[i] = xv; dr[i] = dv;
        }
    };
    const int wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int wi = wave >> 1, wj = wave & 1;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const bool bias_d = a.bias_mode == 1 && blockIdx.y == 0, bias_x = a.bias_mode == 2 && blockIdx.z == 0;
    float bs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (mbeg < mend) load(mbeg);
    for (int mb = mbeg; mb < mend; mb += MS) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *reinterpret_cast<u32x4*>(As + (rl + 32 * i) * RS + kv * 16) = xr[i];
            *reinterpret_cast<u32x4*>(Ds + (rl + 32 * i) * RS + kv * 16) = dr[i];
            if (bias_d || bias_x) {
                const u32x4 v = bias_d ? dr[i] : xr[i];
#pragma unroll
                for (int c = 0; c < 4; ++c) {      // a bf16 is the high half of its fp32
                    bs[2 * c] += __uint_as_float(v[c] << 16);
                    bs[2 * c + 1] += __uint_as_float(v[c] & 0xffff0000u);
                }
            }
        }
        __syncthreads();
        if (mb + MS < mend) load(mb + MS);
#pragma unroll
        for (int s = 0; s < MS / 16; ++s) {
            const bf16x8 av = tr_frag(As, 16 * s, wi * 32, lane);
            const bf16x8 bv = tr_frag(Ds, 16 * s, wj * 32, lane);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    float* ws = a.ws + ((long)blockIdx.x * a.Kpad + kbase + wi * 32) * a.CdPad + cdbase + wj * 32 + r;
#pragma unroll
    for (int e = 0; e < 16; ++e) ws[(long)((e & 3) + 8 * (e >> 2) + 4 * h) * a.CdPad] = acc[e];
    float* red = reinterpret_cast<float*>(As);      // 32 x 64 floats; every wave is past the loop's last barrier
    if (bias_d) wgrad_bias_flush<8, 32>(bs, red, kv, rl, a.bpart + (long)blockIdx.x * a.CdPad + cdbase);
    if (bias_x) wgrad_bias_flush<8, 32>(bs, red, kv, rl, a.bpart + (long)blockIdx.x * a.Kpad + kbase);
}

struct WreduceArgs {
    const float* ws;
    float* dw;
    int S, Kpad, CdPad, Cx, cx_log2, Cd, ntaps, sX, sD;
    int woff[16];
    // bias partials [S][bwidth] -> db[Cb]; the workgroups from nmain on do this part
    const float* bpart;
    float* db;
    int bias_mode, bwidth, Cb, cxs, nmain;
    int nbt, btap[4];                // gathered case: the (at most 4) taps whose column sums add up to the bias gradient
    int nblocks;                     // workgroups of this job: nmain + the bias blocks
};

__device__ __forceinline__ void wgrad_reduce_block(const WreduceArgs& a, const int bx) {
    __shared__ float part[4][64];
    const long total = (long)a.Kpad * a.CdPad;
    const int el = threadIdx.x & 63, sl = threadIdx.x >> 6;
    if (bx >= a.nmain) {                // bias gradient: 16 channels per workgroup, 16 lanes split S
        const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
        const int c = (bx - a.nmain) * 16 + cl;
        float s[4] = {0.f, 0.f, 0.f, 0.f};
        if (c < a.Cb) {
            if (a.bias_mode == 1) {
                for (int i = rl; i < a.S; i += 64) {
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (i + 16 * u < a.S) s[u] += a.bpart[(long)(i + 16 * u) * a.bwidth + c];
                }
            } else {
                for (int i = rl; i < a.S; i += 16) {
                    const float* row = a.bpart + (long)i * a.bwidth + c;
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (u < a.nbt) s[u] += row[a.btap[u] * a.cxs];
                }
            }
        }
        float* red = &part[0][0];
        red[rl * 16 + cl] = (s[0] + s[1]) + (s[2] + s[3]);
        __syncthreads();
        if (rl == 0 && c < a.Cb) {
            float t = 0.f;
            for (int q = 0; q < 16; ++q) t += red[q * 16 + cl];
            a.db[c] = t;
        }
        return;
    }
    // main part: 256 consecutive slab elements per block as 16-byte vectors, the 4 waves split the S slabs
    // (interleaved), four loads of each in flight; partial sums combine in a fixed order (deterministic)
    const long e = (long)bx * 256 + el * 4;
    f32x4 acc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (e < total) {
        for (int i = sl; i < a.S; i += 16) {
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ii = i + 4 * u;
                v[u] = *reinterpret_cast<const f32x4*>(a.ws + (long)(ii < a.S ? ii : i) * total + e);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (i + 4 * u < a.S) acc[u] += v[u];
        }
    }
    __shared__ f32x4 part4[4][64];
    part4[sl][el] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    __syncthreads();
    if (sl != 0 || e >= total) return;
    const f32x4 t = (part4[0][el] + part4[1][el]) + (part4[2][el] + part4[3][el]);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const long eq = e + q;
        const int k = (int)(eq / a.CdPad), cd = (int)(eq % a.CdPad);
        const int tap = k >> a.cx_log2, c = k & ((1 << a.cx_log2) - 1);
        if (tap >= a.ntaps || c >= a.Cx || cd >= a.Cd) continue;
        a.dw[(long)cd * a.sD + (long)c * a.sX + a.woff[tap]] = t[q];
    }
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const WreduceArgs a) { wgrad_reduce_block(a, (int)blockIdx.x); }

// Every conv's slab (and bias) reduction of a backward pass in one launch: blockIdx.y selects a job record in device
// memory, blockIdx.x that job's block; jobs with fewer blocks than the grid is wide let the surplus exit.
__global__ __launch_bounds__(256) void wgrad_reduce_batch_kernel(const WreduceArgs* jobs) {
    const WreduceArgs& a = jobs[blockIdx.y];
    if ((int)blockIdx.x >= a.nblocks) return;
    wgrad_reduce_block(a, (int)blockIdx.x);
}

// column sums of an [rows][CS] activation tensor (bias gradients), two deterministic stages
template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* x, float* partial, int rows, int CS) {
    constexpr int VEC = Act<T>::VEC;
    __shared__ float red[256 * VEC];
    const int CV = CS / VEC, RL = 256 / CV;
    const int cvi = threadIdx.x % CV, rl = threadIdx.x / CV;
    float s[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) s[i] = 0.f;
    if (rl < RL) {
        for (long row = (long)blockIdx.x * RL + rl; row < rows; row += (long)gridDim.x * RL) {
            float v[VEC];
            Act<T>::load(x + row * CS + cvi * VEC, v);
#pragma unroll
            for (int i = 0; i < VEC; ++i) s[i] += v[i];
        }
    }
#pragma unroll
    for (int i = 0; i < VEC; ++i) red[threadIdx.x * VEC + i] = s[i];
    __syncthreads();
    for (int ch = threadIdx.x; ch < CS; ch += 256) {
        const int cv2 = ch / VEC, i = ch % VEC;
        float t = 0.f;
        for (int q = 0; q < RL; ++q) t += red[(q * CV + cv2) * VEC + i];
        partial[(long)blockIdx.x * CS + ch] = t;
    }
}

__global__ __launch_bounds__(256) void colsum_reduce_kernel(const float* partial, float* out, int nblk, int CS, int C) {
    __shared__ float part[4][64];
    const int cl = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    float s0 = 0.f, s1 = 0.f;
    if (c < C) {
        int i = sl;
        for (; i + 4 < nblk; i += 8) { s0 += partial[(long)i * CS + c]; s1 += partial[(long)(i + 4) * CS + c]; }
        if (i < nblk) s0 += partial[(long)i * CS + c];
    }
    part[sl][cl] = s0 + s1;
    __syncthreads();
    if (sl == 0 && c < C) out[c] = (part[0][cl] + part[1][cl]) + (part[2][cl] + part[3][cl]);
}

}  // namespace dvae
