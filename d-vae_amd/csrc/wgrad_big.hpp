// bf16 weight gradient with large slab tiles: dW[k][cd] = sum over pixels m of X[src(m, tap)][c] * D[m][cd], k = (tap, c).
//
// The 64 x 64-tile kernel of igemm.hpp (wgrad_kernel_bf16) re-reads its pixel rows once per tile column: for the
// 128 -> 128 3x3 conv of layer2 that is 2.4 GB through L2 for 134 MB of operands, and the launch runs at the rate L2
// delivers them (6 TB/s), 0.2 PFLOP/s.  Here a workgroup of 8 waves owns a tile of (64 KW) x (64 CW) slab elements -- 256
// x 128 as instantiated -- so a launch has 3 to 5 times fewer tile columns and reads that much less; a wave owns
// 64 x 64 of the tile (four accumulators: 1 KB of LDS reads per MFMA instead of 2).  MS pixel rows per stage, two LDS
// stages, and a ring of three stages of rows in flight in registers: every row is fetched three stages ahead, written to
// the idle LDS buffer one stage ahead, and a stage costs one barrier.  Loads are unconditional (clamped address, rows that
// do not exist are zeroed on the way to LDS): a load inside a branch is waited for where the branch ends.
// Slab layout, bias partial sums and the reduction that follows are those of wgrad_kernel_bf16.
#pragma once
#include "igemm.hpp"

namespace dvae {

__device__ inline bf16x8 tr_frag_rs(const unsigned char* tile, int rs, int mrow0, int col0, int lane) {
    const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3, h = g >> 1;
    const unsigned char* base = tile + (mrow0 + 8 * h + q) * rs + (col0 + 16 * (g & 1) + 4 * p) * 2;
    auto lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(base));
    auto hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(base + 4 * rs));
    bf16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return r;
}

template <int KW, int CW, int MS>
static constexpr int wgrad_big_lds() { return 2 * MS * ((64 * KW * 2 + 64) + (64 * CW * 2 + 64)); }

template <int KW, int CW, int MS>
__global__ __launch_bounds__(512) void wgrad_big_bf16_kernel(const WgradArgs a) {
    constexpr int TK = 64 * KW, TC = 64 * CW;
    constexpr int RSA = TK * 2 + 64, RSD = TC * 2 + 64;   // 64 bytes of padding: the four rows of a transposed read land on
                                                          // disjoint quarters of the banks
    constexpr int VA = TK / 8, VD = TC / 8;               // 16-byte vectors per staged row
    constexpr int RLA = 512 / VA, RLD = 512 / VD;         // row lanes
    constexpr int NA = MS / RLA;                          // gathered vectors per thread and stage
    constexpr int ND = MS >= RLD ? MS / RLD : 1;          // d vectors per thread and stage (MS < RLD: threads tid < MS * VD only)
    constexpr int RLDE = MS >= RLD ? RLD : MS;            // row lanes of the d side that hold rows
    constexpr int RING = 3;
    static_assert(KW * CW == 8 && MS % 16 == 0 && MS % RLA == 0 && NA >= 1 && NA + ND <= 8, "tile / stage split");
    static_assert(RLDE * TC * 4 <= wgrad_big_lds<KW, CW, MS>() && RLA * TK * 4 <= wgrad_big_lds<KW, CW, MS>(), "bias scratch");
    extern __shared__ __attribute__((aligned(16))) unsigned char wb_smem[];
    unsigned char* As = wb_smem;                          // [2][MS][RSA]
    unsigned char* Ds = wb_smem + 2 * MS * RSA;           // [2][MS][RSD]
    const int tid = threadIdx.x;
    const int kv = tid % VA, rl = tid / VA;               // gathered side: rows rl + RLA i
    const int kvd = tid % VD, rld = tid / VD;             // d side: rows rld + RLD i
    const bool d_thread = rld < RLDE;
    const int kbase = blockIdx.y * TK, cdbase = blockIdx.z * TC;
    const int k = kbase + kv * 8;
    const int tap = k >> a.cx_log2, c = k & ((1 << a.cx_log2) - 1);
    const bool tv = tap < a.ntaps;
    const int dy = a.dy[tap & 15], dx = a.dx[tap & 15];
    const int cd = cdbase + kvd * 8;
    const bool cv = cd < a.CdS && d_thread;
    const int mbeg = blockIdx.x * a.chunk_rows;
    const int mend = min(a.M, mbeg + a.chunk_rows);
    const bf16* xg = reinterpret_cast<const bf16*>(a.xg);
    const bf16* dd = reinterpret_cast<const bf16*>(a.d);
    u32x4 xr[RING][NA], dr[RING][ND];
    unsigned live[RING];                                  // bits 0..NA-1: gathered rows, bits 8..8+ND-1: d rows
    auto load = [&](int slot, int mb) {                   // slot: compile-time at every call site
        unsigned lv = 0u;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int m = mb + rl + RLA * i;
            const int mc = m < mend ? m : mbeg;
            int x, y, b;
            if (a.wg_log2 >= 0) {                         // power-of-two maps: no integer division per row
                x = mc & (a.Wg - 1); y = (mc >> a.wg_log2) & (a.Hg - 1); b = mc >> (a.wg_log2 + a.hg_log2);
            } else {
                const int t = mc / a.Wg;
                x = mc % a.Wg; y = t % a.Hg; b = t / a.Hg;
            }
            const int iy = y * a.SY + dy, ix = x * a.SX + dx;
            const bool ok = m < mend && tv && (unsigned)iy < (unsigned)a.Hx && (unsigned)ix < (unsigned)a.Wx;
            const int py = ok ? iy : 0, px = ok ? ix : 0;
            xr[slot][i] = *reinterpret_cast<const u32x4*>(xg + (((long)((b * a.Hx + py) * a.Wx + px)) << a.cx_log2) + (tv ? c : 0));
            lv |= (ok ? 1u : 0u) << i;
        }
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const int m = mb + rld + RLD * i;
            const bool ok = m < mend && cv;
            dr[slot][i] = *reinterpret_cast<const u32x4*>(dd + (long)(ok ? m : mbeg) * a.CdS + (cv ? cd : 0));
            lv |= (ok ? 1u : 0u) << (8 + i);
        }
        live[slot] = lv;
    };
    const bool bias_d = a.bias_mode == 1 && blockIdx.y == 0, bias_x = a.bias_mode == 2 && blockIdx.z == 0;
    float bs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto stage = [&](int slot, int buf) {                 // the fetched rows -> LDS stage buf (+ their column sums)
        const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            xr[slot][i] = (live[slot] >> i & 1u) ? xr[slot][i] : z;
            *reinterpret_cast<u32x4*>(As + (buf * MS + rl + RLA * i) * RSA + kv * 16) = xr[slot][i];
        }
        if (d_thread) {
#pragma unroll
            for (int i = 0; i < ND; ++i) {
                dr[slot][i] = (live[slot] >> (8 + i) & 1u) ? dr[slot][i] : z;
                *reinterpret_cast<u32x4*>(Ds + (buf * MS + rld + RLD * i) * RSD + kvd * 16) = dr[slot][i];
            }
        }
        if (bias_d && d_thread) {
#pragma unroll
            for (int i = 0; i < ND; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {             // a bf16 is the high half of its fp32
                    bs[2 * q] += __uint_as_float(dr[slot][i][q] << 16);
                    bs[2 * q + 1] += __uint_as_float(dr[slot][i][q] & 0xffff0000u);
                }
        }
        if (bias_x) {
#pragma unroll
            for (int i = 0; i < NA; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    bs[2 * q] += __uint_as_float(xr[slot][i][q] << 16);
                    bs[2 * q + 1] += __uint_as_float(xr[slot][i][q] & 0xffff0000u);
                }
        }
    };
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int wk = wave / CW, wc = wave % CW;
    // quadrants of this wave's 64 x 64 that lie inside the slab (uniform per wave)
    const bool k0_live = kbase + 64 * wk < a.Kpad, k1_live = kbase + 64 * wk + 32 < a.Kpad;
    const bool c0_live = cdbase + 64 * wc < a.CdPad, c1_live = cdbase + 64 * wc + 32 < a.CdPad;
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    int cur = 0;
#pragma unroll
    for (int u = 0; u < RING; ++u) load(u, mbeg + u * MS);
    stage(0, 0);
    __syncthreads();
    for (int mb = mbeg; mb < mend; mb += RING * MS) {
#pragma unroll
        for (int u = 0; u < RING; ++u) {
            const int m0 = mb + u * MS;
            if (m0 >= mend) break;                        // uniform over the workgroup
            load(u, m0 + RING * MS);                      // slot u went to LDS a stage ago: refill it
            const unsigned char* at = As + cur * MS * RSA;
            const unsigned char* dt = Ds + cur * MS * RSD;
            if (k0_live && c0_live) {
#pragma unroll
                for (int s = 0; s < MS / 16; ++s) {
                    const bf16x8 a0 = tr_frag_rs(at, RSA, 16 * s, 64 * wk, lane);
                    const bf16x8 b0 = tr_frag_rs(dt, RSD, 16 * s, 64 * wc, lane);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0], 0, 0, 0);
                    if (c1_live) {
                        const bf16x8 b1 = tr_frag_rs(dt, RSD, 16 * s, 64 * wc + 32, lane);
                        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[1], 0, 0, 0);
                        if (k1_live) {
                            const bf16x8 a1 = tr_frag_rs(at, RSA, 16 * s, 64 * wk + 32, lane);
                            acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[2], 0, 0, 0);
                            acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[3], 0, 0, 0);
                        }
                    } else if (k1_live) {
                        const bf16x8 a1 = tr_frag_rs(at, RSA, 16 * s, 64 * wk + 32, lane);
                        acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[2], 0, 0, 0);
                    }
                }
            }
            if (m0 + MS < mend) stage((u + 1) % RING, cur ^ 1);   // that buffer was last read before the previous barrier
            __syncthreads();
            cur ^= 1;
        }
    }
    // ---- the tile -> this chunk's slab ----
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int krow0 = kbase + 64 * wk + 32 * (j >> 1), cd0 = cdbase + 64 * wc + 32 * (j & 1);
        if (krow0 >= a.Kpad || cd0 >= a.CdPad) continue;  // uniform per wave
        float* ws = a.ws + ((long)blockIdx.x * a.Kpad + krow0) * a.CdPad + cd0 + r;
#pragma unroll
        for (int e = 0; e < 16; ++e) ws[(long)((e & 3) + 8 * (e >> 2) + 4 * h) * a.CdPad] = acc[j][e];
    }
    // ---- column sums of the staged rows (bias gradient), reduced over the row lanes in a fixed order ----
    float* red = reinterpret_cast<float*>(wb_smem);       // every wave is past the loop's last barrier
    if (bias_d) {
        if (d_thread) {
#pragma unroll
            for (int i = 0; i < 8; ++i) red[rld * TC + kvd * 8 + i] = bs[i];
        }
        __syncthreads();
        if (tid < TC && cdbase + tid < a.CdPad) {
            float t = 0.f;
            for (int q = 0; q < RLDE; ++q) t += red[q * TC + tid];
            a.bpart[(long)blockIdx.x * a.CdPad + cdbase + tid] = t;
        }
    }
    if (bias_x) {
#pragma unroll
        for (int i = 0; i < 8; ++i) red[rl * TK + kv * 8 + i] = bs[i];
        __syncthreads();
        if (tid < TK && kbase + tid < a.Kpad) {
            float t = 0.f;
            for (int q = 0; q < RLA; ++q) t += red[q * TK + tid];
            a.bpart[(long)blockIdx.x * a.Kpad + kbase + tid] = t;
        }
    }
}

static void launch_wgrad_big(const WgradArgs& a, int S, hipStream_t st) {
    const WgradBigTile t = wgrad_big_tile(a.Kpad, a.CdPad);
    const dim3 grid(S, (a.Kpad + t.tk - 1) / t.tk, (a.CdPad + t.tc - 1) / t.tc), block(512);
#define WGRAD_BIG_CASE(KW, CW, MS)                                                                              \
    if (t.tk == 64 * KW && t.tc == 64 * CW) {                                                                   \
        constexpr int lds = wgrad_big_lds<KW, CW, MS>();                                                        \
        static_assert(lds <= 160 * 1024, "LDS budget");                                                         \
        static LdsOptIn lds_opt;                                                                                \
        lds_opt.ensure(reinterpret_cast<const void*>(&wgrad_big_bf16_kernel<KW, CW, MS>), lds);                 \
        hipLaunchKernelGGL((wgrad_big_bf16_kernel<KW, CW, MS>), grid, block, lds, st, a);                       \
        return;                                                                                                 \
    }
    WGRAD_BIG_CASE(4, 2, 64)
#undef WGRAD_BIG_CASE
}

}  // namespace dvae
