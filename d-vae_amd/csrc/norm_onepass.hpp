// Training BatchNorm2d (+ residual) + ReLU of a tensor small enough to sit in the registers of ONE grid: forward and
// backward as a single launch each (main_vae_disentangle.py:180-211 runs 8 such layers per step at batch 128, resnet.py:27-30).
//
// The chunked kernels of norm.hpp need three launches each way (statistics / finalize / apply; reduce / finalize / apply)
// because the batch statistics are a grid-wide reduction between two passes over the tensor.  At the reference batch the
// tensor is 8-17 MB: at most 256 workgroups of 256 threads hold it in registers (NV 16-byte vectors per thread), so the
// grid can keep it there across a GRID BARRIER: one read of every operand, one write of every result, one launch.
//
//   phase 1   every workgroup: its vectors -> registers, per-channel partial statistics -> `partial[workgroup]`
//   barrier   all workgroups have written their partials (see onepass_grid_barrier)
//   phase 2   every workgroup merges ALL partials itself, in the same fixed order (bitwise the same result everywhere, no
//             second barrier), then normalises its registers and stores
//
// The barrier needs every workgroup resident at once: the host side launches at most 256 workgroups (one per CU at
// worst) of 256 threads with no more LDS than any CU has free, so residency only ever waits for other kernels' workgroups
// to finish, which they do without depending on this one.  The wait is bounded: a workgroup that spins longer than
// ONEPASS_SPIN_LIMIT polls gives up, sets barrier[2] and finishes (results then are wrong, and the flag says so).
#pragma once
#include "norm.hpp"

namespace dvae {

constexpr unsigned ONEPASS_SPIN_LIMIT = 1u << 22;          // polls of >= 0.5 us each: seconds
constexpr int ONEPASS_MAX_BLOCKS = 256;

// barrier[0]: arrivals, barrier[1]: departures, barrier[2]: set if a wait ever timed out.  Zero before the first launch;
// the last workgroup to leave re-arms it, so launches that follow each other on a stream need no reset in between.
// Launches that overlap on one device must not share the words.
__device__ inline void onepass_grid_barrier(unsigned* barrier, unsigned nblocks) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();                                   // this workgroup's partials: visible device-wide before the count
        atomicAdd(&barrier[0], 1u);
        unsigned polls = 0;
        while (__hip_atomic_load(&barrier[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nblocks) {
            __builtin_amdgcn_s_sleep(8);
            if (++polls > ONEPASS_SPIN_LIMIT) { atomicOr(&barrier[2], 1u); break; }
        }
        __threadfence();                                   // the other workgroups' partials: no stale cache line survives
        if (atomicAdd(&barrier[1], 1u) == nblocks - 1) {   // everybody has left the loop above
            __hip_atomic_store(&barrier[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&barrier[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
}

__device__ inline void unpack8(const u32x4& r, float (&v)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[2 * i] = bf_lo(r[i]); v[2 * i + 1] = bf_hi(r[i]); }
}
__device__ inline u32x4 pack8(const float (&v)[8]) {
    u32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = bf_pack(v[2 * i], v[2 * i + 1]);
    return r;
}

// Thread t of workgroup b owns vectors (b * NV + k) * 256 + t, k < NV: always the same 8 channels (256 is a multiple of
// the C / 8 vectors of a row).
template <int NV>
__global__ __launch_bounds__(256) void bn_onepass_fwd_kernel(const bf16* __restrict__ x, const bf16* __restrict__ res,
                                                             bf16* __restrict__ y, const float* gamma, const float* beta,
                                                             float* running_mean, float* running_var, float* save_mean,
                                                             float* save_invstd, float* partial /* [blocks][3][C] */,
                                                             unsigned* barrier, long nvec, int C, int relu, float momentum,
                                                             float eps) {
    __shared__ float sn[256], sm[256 * 8], sq[256 * 8];
    __shared__ float ssc[256], ssh[256];
    const int tid = threadIdx.x, CV = C >> 3, RL = 256 / CV, cvi = tid % CV, rl = tid / CV;
    const long v0 = (long)blockIdx.x * NV * 256 + tid;
    const u32x4* xv = reinterpret_cast<const u32x4*>(x);
    u32x4 xr[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const long i = v0 + k * 256;
        xr[k] = i < nvec ? xv[i] : u32x4{0u, 0u, 0u, 0u};
    }
    // ---- phase 1: (count, mean, M2) of this thread's values per channel, sums shifted by the first value ----
    float s[8], ss[8], k0[8];
    int cnt = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { s[i] = 0.f; ss[i] = 0.f; k0[i] = 0.f; }
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        if (v0 + k * 256 < nvec) {
            float v[8];
            unpack8(xr[k], v);
            if (cnt == 0) {
#pragma unroll
                for (int i = 0; i < 8; ++i) k0[i] = v[i];
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) { const float d = v[i] - k0[i]; s[i] += d; ss[i] += d * d; }
            ++cnt;
        }
    }
    const float n = (float)cnt;
    sn[tid] = n;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float mean = k0[i], m2 = 0.f;
        if (cnt > 0) { const float ms = s[i] / n; mean = k0[i] + ms; m2 = ss[i] - s[i] * ms; }
        sm[tid * 8 + i] = mean;
        sq[tid * 8 + i] = m2 < 0.f ? 0.f : m2;
    }
    __syncthreads();
    for (int step = RL >> 1; step > 0; step >>= 1) {
        if (rl < step) {
            const int o = tid + step * CV;
            const float na = sn[tid], nb = sn[o];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float a_n = na, a_m = sm[tid * 8 + i], a_q = sq[tid * 8 + i];
                chan_merge(a_n, a_m, a_q, nb, sm[o * 8 + i], sq[o * 8 + i]);
                sm[tid * 8 + i] = a_m; sq[tid * 8 + i] = a_q;
            }
            sn[tid] = na + nb;
        }
        __syncthreads();
    }
    if (rl == 0) {
        float* p = partial + (long)blockIdx.x * 3 * C + cvi * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) { p[i] = sn[tid]; p[C + i] = sm[tid * 8 + i]; p[2 * C + i] = sq[tid * 8 + i]; }
    }
    onepass_grid_barrier(barrier, gridDim.x);
    // the residual's loads fly while the partials are merged
    u32x4 rr[NV];
    if (res) {
        const u32x4* rv = reinterpret_cast<const u32x4*>(res);
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const long i = v0 + k * 256;
            rr[k] = i < nvec ? rv[i] : u32x4{0u, 0u, 0u, 0u};
        }
    }
    // ---- phase 2: every workgroup merges all partials, 256 / C lanes per channel over interleaved workgroups ----
    {
        const int lpc = 256 / C, c = tid % C, q = tid / C, nb = gridDim.x;
        float an = 0.f, am = 0.f, aq = 0.f;
        for (int b0 = q; b0 < nb; b0 += 8 * lpc) {
            float pn[8], pm[8], pq[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int b = b0 + u * lpc;
                const float* p = partial + (long)(b < nb ? b : b0) * 3 * C + c;
                pn[u] = p[0]; pm[u] = p[C]; pq[u] = p[2 * C];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (b0 + u * lpc >= nb) break;
                chan_merge(an, am, aq, pn[u], pm[u], pq[u]);
            }
        }
        __syncthreads();
        sn[tid] = an; sm[tid] = am; sq[tid] = aq;
        __syncthreads();
        if (q == 0) {
            for (int o = 1; o < lpc; ++o) chan_merge(an, am, aq, sn[o * C + c], sm[o * C + c], sq[o * C + c]);
            const float var = aq / an;
            const float invstd = 1.0f / sqrtf(var + eps);
            const float sc = gamma[c] * invstd;
            ssc[c] = sc;
            ssh[c] = beta[c] - am * sc;
            if (blockIdx.x == 0) {
                save_mean[c] = am;
                save_invstd[c] = invstd;
                if (running_mean) {
                    const float unb = an > 1.f ? aq / (an - 1.f) : var;
                    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * am;
                    running_var[c] = (1.f - momentum) * running_var[c] + momentum * unb;
                }
            }
        }
        __syncthreads();
    }
    float sc[8], sh[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { sc[i] = ssc[cvi * 8 + i]; sh[i] = ssh[cvi * 8 + i]; }
    u32x4* yv = reinterpret_cast<u32x4*>(y);
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const long i = v0 + k * 256;
        if (i < nvec) {
            float v[8], r[8];
            unpack8(xr[k], v);
            if (res) unpack8(rr[k], r);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float o = v[j] * sc[j] + sh[j];
                if (res) o += r[j];
                if (relu) o = fmaxf(o, 0.f);
                v[j] = o;
            }
            yv[i] = pack8(v);
        }
    }
}

// dx = gamma * invstd * (g - mean(g) - xhat * mean(g * xhat)), g = dy * [y > 0]; dresidual = g; dgamma = sum(g * xhat),
// dbeta = sum(g)
template <int NV>
__global__ __launch_bounds__(256) void bn_onepass_bwd_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ y,
                                                             const bf16* __restrict__ x, bf16* __restrict__ dx,
                                                             bf16* __restrict__ dres, const float* gamma,
                                                             const float* save_mean, const float* save_invstd,
                                                             float* dgamma, float* dbeta, float* partial /* [blocks][2][C] */,
                                                             unsigned* barrier, long nvec, int C, float nrows, int relu) {
    __shared__ float s1s[256 * 8], s2s[256 * 8];
    __shared__ float ca[256], c1[256], c2[256];
    const int tid = threadIdx.x, CV = C >> 3, RL = 256 / CV, cvi = tid % CV, rl = tid / CV;
    const long v0 = (long)blockIdx.x * NV * 256 + tid;
    const u32x4* gv = reinterpret_cast<const u32x4*>(dy);
    const u32x4* xv = reinterpret_cast<const u32x4*>(x);
    const u32x4* yv = reinterpret_cast<const u32x4*>(y);
    u32x4 gr[NV], xr[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const long i = v0 + k * 256;
        const bool in = i < nvec;
        gr[k] = in ? gv[i] : u32x4{0u, 0u, 0u, 0u};
        xr[k] = in ? xv[i] : u32x4{0u, 0u, 0u, 0u};
        if (relu) {                                        // the ReLU mask goes into the kept gradient at once
            const u32x4 yy = in ? yv[i] : u32x4{0u, 0u, 0u, 0u};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                unsigned m = 0u;
                if (bf_lo(yy[j]) > 0.f) m |= 0x0000ffffu;
                if (bf_hi(yy[j]) > 0.f) m |= 0xffff0000u;
                gr[k][j] &= m;
            }
        }
    }
    float mean[8], istd[8], s1[8], s2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        mean[i] = save_mean[cvi * 8 + i]; istd[i] = save_invstd[cvi * 8 + i];
        s1[i] = 0.f; s2[i] = 0.f;
    }
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        float g[8], v[8];
        unpack8(gr[k], g);
        unpack8(xr[k], v);
#pragma unroll
        for (int i = 0; i < 8; ++i) { s1[i] += g[i]; s2[i] += g[i] * (v[i] - mean[i]) * istd[i]; }     // (absent vectors: g = 0)
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) { s1s[tid * 8 + i] = s1[i]; s2s[tid * 8 + i] = s2[i]; }
    __syncthreads();
    for (int step = RL >> 1; step > 0; step >>= 1) {
        if (rl < step) {
            const int o = tid + step * CV;
#pragma unroll
            for (int i = 0; i < 8; ++i) { s1s[tid * 8 + i] += s1s[o * 8 + i]; s2s[tid * 8 + i] += s2s[o * 8 + i]; }
        }
        __syncthreads();
    }
    if (rl == 0) {
        float* p = partial + (long)blockIdx.x * 2 * C + cvi * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) { p[i] = s1s[tid * 8 + i]; p[C + i] = s2s[tid * 8 + i]; }
    }
    onepass_grid_barrier(barrier, gridDim.x);
    {
        const int lpc = 256 / C, c = tid % C, q = tid / C, nb = gridDim.x;
        float a1 = 0.f, a2 = 0.f;
        for (int b0 = q; b0 < nb; b0 += 8 * lpc) {
            float p1[8], p2[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int b = b0 + u * lpc;
                const float* p = partial + (long)(b < nb ? b : b0) * 2 * C + c;
                p1[u] = p[0]; p2[u] = p[C];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (b0 + u * lpc < nb) { a1 += p1[u]; a2 += p2[u]; }
        }
        __syncthreads();
        s1s[tid] = a1; s2s[tid] = a2;
        __syncthreads();
        if (q == 0) {
            for (int o = 1; o < lpc; ++o) { a1 += s1s[o * C + c]; a2 += s2s[o * C + c]; }
            ca[c] = (gamma ? gamma[c] : 1.f) * save_invstd[c];
            c1[c] = a1 / nrows;
            c2[c] = a2 / nrows;
            if (blockIdx.x == 0) {
                if (dgamma) dgamma[c] = a2;
                if (dbeta) dbeta[c] = a1;
            }
        }
        __syncthreads();
    }
    float fa[8], f1[8], f2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { fa[i] = ca[cvi * 8 + i]; f1[i] = c1[cvi * 8 + i]; f2[i] = c2[cvi * 8 + i]; }
    u32x4* ov = reinterpret_cast<u32x4*>(dx);
    u32x4* rv = reinterpret_cast<u32x4*>(dres);
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const long i = v0 + k * 256;
        if (i < nvec) {
            float g[8], v[8], o[8];
            unpack8(gr[k], g);
            unpack8(xr[k], v);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float xhat = (v[j] - mean[j]) * istd[j];
                o[j] = fa[j] * (g[j] - f1[j] - xhat * f2[j]);
            }
            ov[i] = pack8(o);
            if (dres) rv[i] = gr[k];
        }
    }
}

// vectors per thread (0: the tensor does not fit one grid) and workgroups for a [rows][C] bf16 tensor
struct OnepassPlan { int nv, blocks; };
static inline OnepassPlan onepass_plan(long rows, int C) {
    if (C < 8 || C > 256 || (C & (C - 1)) || rows <= 0) return {0, 0};
    const long nvec = rows * C / 8;
    for (int nv = 4; nv <= 16; nv *= 2) {
        const long blocks = (nvec + 256L * nv - 1) / (256L * nv);
        if (blocks <= ONEPASS_MAX_BLOCKS) return {nv, (int)blocks};
    }
    return {0, 0};
}

}  // namespace dvae
