// HBM-bound kernels of the D-VAE path: layout conversion at the module boundary, class embedding,
// reparameterise + per-latent KL, reconstruction losses + their gradients, purification quantiser,
// Philox noise, global-norm clip + Adam.  All reductions are fixed-order (per-block partials summed
// by one block) so results are bitwise reproducible.
#pragma once
#include "common.hpp"
#include "philox.hpp"

namespace dvae {

// ------------------------------------------------------------------------------------------------
// layout conversion
// ------------------------------------------------------------------------------------------------
// uint8 HWC image rows -> act [B][HW][CS] with x/255 (ToTensor); pad channels = 0.
template <typename T>
__global__ void u8_to_act_kernel(const uint8_t* u8, const int32_t* index, long first, T* act, int B, int HW, int CS) {
    const long total = (long)B * HW;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int b = (int)(i / HW), px = (int)(i % HW);
        const long img = index ? (long)index[b] : first + b;
        const uint8_t* s = u8 + (img * HW + px) * 3;
        T* d = act + i * CS;
        // the reference divides the uint8 value by 255 in fp32 (torchvision ToTensor)
        d[0] = (T)((float)s[0] / 255.f);
        d[1] = (T)((float)s[1] / 255.f);
        d[2] = (T)((float)s[2] / 255.f);
        for (int c = 3; c < CS; ++c) d[c] = (T)0.f;
    }
}

template <typename T>
__global__ void nchw_to_act_kernel(const float* nchw, T* act, int B, int C, int HW, int CS) {
    const long total = (long)B * HW * CS;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % CS);
        const long bp = i / CS;
        const int px = (int)(bp % HW), b = (int)(bp / HW);
        act[i] = (T)(c < C ? nchw[((long)b * C + c) * HW + px] : 0.f);
    }
}

template <typename T>
__global__ void act_to_nchw_kernel(const T* act, float* nchw, int B, int C, int HW, int CS, int coff) {
    const long total = (long)B * C * HW;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int px = (int)(i % HW);
        const long bc = i / HW;
        const int c = (int)(bc % C), b = (int)(bc / C);
        nchw[i] = (float)act[((long)b * HW + px) * CS + coff + c];
    }
}

// ------------------------------------------------------------------------------------------------
// class embedding.  The reference reshapes emb[y] (width C*HW) to NCHW; the packed table holds it NHWC in
// the activation dtype ([class][px][c], or [class][c] when not spatial) so that the per-step gather-add
//     out[b][px][c] = table[y[b]][px][c] + add[b][px][c]
// is a pure 16-byte-vector stream.  The table is re-packed whenever the parameters change.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void emb_pack_kernel(const float* emb, T* table, int num_classes, int HW, int C, int spatial) {
    const long total = (long)num_classes * (spatial ? (long)HW * C : C);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        if (spatial) {
            const int c = (int)(i % C);
            const long kp = i / C;
            const int px = (int)(kp % HW), k = (int)(kp / HW);
            table[i] = (T)emb[((long)k * C + c) * HW + px];
        } else {
            table[i] = (T)emb[i];
        }
    }
}

template <typename T>
__global__ void emb_add_kernel(const T* table, const int64_t* y, const T* add, T* out, int B, int HW, int C, int spatial) {
    constexpr int VEC = Act<T>::VEC;
    const int CV = C / VEC;
    const long per = (long)HW * CV;
    const long total = (long)B * per;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int b = (int)(i / per);
        const long r = i % per;
        const T* src = table + (spatial ? ((long)y[b] * per + r) * VEC : ((long)y[b] * CV + r % CV) * VEC);
        float v[VEC], a[VEC];
        Act<T>::load(src, v);
        if (add) {
            Act<T>::load(add + i * VEC, a);
#pragma unroll
            for (int k = 0; k < VEC; ++k) v[k] += a[k];
        }
        Act<T>::store(out + i * VEC, v);
    }
}

// demb[k][j] = sum over samples b with y[b] == k (ascending b) of the matching dout entries
template <typename T>
__global__ void emb_bwd_kernel(const T* dout, const int64_t* y, float* demb, int B, int HW, int C, int num_classes, int spatial) {
    const long width = spatial ? (long)C * HW : C;
    const long total = (long)num_classes * width;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int k = (int)(i / width);
        const long j = i % width;
        float s = 0.f;
        for (int b = 0; b < B; ++b) {
            if ((int)y[b] != k) continue;
            if (spatial) {
                const int c = (int)(j / HW), px = (int)(j % HW);
                s += (float)dout[((long)b * HW + px) * C + c];
            } else {
                for (int px = 0; px < HW; ++px) s += (float)dout[((long)b * HW + px) * C + j];
            }
        }
        demb[i] = s;
    }
}

// Spatial embedding (one [C][HW] table row per class): one workgroup per (class, block of pixels).  Wave 0 compacts the
// samples of the class into an LDS list (ballot + prefix count, 1024 samples per round); every thread then sums its
// 16-byte channel vector over the list in ascending sample order, four loads in flight (a scan of all B labels with a
// dependent load per match is a chain of memory round trips).  Reads are whole channel runs of a pixel row; the result
// goes to the reference's [class][channel][pixel] layout.
template <typename T>
__global__ __launch_bounds__(256) void emb_bwd_spatial_kernel(const T* dout, const int64_t* y, float* demb, int B, int HW, int C) {
    constexpr int VEC = Act<T>::VEC;
    __shared__ int list[1024];
    __shared__ int count;
    const int CV = C / VEC, PXW = 256 / CV;               // channel vectors per pixel; pixels per workgroup
    const int blocks_per_class = HW / PXW;
    const int k = blockIdx.x / blocks_per_class, px = (blockIdx.x % blocks_per_class) * PXW + threadIdx.x / CV;
    const int cv = threadIdx.x % CV;
    float s[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) s[e] = 0.f;
    for (int b0 = 0; b0 < B; b0 += 1024) {
        if (threadIdx.x < 64) {
            int n = 0;
            for (int c0 = b0; c0 < min(B, b0 + 1024); c0 += 64) {
                const int b = c0 + (int)threadIdx.x;
                const bool hit = b < B && (int)y[b] == k;
                const unsigned long long m = __ballot(hit);
                if (hit) list[n + __popcll(m & ((1ull << threadIdx.x) - 1ull))] = b;
                n += __popcll(m);
            }
            if (threadIdx.x == 0) count = n;
        }
        __syncthreads();
        const int n = count;
        const T* base = dout + (long)px * C + cv * VEC;
        for (int i = 0; i < n; i += 4) {
            float v[4][VEC];
#pragma unroll
            for (int u = 0; u < 4; ++u) Act<T>::load(base + (long)list[min(i + u, n - 1)] * HW * C, v[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (i + u >= n) break;
#pragma unroll
                for (int e = 0; e < VEC; ++e) s[e] += v[u][e];
            }
        }
        __syncthreads();
    }
    float* o = demb + (long)k * C * HW + (long)(cv * VEC) * HW + px;
#pragma unroll
    for (int e = 0; e < VEC; ++e) o[(long)e * HW] = s[e];
}

__global__ void philox_normal_kernel(float* out, int B, int per_image, const PhiloxKey key) {
    const int quads = per_image / 4;
    const long total = (long)B * quads;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int b = (int)(i / quads), q = (int)(i % quads);
        const uint64_t img = key.index ? (uint64_t)key.index[b] : (uint64_t)(key.first + b);
        *reinterpret_cast<f32x4*>(out + ((long)b * per_image + (long)q * 4)) = philox_normal4(key, img, (uint32_t)q);
    }
}

// ------------------------------------------------------------------------------------------------
// reparameterise + KL.  mulv fp32 [rows][2L]; one wave walks rows, 4 channels per lane per step.
// z1 = eps * exp(0.5*lv) + mu ;  KL row term = -0.5 * sum_c (1 + lv - mu^2 - exp(lv))
// ------------------------------------------------------------------------------------------------
// eps == NULL: the noise is drawn in-kernel (same Philox stream as philox_normal_kernel for an
// [image][row][channel] tensor) and, when eps_out != NULL, stored for the backward pass.
// TMV: element type of mulv (fp32, or the act dtype on the inference path where no loss is taken).
template <typename T, typename TMV>
__global__ __launch_bounds__(256) void reparam_kl_fwd_kernel(const TMV* mulv, const float* eps, float* eps_out, T* z1,
                                                             float* kl_partial, int rows, int L, int rows_per_image,
                                                             const PhiloxKey key) {
    __shared__ float red[4];
    const int LV = L / 4;
    const long total = (long)rows * LV;
    float acc = 0.f;
    // two independent 4-channel groups per iteration keep more bytes in flight per lane
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x; i0 < total; i0 += 2 * stride) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const long i = i0 + u * stride;
            if (i >= total) break;
            const long row = i / LV;
            const int c = (int)(i % LV) * 4;
            float mu[4], lv[4];
            Act<TMV>::load4(mulv + row * 2 * L + c, mu);
            Act<TMV>::load4(mulv + row * 2 * L + L + c, lv);
            f32x4 e;
            if (eps) {
                e = *reinterpret_cast<const f32x4*>(eps + row * L + c);
            } else {
                const int b = (int)(row / rows_per_image);
                const long in_img = (row % rows_per_image) * L + c;
                const uint64_t img = key.index ? (uint64_t)key.index[b] : (uint64_t)(key.first + b);
                e = philox_normal4(key, img, (uint32_t)(in_img >> 2));
                if (eps_out) *reinterpret_cast<f32x4*>(eps_out + row * L + c) = e;
            }
            float z[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                z[k] = e[k] * expf(0.5f * lv[k]) + mu[k];
                acc += 1.f + lv[k] - mu[k] * mu[k] - expf(lv[k]);
            }
            Act<T>::store4(z1 + row * L + c, z);
        }
    }
    const float s = block_sum_256(acc, red);
    if (threadIdx.x == 0) kl_partial[blockIdx.x] = -0.5f * s;
}

// dmu = dz1 + c_kl * mu ;  dlv = dz1 * eps * 0.5 * exp(0.5 lv) + c_kl * 0.5 * (exp(lv) - 1)
template <typename T>
__global__ __launch_bounds__(256) void reparam_kl_bwd_kernel(const T* dz1, const float* mulv, const float* eps,
                                                             const float* scal, T* dmulv, int rows, int L) {
    const int LV = L / 4;
    const long total = (long)rows * LV;
    const float ckl = scal[DVAE_SCAL_C_KL];
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long row = i / LV;
        const int c = (int)(i % LV) * 4;
        const f32x4 mu = *reinterpret_cast<const f32x4*>(mulv + row * 2 * L + c);
        const f32x4 lv = *reinterpret_cast<const f32x4*>(mulv + row * 2 * L + L + c);
        const f32x4 e = *reinterpret_cast<const f32x4*>(eps + row * L + c);
        float dz[4], dm[4], dl[4];
        Act<T>::load4(dz1 + row * L + c, dz);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            dm[k] = dz[k] + ckl * mu[k];
            dl[k] = dz[k] * e[k] * 0.5f * expf(0.5f * lv[k]) + ckl * 0.5f * (expf(lv[k]) - 1.f);
        }
        Act<T>::store4(dmulv + row * 2 * L + c, dm);
        Act<T>::store4(dmulv + row * 2 * L + L + c, dl);
    }
}

// ------------------------------------------------------------------------------------------------
// reconstruction losses.  One pixel (4 padded channels, 3 live) per thread per step.
// ------------------------------------------------------------------------------------------------
// rec_pre and p (the 3-channel decoder outputs) are always fp32 [npix][4]; x is an act tensor.
template <typename T>
__global__ __launch_bounds__(256) void recon_loss_fwd_kernel(const float* rec_pre, const float* p, const T* x, float* rec,
                                                             float* partial, int npix, int CS) {
    __shared__ float red[4];
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (long)gridDim.x * blockDim.x) {
        float rp[4], pv[4], xv[4], r[4];
        Act<float>::load4(rec_pre + i * 4, rp);
        Act<float>::load4(p + i * 4, pv);
        Act<T>::load4(x + i * CS, xv);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            r[c] = 1.f / (1.f + expf(-rp[c]));
            const float d0 = r[c] - xv[c];
            const float d1 = fminf(fmaxf(r[c] + pv[c], 0.f), 1.f) - xv[c];
            a0 += d0 * d0; a1 += d1 * d1; a2 += pv[c] * pv[c];
        }
        r[3] = 0.f;
        Act<float>::store4(rec + i * 4, r);
    }
    const float s0 = block_sum_256(a0, red);
    const float s1 = block_sum_256(a1, red);
    const float s2 = block_sum_256(a2, red);
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = s0;
        partial[DVAE_NPARTIAL + blockIdx.x] = s1;
        partial[2 * DVAE_NPARTIAL + blockIdx.x] = s2;
    }
}

__global__ __launch_bounds__(256) void loss_reduce_kernel(const float* recon_partial, const float* kl_partial, float* sums) {
    __shared__ float red[4];
    for (int k = 0; k < 4; ++k) {
        const float* src = k < 3 ? recon_partial + k * DVAE_NPARTIAL : kl_partial;
        float a = 0.f;
        if (src) for (int i = threadIdx.x; i < DVAE_NPARTIAL; i += 256) a += src[i];
        const float s = block_sum_256(a, red);
        if (threadIdx.x == 0) sums[k] = s;
    }
}

// main_vae_disentangle.py:205-207: builtin max(tensor, float) keeps the tensor unless float > tensor.
__global__ void loss_gate_kernel(const float* sums, float* scal, double n_elem, double rows, double local_n_elem,
                                 double local_rows, float kld_target, float kld_weight, int epoch_gt_10) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const float loss_rec = (float)(sums[0] / n_elem);
    const float loss_rec_p = (float)(sums[1] / n_elem);
    const float loss_p = (float)(sums[2] / n_elem);
    const float kld = (float)(sums[3] / rows);
    const bool gate_kl = !(kld_target > kld);
    float loss = loss_rec + loss_rec_p + kld_weight * (gate_kl ? kld : kld_target);
    bool gate_p = false;
    if (epoch_gt_10) {
        gate_p = !(0.002f > loss_p);
        loss += gate_p ? loss_p : 0.002f;
    }
    scal[DVAE_SCAL_LOSS] = loss;
    scal[DVAE_SCAL_LOSS_REC] = loss_rec;
    scal[DVAE_SCAL_LOSS_REC_P] = loss_rec_p;
    scal[DVAE_SCAL_LOSS_P] = loss_p;
    scal[DVAE_SCAL_KLD] = kld;
    scal[DVAE_SCAL_GATE_KL] = gate_kl ? 1.f : 0.f;
    scal[DVAE_SCAL_GATE_P] = gate_p ? 1.f : 0.f;
    scal[DVAE_SCAL_C_REC] = (float)(2.0 / local_n_elem);
    scal[DVAE_SCAL_C_P] = gate_p ? (float)(2.0 / local_n_elem) : 0.f;
    scal[DVAE_SCAL_C_KL] = gate_kl ? (float)((double)kld_weight / local_rows) : 0.f;
}

// d rec = c (rec - x) + c (clamp(rec+p) - x) [0 <= rec+p <= 1] ; d p = c (clamp(rec+p) - x)[..] + c_p p
// d rec_pre = d rec * rec (1 - rec)
template <typename T>
__global__ __launch_bounds__(256) void recon_loss_bwd_kernel(const float* rec, const float* p, const T* x, const float* scal,
                                                             T* d_rec_pre, T* d_p, int npix, int CS) {
    const float c = scal[DVAE_SCAL_C_REC], cp = scal[DVAE_SCAL_C_P];
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (long)gridDim.x * blockDim.x) {
        float r[4], pv[4], xv[4], dr[4], dp[4];
        Act<float>::load4(rec + i * 4, r);
        Act<float>::load4(p + i * 4, pv);
        Act<T>::load4(x + i * CS, xv);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float s = r[k] + pv[k];
            const bool in = (s >= 0.f) && (s <= 1.f);
            const float g1 = in ? c * (s - xv[k]) : 0.f;
            const float drec = c * (r[k] - xv[k]) + g1;
            dr[k] = drec * r[k] * (1.f - r[k]);
            dp[k] = g1 + cp * pv[k];
        }
        dr[3] = 0.f; dp[3] = 0.f;
        Act<T>::store4(d_rec_pre + i * CS, dr);
        Act<T>::store4(d_p + i * CS, dp);
        for (int k = 4; k < CS; k += 4) {   // bf16 acts carry 8 channels: keep the pad at zero
            const float z[4] = {0.f, 0.f, 0.f, 0.f};
            Act<T>::store4(d_rec_pre + i * CS + k, z);
            Act<T>::store4(d_p + i * CS + k, z);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// purification epilogue: v = clamp(a - b, 0, 1); u8 = trunc(255 * v) (fp32 product, as numpy does)
// ------------------------------------------------------------------------------------------------
// a: the image (act dtype), rec (fp32 [.,4]) or, with `sigmoid`, the decoder's pre-activation (fp32 [.,4]);
// b: the perturbation, fp32 [.,4], or NULL.
template <typename TA, typename T>
__global__ void purify_quant_kernel(const TA* a, int CSA, const float* b, int CS, T* vout, uint8_t* u8, long first, int B, int HW,
                                    int sigmoid) {
    const long total = (long)B * HW;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        float av[4], bv[4] = {0.f, 0.f, 0.f, 0.f}, v[4];
        Act<TA>::load4(a + i * CSA, av);
        if (b) Act<float>::load4(b + i * 4, bv);
        if (sigmoid) {                                   // same expression as recon_loss_fwd_kernel: identical bits
#pragma unroll
            for (int c = 0; c < 3; ++c) av[c] = 1.f / (1.f + expf(-av[c]));
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = fminf(fmaxf(av[c] - bv[c], 0.f), 1.f);
        v[3] = 0.f;
        if (vout) {
            Act<T>::store4(vout + i * CS, v);
            for (int k = 4; k < CS; k += 4) { const float z[4] = {0.f, 0.f, 0.f, 0.f}; Act<T>::store4(vout + i * CS + k, z); }
        }
        if (u8) {
            uint8_t* d = u8 + (first * HW + i) * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) d[c] = (uint8_t)(int)fminf(fmaxf(255.f * v[c], 0.f), 255.f);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// unlearnable-example ingress: data = uint8(clip(float(data) + clamp(255 * perturbation, -255, 255), 0, 255))
// (PoisonCIFAR10, dataset.py:1235, :1287-1289).  perturbation is fp32 NCHW, one image per sample
// (sample-wise) or per class (class-wise: row = label).  Every step is a separately rounded fp32
// operation, as numpy / torch perform them: no fused multiply-add.  prescaled: pert already holds
// clamp(255 * perturbation, -255, 255) rounded to fp32 (the product of a float64 perturbation file is formed in
// float64 and rounded once, dataset.py:1235 then :534/:550).
// ------------------------------------------------------------------------------------------------
__global__ void poison_u8_kernel(const uint8_t* src, const float* pert, const int64_t* labels, const uint8_t* mask,
                                 uint8_t* dst, long N, int HW, int prescaled) {
    const long total = N * HW;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long n = i / HW;
        const int px = (int)(i % HW);
        const uint8_t* s = src + i * 3;
        uint8_t* d = dst + i * 3;
        if (mask && !mask[n]) { d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; continue; }
        const long row = labels ? (long)labels[n] : n;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float pv = pert[(row * 3 + c) * HW + px];
            const float noise = prescaled ? pv : fminf(fmaxf(__fmul_rn(pv, 255.f), -255.f), 255.f);
            const float v = fminf(fmaxf(__fadd_rn((float)s[c], noise), 0.f), 255.f);
            d[c] = (uint8_t)(int)v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// clip_grad_norm_ + Adam over one flat fp32 bucket
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sqnorm_partial_kernel(const float* g, long n, float* partial) {
    __shared__ float red[4];
    float a = 0.f;
    const long n4 = n / 4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(g + i * 4);
        a += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    if (blockIdx.x == 0 && threadIdx.x < (int)(n - n4 * 4)) { const float v = g[n4 * 4 + threadIdx.x]; a += v * v; }
    const float s = block_sum_256(a, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// hyper: lr, beta1, beta2, eps, max_norm, grad_scale, steps_done
__global__ __launch_bounds__(256) void clip_adam_kernel(float* p, const float* g, float* m, float* v, long n,
                                                        const float* hyper, const float* partial, float* scal) {
    __shared__ float red[4];
    float a = 0.f;
    for (int i = threadIdx.x; i < DVAE_NPARTIAL; i += 256) a += partial[i];
    const float sq = block_sum_256(a, red);
    const float gs = hyper[5];
    const float total_norm = sqrtf(sq) * gs;
    const float coef = fminf(hyper[4] / (total_norm + 1e-6f), 1.0f) * gs;
    if (blockIdx.x == 0 && threadIdx.x == 0 && scal) scal[DVAE_SCAL_GRAD_NORM] = total_norm;
    // bias corrections in double (as the reference optimiser computes them), once per workgroup
    __shared__ float corr[2];
    if (threadIdx.x == 0) {
        const double b1 = hyper[1], b2 = hyper[2];
        const double t = (double)hyper[6] + 1.0;
        corr[0] = (float)sqrt(1.0 - pow(b2, t));
        corr[1] = (float)((double)hyper[0] / (1.0 - pow(b1, t)));
    }
    __syncthreads();
    const float bc2s = corr[0], step_size = corr[1];
    const float fb1 = hyper[1], fb2 = hyper[2], eps = hyper[3];
    const long n4 = n / 4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const f32x4 g4 = *reinterpret_cast<const f32x4*>(g + i * 4);
        f32x4 m4 = *reinterpret_cast<const f32x4*>(m + i * 4), v4 = *reinterpret_cast<const f32x4*>(v + i * 4);
        f32x4 p4 = *reinterpret_cast<const f32x4*>(p + i * 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float gi = g4[k] * coef;
            m4[k] = m4[k] * fb1 + (1.f - fb1) * gi;
            v4[k] = v4[k] * fb2 + (1.f - fb2) * gi * gi;
            p4[k] = p4[k] - step_size * m4[k] / (sqrtf(v4[k]) / bc2s + eps);
        }
        *reinterpret_cast<f32x4*>(m + i * 4) = m4;
        *reinterpret_cast<f32x4*>(v + i * 4) = v4;
        *reinterpret_cast<f32x4*>(p + i * 4) = p4;
    }
    for (long i = n4 * 4 + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float gi = g[i] * coef;
        const float mi = m[i] * fb1 + (1.f - fb1) * gi;
        const float vi = v[i] * fb2 + (1.f - fb2) * gi * gi;
        m[i] = mi; v[i] = vi;
        p[i] = p[i] - step_size * mi / (sqrtf(vi) / bc2s + eps);
    }
}

__global__ void hyper_tick_kernel(float* hyper) {
    if (threadIdx.x == 0 && blockIdx.x == 0) hyper[6] += 1.f;
}

}  // namespace dvae
