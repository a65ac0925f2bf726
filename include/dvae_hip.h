/*
 * dvae_hip.h -- C ABI of libdvae_hip.so, the MI355X (gfx950) kernels of the D-VAE hot path.
 *
 * The reference (yuyi-sd/D-VAE) is pure PyTorch and has no plugin ABI of its own: the seam this
 * library replaces is the set of ATen ops that ResNetDVAE.forward, VAE_Trainer.train_batch and the
 * epoch-0 branch of Trainer.train dispatch to.  Each entry point below names the reference lines
 * whose arithmetic it performs.
 *
 * Conventions
 *  - Every pointer is a device pointer owned by the caller (torch tensors' data_ptr()).  The library
 *    never allocates, frees or synchronises; every launch is asynchronous on `stream`
 *    (a hipStream_t passed as void*; NULL = the default stream).
 *  - Activations ("act" tensors) are NHWC with the channel count padded to dvae_act_channels(C)
 *    (a power of two >= 4); pad channels hold zeros.  Element type is selected by `dtype`
 *    (DVAE_F32 or DVAE_BF16); reductions, statistics, losses, gradients of parameters and the
 *    optimiser state are always fp32.
 *  - Parameters and their gradients keep the reference's layouts (Conv2d [Cout,Cin,kh,kw],
 *    ConvTranspose2d [Cin,Cout,kh,kw], fp32); dvae_conv_pack produces the GEMM operand images.
 *  - Return value: 0 on success, a negative DVAE_E* code otherwise; dvae_last_error() gives the
 *    message of the calling thread's last failure.
 */
#ifndef DVAE_HIP_H
#define DVAE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DVAE_F32 0
#define DVAE_BF16 1

#define DVAE_OK 0
#define DVAE_EINVAL (-1)   /* bad descriptor / unsupported shape */
#define DVAE_EDTYPE (-2)   /* unsupported dtype */
#define DVAE_ELAUNCH (-3)  /* hipGetLastError() != hipSuccess after a launch */
#define DVAE_EWORKSPACE (-4) /* workspace too small */

#define DVAE_OP_FWD 0
#define DVAE_OP_DGRAD 1

/* One nn.Conv2d / nn.ConvTranspose2d as the reference constructs it
 * (configs/models/resnet.py:12-24, :1964, :2008-2050).  H, W are the spatial dims of the module's
 * input; square kernels, symmetric padding, dilation 1, groups 1. */
typedef struct dvae_conv_desc {
    int32_t B, H, W;
    int32_t Cin, Cout;        /* module in_channels / out_channels (logical, unpadded) */
    int32_t kh, kw, stride, pad;
    int32_t transposed;       /* 0 = Conv2d, 1 = ConvTranspose2d */
} dvae_conv_desc;

/* Fused epilogue of a forward conv: y = act( acc * scale[c] + shift[c] + residual ).
 * scale == NULL means 1, shift == NULL means 0 (a conv bias is passed as shift). */
typedef struct dvae_epilogue {
    const float* scale;       /* [Cout] or NULL */
    const float* shift;       /* [Cout] or NULL */
    const void* residual;     /* act tensor shaped like the output, or NULL */
    int32_t relu;
    int32_t out_f32;          /* write fp32 (channel padding of the fp32 rule) even when dtype is bf16: mu|log_var, rec_pre, p */
    /* Optional fused prologue (inference): the input is normalised per image and channel while it is staged,
     * x <- relu?(x * scale[b][c] + shift[b][c]), with in_scale_shift = [B][2][Cin] fp32 as dvae_norm_fwd_stats
     * produces it (InstanceNorm2d + ReLU folded into the consuming conv).  Only where
     * dvae_conv_supports_input_norm() says so; NULL = off. */
    const float* in_scale_shift;
    int32_t in_relu;
    /* Optional fused statistics of the stored output (the InstanceNorm2d that follows the conv, resnet.py:2030-2046):
     * per image, per chunk of H*W/8 pixels and per channel the sum and the sum of squares, fp32
     * [B][8][2][act channels]; dvae_norm_stats_from_chunks (chunk_rows = H*W/8) turns them into mean / invstd /
     * scale-shift.  An image's chunks do not depend on the batch it is in.
     * dvae_conv_out_stats_floats() gives the size, 0 where the conv cannot produce them; NULL = off. */
    float* out_stats;
    /* Optional second output (inference): emb_out = y + emb_table[emb_labels[b]], the input of decoder_p
     * (emb_p(y).reshape(stem_out.shape) + z, resnet.py:2212, :2219), with emb_table the table in activation layout as
     * dvae_emb_pack leaves it (spatial: [classes][H*W][act channels]; plain, see emb_plain: [classes][act channels]) and
     * emb_out shaped like y.  Same values as
     * dvae_emb_add_fwd on the stored y.  Only where dvae_conv_supports_emb_out() says so; emb_out == NULL = off. */
    const void* emb_table;
    const int64_t* emb_labels;
    void* emb_out;
    int32_t emb_plain;        /* 1: emb_table is the plain class embedding [classes][act channels] (spatial_emb off,
                               * resnet.py:2214-2216), one row for every pixel of an image; 0: the spatial table */
    int32_t reverse_walk;     /* a scheduling hint, never a change of results: kernels that sweep the batch first image to
                               * last take it last image to first.  A launch that starts where the launch before it has
                               * just finished writing finds that data in the memory-side cache (tensors larger than the
                               * cache: 2048-image batches); ignored by kernels that do not sweep.  No counterpart in the
                               * reference. */
} dvae_epilogue;

const char* dvae_version(void);
const char* dvae_last_error(void);
/* Diagnostic switch, process-wide: "conv_wreg" = 0 routes the 3x3 unit-stride bf16 convolutions to the kernel that keeps
 * the weights in LDS instead of the one that keeps them in registers (same results to rounding; for A/B timing inside
 * one process).  No counterpart in the reference. */
int dvae_set_option(const char* name, int value);
int dvae_act_channels(int C);

/* ---- layout conversion at the module boundary ------------------------------------------------
 * ToTensor (uint8 HWC -> float CHW / 255, main_vae_disentangle.py:409) and NCHW <-> NHWC. */
int dvae_u8_to_act(const uint8_t* u8_nhwc, const int32_t* index /* [B] or NULL */, int64_t first,
                   void* act, int B, int HW, int dtype, void* stream);
int dvae_nchw_to_act(const float* nchw, void* act, int B, int C, int H, int W, int dtype, void* stream);
int dvae_act_to_nchw(const void* act, float* nchw, int B, int C, int H, int W, int act_channels,
                     int channel_offset, int act_is_f32, int dtype, void* stream);

/* ---- convolutions (aten::conv2d / aten::conv_transpose2d and their autograd formulas) --------
 * resnet.py:27-29, :2140, :2245-2246, :2199, :2012-2023, :2034-2045. */
size_t dvae_conv_packed_elems(const dvae_conv_desc* d, int op);
int dvae_conv_pack(const float* w_ref, void* w_packed, const dvae_conv_desc* d, int op, int dtype, void* stream);
/* Batched form: dvae_conv_pack_job fills one job record (dvae_conv_pack_job_bytes() bytes of host memory)
 * per (weight, direction); the records, copied to device memory back to back, are all executed by one
 * dvae_conv_pack_batch launch (re-packing every conv of the model after an optimiser step). */
size_t dvae_conv_pack_job_bytes(void);
int dvae_conv_pack_job(const float* w_ref, void* w_packed, const dvae_conv_desc* d, int op, int dtype, void* job_host);
int dvae_conv_pack_batch(const void* jobs_dev, int njobs, int dtype, void* stream);
int dvae_conv_fwd(const void* x, const void* w_packed_fwd, const dvae_epilogue* ep, void* y,
                  const dvae_conv_desc* d, int dtype, void* stream);
int dvae_conv_supports_input_norm(const dvae_conv_desc* d, int dtype);
int dvae_conv_supports_emb_out(const dvae_conv_desc* d, int dtype);
int dvae_conv_dgrad(const void* dy, const void* w_packed_dgrad, void* dx, int accumulate,
                    const dvae_conv_desc* d, int dtype, void* stream);
size_t dvae_conv_wgrad_workspace_bytes(const dvae_conv_desc* d);   /* the workspace must be 16-byte aligned */
int dvae_conv_wgrad(const void* x, const void* dy, float* dw_ref, float* db_ref /* or NULL */,
                    void* workspace, size_t workspace_bytes, const dvae_conv_desc* d, int dtype, void* stream);
/* 1 when this (desc, dtype) is served by the 256 x 128-tile weight-gradient kernel (bf16, K >= 512, >= 128 dense
 * channels, >= 100 000 pixel rows), 0 otherwise: lets a test assert which kernel a launch exercised. */
int dvae_conv_wgrad_uses_big_tiles(const dvae_conv_desc* d, int dtype);
/* The same gradient in two steps, so that a backward pass reduces the slabs of ALL its convs in one launch (autograd
 * of the convs in resnet.py:2139-2150, :2008-2050; main_vae_disentangle.py:209).  dvae_conv_wgrad_partial leaves the
 * per-chunk slabs (and, with want_bias, the bias partials) in a workspace that belongs to this conv until the batch
 * reduction has run; dvae_wgrad_reduce_job fills one job record (dvae_wgrad_reduce_job_bytes() bytes of host memory)
 * for that workspace and reports its block count; dvae_wgrad_reduce_batch runs njobs records (copied to device
 * memory by the caller) with max_blocks = the largest block count among them.  Shapes whose bias gradient cannot ride
 * on the slab launches return DVAE_EINVAL from the partial / job calls: use dvae_conv_wgrad for those. */
int dvae_conv_wgrad_partial(const void* x, const void* dy, int want_bias, void* workspace, size_t workspace_bytes,
                            const dvae_conv_desc* d, int dtype, void* stream);
size_t dvae_wgrad_reduce_job_bytes(void);
int dvae_wgrad_reduce_job(const void* workspace, float* dw_ref, float* db_ref /* or NULL */, const dvae_conv_desc* d,
                          int dtype, void* job_host, int* nblocks /* out, or NULL */);
int dvae_wgrad_reduce_batch(const void* jobs_dev, int njobs, int max_blocks, void* stream);

/* ---- BatchNorm2d (+ residual add) + ReLU  (resnet.py:27-30) and InstanceNorm2d + ReLU
 * (resnet.py:2014-2015, :2019-2020): one family of kernels over `groups` row groups of
 * `rows_per_group` NHWC rows each (BatchNorm: 1 group of B*H*W rows; InstanceNorm: B groups of H*W).
 * save_mean / save_invstd / scale / shift are [groups][C] fp32. */
size_t dvae_norm_workspace_bytes(int groups, int rows_per_group, int C);
int dvae_norm_fwd_train(const void* x, const void* residual, void* y,
                        const float* gamma, const float* beta,          /* NULL for InstanceNorm */
                        float* running_mean, float* running_var,        /* NULL for InstanceNorm */
                        float* save_mean, float* save_invstd,
                        void* workspace, size_t workspace_bytes,
                        int groups, int rows_per_group, int C, int relu,
                        float momentum, float eps, int dtype, void* stream);
/* dvae_norm_fwd_train without a workspace argument, for callers that cannot allocate device memory: code under HIP
 * graph capture, tight serving loops, harnesses that forbid hipMalloc.  It has no counterpart in the reference.  Same
 * kernels and bit-identical results; the partial statistics and scale / shift pairs live in a scratch array that is part
 * of the library's device image (DVAE_STATIC_SCRATCH_FLOATS floats = 1 MiB on every GPU the library is used on, held
 * until the process ends).  The call allocates and frees nothing.  Shapes that need more scratch than that
 * (dvae_norm_workspace_bytes(groups, rows_per_group, C) > 1 MiB: BatchNorm with C > 128 at large row counts,
 * InstanceNorm of more than ~240 images of 32x32x64) return DVAE_EWORKSPACE and run nothing.  Every call on a GPU uses
 * that GPU's one array: calls that overlap on one GPU (different streams or threads) race; nothing in it needs to start
 * at zero (each call overwrites every entry it later reads). */
#define DVAE_STATIC_SCRATCH_FLOATS 262144
int dvae_norm_fwd_train_static(const void* x, const void* residual, void* y,
                               const float* gamma, const float* beta,          /* NULL for InstanceNorm */
                               float* running_mean, float* running_var,        /* NULL for InstanceNorm */
                               float* save_mean, float* save_invstd,
                               int groups, int rows_per_group, int C, int relu,
                               float momentum, float eps, int dtype, void* stream);
/* statistics only: save_mean / save_invstd and scale_shift [groups][2][C] (scale = gamma * invstd, shift = beta -
 * mean * scale), for a consumer that applies the normalisation itself (dvae_epilogue.in_scale_shift) */
int dvae_norm_fwd_stats(const void* x, const float* gamma, const float* beta, float* save_mean, float* save_invstd,
                        float* scale_shift, void* workspace, size_t workspace_bytes, int groups, int rows_per_group,
                        int C, float eps, int dtype, void* stream);
int dvae_bn_fwd_eval(const void* x, const void* residual, void* y,
                     const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                     float* scale_shift /* [2][C] scratch */, int rows, int C, int relu, float eps,
                     int dtype, void* stream);
/* eval-mode BatchNorm folded to a per-channel scale/shift for dvae_conv_fwd's epilogue */
/* Statistics of a normalisation group from the (sum, sum of squares) chunks a conv epilogue left in `chunks`
 * ([groups][rows_per_group / chunk_rows][2][C], see dvae_epilogue.out_stats): same outputs as dvae_norm_fwd_stats. */
size_t dvae_conv_out_stats_floats(const dvae_conv_desc* d, int dtype);
int dvae_norm_stats_from_chunks(const float* chunks, int chunk_rows, const float* gamma, const float* beta,
                                float* save_mean, float* save_invstd, float* scale_shift, int groups,
                                int rows_per_group, int C, float eps, int dtype, void* stream);
int dvae_bn_fold_eval(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                      float* scale, float* shift, int C, float eps, void* stream);
int dvae_norm_bwd(const void* dy, const void* y, const void* x,
                  void* dx, void* dresidual /* or NULL: receives the ReLU-masked dy */,
                  const float* gamma /* NULL for InstanceNorm */,
                  const float* save_mean, const float* save_invstd,
                  float* dgamma, float* dbeta /* NULL for InstanceNorm */,
                  void* workspace, size_t workspace_bytes,
                  int groups, int rows_per_group, int C, int relu, int dtype, void* stream);

/* ---- class embedding of decoder_p's input (resnet.py:2211-2221) ------------------------------ */
/* emb [num_classes][C*HW] (reference layout, NCHW per class) -> table [class][px][c] in the act dtype */
int dvae_emb_pack(const float* emb, void* table, int num_classes, int HW, int C, int spatial, int dtype, void* stream);
int dvae_emb_add_fwd(const void* table, const int64_t* y, const void* add /* or NULL */, void* out,
                     int B, int HW, int C, int spatial, int dtype, void* stream);
int dvae_emb_bwd(const void* dout, const int64_t* y, float* demb, int B, int HW, int C,
                 int num_classes, int spatial, int dtype, void* stream);

/* ---- reparameterise + per-latent KL + rate constraint ---------------------------------------
 * resnet.py:2257-2259 and main_vae_disentangle.py:204-205.  mulv is fp32 [rows][2L]: mu in
 * columns [0,L), log_var in [L,2L).  eps is fp32 [rows][L]. */
int dvae_philox_normal(float* out, const int32_t* image_index /* [B] or NULL */, int64_t first_image,
                       int B, int per_image, uint64_t seed, uint32_t stream_id, void* stream);
int dvae_reparam_kl_fwd(const float* mulv, const float* eps, void* z1, float* kl_partial /* [DVAE_NPARTIAL] */,
                        int rows, int L, int dtype, void* stream);
/* same, with eps drawn in-kernel from the Philox stream of dvae_philox_normal for an [image][rows][L]
 * tensor; eps_out (or NULL) receives it for the backward pass. */
int dvae_reparam_kl_fwd_philox(const void* mulv, int mulv_is_act /* mulv stored in the act dtype (inference) */,
                               float* eps_out, void* z1, float* kl_partial, int B, int rows_per_image, int L,
                               const int32_t* image_index /* [B] or NULL */, int64_t first_image, uint64_t seed,
                               uint32_t stream_id, int dtype, void* stream);
int dvae_reparam_kl_bwd(const void* dz1, const float* mulv, const float* eps, const float* scal,
                        void* dmulv, int rows, int L, int dtype, void* stream);
/* Inference encoder head in one launch: the 1x1 conv fc_mu|fc_var (resnet.py:2245-2246; packed forward weights of
 * a conv with 2L outputs, mu rows first, bias fp32 [2L]) and the reparameterisation (resnet.py:2257-2259) with eps
 * drawn in-kernel from the Philox stream of dvae_philox_normal.  Writes only z [B*H*W][L] in the act dtype; mu and
 * log_var stay in registers (fp32).  No KL sums: the purification pass (main_vae_disentangle.py:45-95) takes no
 * loss.  dvae_fc_reparam_supported tells whether (desc, L, dtype) can run here (bf16, 1x1 unit stride, L % 128 == 0). */
int dvae_fc_reparam_supported(const dvae_conv_desc* d, int L, int dtype);
int dvae_fc_reparam_fwd_philox(const void* x, const void* wp, const float* bias, void* z, const dvae_conv_desc* d,
                               int L, const int32_t* image_index /* [B] or NULL */, int64_t first_image,
                               uint64_t seed, uint32_t stream_id, int dtype, void* stream);

/* The whole inference latent path in one launch: fc_mu|fc_var, the reparameterisation (as dvae_fc_reparam_fwd_philox) and
 * post_conv (resnet.py:2197-2199: packed forward weights of the 1x1 conv latent -> 128, bias fp32 [128] or NULL), with
 * post_conv's optional second output emb_out = out + emb_table[emb_labels[image]][pixel] (resnet.py:2212, :2219; emb_plain:
 * one table row per class) as in dvae_epilogue.  z itself is never written.  Same bits as the two launches it replaces.
 * dvae_fc_reparam_post_supported: bf16, L = 256, 128 stem channels, H * W a multiple of 64. */
int dvae_fc_reparam_post_supported(const dvae_conv_desc* d_fc, const dvae_conv_desc* d_post, int L, int dtype);
int dvae_fc_reparam_post_fwd_philox(const void* x, const void* wp_fc, const float* bias_fc, const void* wp_post,
                                    const float* bias_post, void* out, const void* emb_table, const int64_t* emb_labels,
                                    void* emb_out, int emb_plain, const dvae_conv_desc* d_fc, const dvae_conv_desc* d_post,
                                    int L, const int32_t* image_index /* [B] or NULL */, int64_t first_image, uint64_t seed,
                                    uint32_t stream_id, int dtype, void* stream);

/* ---- reconstruction losses (main_vae_disentangle.py:190-192, :205-207) -----------------------
 * rec_pre and p (the decoders' 3-channel outputs, written by dvae_conv_fwd with out_f32) are fp32
 * [npix][4]; x is an act tensor; rec = sigmoid(rec_pre) is written fp32 [npix][4]; d_rec_pre / d_p are
 * act tensors (operands of the transposed convs' gradients). */
#define DVAE_NPARTIAL 1024
#define DVAE_SCAL_LOSS 0
#define DVAE_SCAL_LOSS_REC 1
#define DVAE_SCAL_LOSS_REC_P 2
#define DVAE_SCAL_LOSS_P 3
#define DVAE_SCAL_KLD 4
#define DVAE_SCAL_GATE_KL 5
#define DVAE_SCAL_GATE_P 6
#define DVAE_SCAL_GRAD_NORM 7
#define DVAE_SCAL_C_REC 8     /* 2 / (B*3*H*W) */
#define DVAE_SCAL_C_P 9       /* gate_p * 2 / (B*3*H*W) */
#define DVAE_SCAL_C_KL 10     /* gate_kl * kld_weight / rows */
#define DVAE_SCAL_COUNT 16
int dvae_recon_loss_fwd(const float* rec_pre, const float* p, const void* x, float* rec,
                        float* partial /* [3][DVAE_NPARTIAL] */, int npix, int dtype, void* stream);
/* partial sums -> sums[4] = { sum (rec-x)^2, sum (clamp(rec+p)-x)^2, sum p^2, sum_rows KL } */
int dvae_loss_reduce(const float* recon_partial, const float* kl_partial, float* sums, void* stream);
/* sums (possibly summed over ranks) -> scal[DVAE_SCAL_COUNT]; n_elem = B*3*H*W and rows = B*h*w over
 * the ranks that contributed to sums; local_* are this rank's counts (gradient scaling). */
int dvae_loss_gate(const float* sums, float* scal, double n_elem, double rows, double local_n_elem,
                   double local_rows, float kld_target, float kld_weight, int epoch_gt_10, void* stream);
int dvae_recon_loss_bwd(const float* rec, const float* p, const void* x, const float* scal,
                        void* d_rec_pre, void* d_p, int npix, int dtype, void* stream);

/* ---- purification epilogue (main_vae_disentangle.py:62, :65, :67) ----------------------------
 * v = clamp(a - b, 0, 1); a is the image (act tensor), or with a_is_f32 == 1 rec (fp32 [npix][4]), or with
 * a_is_f32 == 2 the decoder's pre-activation (fp32 [npix][4]; rec = sigmoid(a) is formed here, resnet.py:2028);
 * b is the perturbation (fp32 [npix][4]) or NULL; u8 = trunc(255 * v) stored HWC at image `first`+i; v itself is
 * optionally stored as the next pass's input (act tensor). */
int dvae_purify_quant(const void* a, const float* b, int a_is_f32, void* v_out /* or NULL */,
                      uint8_t* out_u8_nhwc /* or NULL */, int64_t first, int B, int HW, int dtype, void* stream);

/* ---- unlearnable-example ingress (PoisonCIFAR10, dataset.py:1235, :1287-1289) ------------------
 * dst = uint8(clip(float(src) + clamp(255 * perturbation, -255, 255), 0, 255)) on uint8 NHWC images, every step one
 * rounded fp32 operation as numpy performs it.  perturbation is fp32 NCHW [rows][3][HW]: row = image index
 * (sample-wise, labels == NULL) or row = labels[image] (class-wise).  The reference scales and clamps the tensor in
 * the dtype of the perturbation file (dataset.py:1235) and the result is rounded to float32 when it is placed into the
 * float32 image-sized mask (dataset.py:534, :550): with prescaled == 0 the kernel forms the fp32 product itself
 * (float32 files); for a float64 file the caller forms clamp(255 * p, -255, 255) in float64, rounds it to fp32 once
 * and passes that with prescaled == 1.
 * mask (uint8 [N], or NULL = all) selects the poisoned images (poison_rate < 1); the others are copied.
 * dst may alias src. */
int dvae_poison_u8(const uint8_t* src_u8_nhwc, const float* perturbation_nchw, const int64_t* labels /* or NULL */,
                   const uint8_t* mask /* or NULL */, uint8_t* dst_u8_nhwc, int64_t N, int HW, int prescaled,
                   void* stream);

/* ---- clip_grad_norm_ + Adam (main_vae_disentangle.py:210-211, resnet18DVAE.yaml:21-23) -------
 * hyper (device, fp32[8]): lr, beta1, beta2, eps, max_norm, grad_scale, steps_done, unused;
 * steps_done is incremented on the device after the update (graph-replay friendly).
 * p, g, m, v must be 16-byte aligned (DVAE_EINVAL otherwise). */
int dvae_sqnorm_partial(const float* g, int64_t n, float* partial /* [DVAE_NPARTIAL] */, void* stream);
int dvae_clip_adam(float* p, const float* g, float* m, float* v, int64_t n, float* hyper,
                   const float* partial, float* scal, void* stream);

#ifdef __cplusplus
}
#endif
#endif
