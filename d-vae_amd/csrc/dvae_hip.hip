// libdvae_hip.so: extern "C" entry points (declared in include/dvae_hip.h) over the gfx950 kernels.
// Host code here only validates shapes, derives launch geometry and launches on the caller's stream.
#include "common.hpp"
#include "elementwise.hpp"
#include "igemm.hpp"
#include "conv_halo.hpp"
#include "conv_wreg.hpp"
#include "conv_image.hpp"
#include "wgrad_halo.hpp"
#include "wgrad_big.hpp"
#include "conv_stem.hpp"
#include "conv_point.hpp"
#include "conv_rgb.hpp"
#include "conv_down.hpp"
#include "conv_up.hpp"
#include "conv_mid.hpp"
#include "head_post.hpp"
#include "norm.hpp"

namespace dvae {
thread_local char g_err[512] = "";
}
using namespace dvae;

// Library-owned scratch of dvae_norm_fwd_train_static (see there).  `scratch` names the array in host code only as a
// symbol: its device address is looked up per device, once, and kept in g_scratch_ptr.
static __device__ __align__(256) float scratch[DVAE_STATIC_SCRATCH_FLOATS];
static float* g_scratch_ptr[64];
static std::atomic<unsigned long long> g_scratch_known{0ull};

static float* static_scratch_for_current_device() {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess || dev < 0 || dev >= 64) {
        snprintf(g_err, sizeof(g_err), "static scratch: hipGetDevice: %s (device %d)", hipGetErrorString(e), dev);
        return nullptr;
    }
    const unsigned long long bit = 1ull << dev;
    if (!(g_scratch_known.load(std::memory_order_acquire) & bit)) {
        float* p = nullptr;
        e = hipGetSymbolAddress((void**)&p, HIP_SYMBOL(scratch));      // address of this device's copy of the array
        if (e != hipSuccess || !p) {
            snprintf(g_err, sizeof(g_err), "static scratch: hipGetSymbolAddress on device %d: %s", dev, hipGetErrorString(e));
            return nullptr;
        }
        g_scratch_ptr[dev] = p;                   // two threads racing here store the same address
        g_scratch_known.fetch_or(bit, std::memory_order_release);
    }
    return g_scratch_ptr[dev];
}

// diagnostic switches (dvae_set_option): which of two equivalent kernels serves a launch
static std::atomic<int> g_opt_wreg{1};
static std::atomic<int> g_opt_rgb{1};
static std::atomic<int> g_opt_down{1};
static std::atomic<int> g_opt_up{1};
static std::atomic<int> g_opt_mid{1};

static inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline int grid_for(long n, int block = 256, int cap = 2048) {
    long g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}
static inline int act_c(int C, int dtype) {
    int c = act_channels(C);
    if (dtype == DVAE_BF16 && c < 8) c = 8;
    return c;
}

#define DISPATCH_T(dtype, ...)                                   \
    if ((dtype) == DVAE_F32) { using T = float; __VA_ARGS__ }    \
    else if ((dtype) == DVAE_BF16) { using T = bf16; __VA_ARGS__ } \
    else DVAE_FAIL(DVAE_EDTYPE, "unknown dtype %d", (dtype));

extern "C" {

const char* dvae_version(void) { return "dvae_hip 0.2 (gfx950)"; }

int dvae_set_option(const char* name, int value) {
    if (!name) DVAE_FAIL(DVAE_EINVAL, "set_option: null name");
    if (!strcmp(name, "conv_wreg")) { g_opt_wreg.store(value != 0); return DVAE_OK; }
    if (!strcmp(name, "conv_rgb")) { g_opt_rgb.store(value != 0); return DVAE_OK; }
    if (!strcmp(name, "conv_down")) { g_opt_down.store(value != 0); return DVAE_OK; }
    if (!strcmp(name, "conv_up")) { g_opt_up.store(value != 0); return DVAE_OK; }
    if (!strcmp(name, "conv_mid")) { g_opt_mid.store(value != 0); return DVAE_OK; }
    DVAE_FAIL(DVAE_EINVAL, "set_option: unknown option '%s'", name);
}
const char* dvae_last_error(void) { return g_err; }
int dvae_act_channels(int C) { return act_channels(C); }

// ---------------------------------------------------------------------------------------------
int dvae_u8_to_act(const uint8_t* u8, const int32_t* index, int64_t first, void* act, int B, int HW, int dtype, void* st) {
    if (B <= 0 || HW <= 0) DVAE_FAIL(DVAE_EINVAL, "u8_to_act: empty");
    DISPATCH_T(dtype, hipLaunchKernelGGL((u8_to_act_kernel<T>), dim3(grid_for((long)B * HW)), dim3(256), 0, S(st), u8,
                                         index, (long)first, (T*)act, B, HW, act_c(3, dtype));)
    DVAE_CHECK_LAUNCH("u8_to_act");
    return DVAE_OK;
}

int dvae_nchw_to_act(const float* nchw, void* act, int B, int C, int H, int W, int dtype, void* st) {
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0) DVAE_FAIL(DVAE_EINVAL, "nchw_to_act: empty");
    const int CS = act_c(C, dtype);
    DISPATCH_T(dtype, hipLaunchKernelGGL((nchw_to_act_kernel<T>), dim3(grid_for((long)B * H * W * CS)), dim3(256), 0,
                                         S(st), nchw, (T*)act, B, C, H * W, CS);)
    DVAE_CHECK_LAUNCH("nchw_to_act");
    return DVAE_OK;
}

int dvae_act_to_nchw(const void* act, float* nchw, int B, int C, int H, int W, int act_channels_, int coff,
                     int act_is_f32, int dtype, void* st) {
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || coff < 0 || coff + C > act_channels_)
        DVAE_FAIL(DVAE_EINVAL, "act_to_nchw: bad shape");
    const int dt = act_is_f32 ? DVAE_F32 : dtype;
    DISPATCH_T(dt, hipLaunchKernelGGL((act_to_nchw_kernel<T>), dim3(grid_for((long)B * C * H * W)), dim3(256), 0, S(st),
                                      (const T*)act, nchw, B, C, H * W, act_channels_, coff);)
    DVAE_CHECK_LAUNCH("act_to_nchw");
    return DVAE_OK;
}

// ---------------------------------------------------------------------------------------------
size_t dvae_conv_packed_elems(const dvae_conv_desc* d, int op) {
    GemmPlan P;
    // element count is the same for both dtypes only if BK matches; report the larger (bf16, BK = 64)
    if (!d || build_plan(*d, op, DVAE_BF16, P)) return 0;
    return (size_t)P.Npad * P.Ktot;
}

int dvae_conv_pack(const float* w, void* out, const dvae_conv_desc* d, int op, int dtype, void* st) {
    GemmPlan P;
    if (!d) DVAE_FAIL(DVAE_EINVAL, "null desc");
    int rc = build_plan(*d, op, dtype, P);
    if (rc) return rc;
    PackArgs a;
    a.w = w; a.out = out; a.N = P.N; a.Npad = P.Npad; a.Cg = P.Cg; a.cg_log2 = P.cg_log2; a.Ktot = P.Ktot;
    a.sN = P.sN; a.sC = P.sC; a.nphase = P.nphase;
    for (int i = 0; i < 4; ++i) a.pk[i] = P.pk[i];
    DISPATCH_T(dtype, hipLaunchKernelGGL((pack_weight_kernel<T>), dim3(grid_for((long)P.Npad * P.Ktot)), dim3(256), 0,
                                         S(st), a);)
    DVAE_CHECK_LAUNCH("conv_pack");
    return DVAE_OK;
}

size_t dvae_conv_pack_job_bytes(void) { return sizeof(PackArgs); }

int dvae_conv_pack_job(const float* w, void* out, const dvae_conv_desc* d, int op, int dtype, void* job_host) {
    GemmPlan P;
    if (!d || !job_host) DVAE_FAIL(DVAE_EINVAL, "null desc / job");
    int rc = build_plan(*d, op, dtype, P);
    if (rc) return rc;
    PackArgs a;
    memset(&a, 0, sizeof(a));
    a.w = w; a.out = out; a.N = P.N; a.Npad = P.Npad; a.Cg = P.Cg; a.cg_log2 = P.cg_log2; a.Ktot = P.Ktot;
    a.sN = P.sN; a.sC = P.sC; a.nphase = P.nphase;
    for (int i = 0; i < 4; ++i) a.pk[i] = P.pk[i];
    memcpy(job_host, &a, sizeof(a));
    return DVAE_OK;
}

int dvae_conv_pack_batch(const void* jobs_dev, int njobs, int dtype, void* st) {
    if (!jobs_dev || njobs <= 0) DVAE_FAIL(DVAE_EINVAL, "pack_batch: no jobs");
    DISPATCH_T(dtype, hipLaunchKernelGGL((pack_weight_batch_kernel<T>), dim3(48, njobs), dim3(256), 0, S(st),
                                         (const PackArgs*)jobs_dev);)
    DVAE_CHECK_LAUNCH("conv_pack_batch");
    return DVAE_OK;
}

int dvae_conv_fwd(const void* x, const void* wp, const dvae_epilogue* ep, void* y, const dvae_conv_desc* d, int dtype,
                  void* st) {
    GemmPlan P;
    if (!d) DVAE_FAIL(DVAE_EINVAL, "null desc");
    int rc = build_plan(*d, DVAE_OP_FWD, dtype, P);
    if (rc) return rc;
    dvae_epilogue e = {nullptr, nullptr, nullptr, 0, 0, nullptr, 0, nullptr, nullptr, nullptr, nullptr, 0, 0};
    if (ep) e = *ep;
    if (e.out_stats && (dtype != DVAE_BF16 || e.out_f32 || !halo_features_applicable(P)))
        DVAE_FAIL(DVAE_EINVAL, "conv_fwd: fused output statistics are not available for this shape/dtype "
                               "(see dvae_conv_out_stats_floats)");
    if (e.in_scale_shift && (dtype != DVAE_BF16 || !halo_features_applicable(P)))
        DVAE_FAIL(DVAE_EINVAL, "conv_fwd: fused input normalisation is not available for this shape/dtype "
                               "(see dvae_conv_supports_input_norm)");
    if (dtype == DVAE_F32)
        return launch_igemm<float>(P, x, wp, y, e.scale, e.shift, e.residual, e.relu, 0, 1, S(st));
    if (g_opt_rgb.load() &&
        try_launch_rgb_bf16(P, x, wp, y, e.scale, e.shift, e.residual, e.relu, 0, e.out_f32, S(st), e.in_scale_shift, e.in_relu,
                            e.out_stats)) {
        DVAE_CHECK_LAUNCH("conv_rgb");
        return DVAE_OK;
    }
    if (g_opt_wreg.load() &&
        try_launch_wreg_bf16(P, x, wp, y, e.scale, e.shift, e.residual, e.relu, 0, e.out_f32, S(st), nullptr, e.in_scale_shift,
                             e.in_relu, e.out_stats, e.reverse_walk)) {
        DVAE_CHECK_LAUNCH("conv_wreg");
        return DVAE_OK;
    }
    if (try_launch_halo_bf16(P, x, wp, y, e.scale, e.shift, e.residual, e.relu, 0, e.out_f32, S(st), e.in_scale_shift,
                             e.in_relu, nullptr, e.out_stats)) {
        DVAE_CHECK_LAUNCH("conv_halo");
        return DVAE_OK;
    }
    if (e.in_scale_shift || e.out_stats)                  // the gather GEMM has neither: never drop them silently
        DVAE_FAIL(DVAE_EINVAL, "conv_fwd: fused input normalisation / output statistics unavailable for this launch");
    if (g_opt_down.load() && try_launch_down_bf16(P, x, wp, y, e.scale, e.shift, e.residual, e.relu, 0, e.out_f32, S(st))) {
        DVAE_CHECK_LAUNCH("conv_down");
        return DVAE_OK;
    }
    if (g_opt_up.load() && try_launch_up_bf16(P, x, wp, y, e.scale, e.shift, e.residual, e.relu, 0, e.out_f32, S(st))) {
        DVAE_CHECK_LAUNCH("conv_up");
        return DVAE_OK;
    }
    if (g_opt_mid.load() && try_launch_mid_bf16(P, x, wp, y, e.scale, e.shift, e.residual, e.relu, 0, e.out_f32, S(st))) {
        DVAE_CHECK_LAUNCH("conv_mid");
        return DVAE_OK;
    }
    if (try_launch_image_bf16(P, x, wp, y, e.scale, e.shift, e.residual, e.relu, 0, e.out_f32, S(st))) {
        DVAE_CHECK_LAUNCH("conv_image");
        return DVAE_OK;
    }
    if (try_launch_stem_bf16(P, x, wp, y, e.scale, e.shift, e.residual, e.relu, 0, e.out_f32, S(st))) {
        DVAE_CHECK_LAUNCH("conv_stem");
        return DVAE_OK;
    }
    if (try_launch_point_bf16(P, x, wp, y, e.scale, e.shift, e.residual, e.relu, 0, e.out_f32, S(st), e.emb_table, e.emb_labels,
                              e.emb_out, e.emb_plain)) {
        DVAE_CHECK_LAUNCH("conv_point");
        return DVAE_OK;
    }
    if (e.emb_out) DVAE_FAIL(DVAE_EINVAL, "conv_fwd: fused embedding output unavailable for this launch (see dvae_conv_supports_emb_out)");
    return launch_igemm<bf16>(P, x, wp, y, e.scale, e.shift, e.residual, e.relu, 0, e.out_f32, S(st));
}

int dvae_conv_supports_emb_out(const dvae_conv_desc* d, int dtype) {
    GemmPlan P;
    if (!d || dtype != DVAE_BF16 || build_plan(*d, DVAE_OP_FWD, dtype, P)) return 0;
    return (point_cin(P) == 256 && (P.Hout * P.Wout) % 128 == 0) ? 1 : 0;
}

int dvae_conv_supports_input_norm(const dvae_conv_desc* d, int dtype) {
    GemmPlan P;
    if (!d || dtype != DVAE_BF16 || build_plan(*d, DVAE_OP_FWD, dtype, P)) return 0;
    return halo_features_applicable(P) ? 1 : 0;
}

int dvae_conv_dgrad(const void* dy, const void* wp, void* dx, int accumulate, const dvae_conv_desc* d, int dtype,
                    void* st) {
    GemmPlan P;
    if (!d) DVAE_FAIL(DVAE_EINVAL, "null desc");
    int rc = build_plan(*d, DVAE_OP_DGRAD, dtype, P);
    if (rc) return rc;
    if (dtype == DVAE_F32)
        return launch_igemm<float>(P, dy, wp, dx, nullptr, nullptr, nullptr, 0, accumulate, 1, S(st));
    if (g_opt_wreg.load() && try_launch_wreg_bf16(P, dy, wp, dx, nullptr, nullptr, nullptr, 0, accumulate, 0, S(st))) {
        DVAE_CHECK_LAUNCH("conv_wreg");
        return DVAE_OK;
    }
    if (try_launch_halo_bf16(P, dy, wp, dx, nullptr, nullptr, nullptr, 0, accumulate, 0, S(st))) {
        DVAE_CHECK_LAUNCH("conv_halo");
        return DVAE_OK;
    }
    if (g_opt_mid.load() && try_launch_mid_bf16(P, dy, wp, dx, nullptr, nullptr, nullptr, 0, accumulate, 0, S(st))) {
        DVAE_CHECK_LAUNCH("conv_mid");
        return DVAE_OK;
    }
    if (try_launch_image_bf16(P, dy, wp, dx, nullptr, nullptr, nullptr, 0, accumulate, 0, S(st))) {
        DVAE_CHECK_LAUNCH("conv_image");
        return DVAE_OK;
    }
    return launch_igemm<bf16>(P, dy, wp, dx, nullptr, nullptr, nullptr, 0, accumulate, 0, S(st));
}

size_t dvae_conv_wgrad_workspace_bytes(const dvae_conv_desc* d) {
    WgradPlan W;
    if (!d || build_wgrad_plan(*d, DVAE_BF16, W)) return 0;
    size_t a = wgrad_workspace_floats(W);
    if (build_wgrad_plan(*d, DVAE_F32, W)) return 0;
    size_t b = wgrad_workspace_floats(W);
    return (a > b ? a : b) * sizeof(float);
}

int dvae_conv_wgrad_uses_big_tiles(const dvae_conv_desc* d, int dtype) {
    WgradPlan W;
    if (!d || build_wgrad_plan(*d, dtype, W)) return 0;
    return W.big ? 1 : 0;
}

// slab (and bias-partial) launches of one weight gradient into ws; fused: the bias partials ride along
static int wgrad_partials(const WgradPlan& W, const dvae_conv_desc* d, const void* x, const void* dy, void* ws, bool fused,
                          int dtype, void* st) {
    WgradArgs a;
    a.xg = d->transposed ? dy : x;
    a.d = d->transposed ? x : dy;
    a.ws = (float*)ws;
    a.B = W.B; a.Hx = W.Hx; a.Wx = W.Wx; a.cx_log2 = W.cx_log2; a.Hg = W.Hg; a.Wg = W.Wg; a.SY = W.SY; a.SX = W.SX;
    a.CdS = W.CdS; a.ntaps = W.ntaps;
    memcpy(a.dy, W.dy, 16); memcpy(a.dx, W.dx, 16);
    a.M = W.M; a.chunk_rows = W.chunk_rows; a.Kpad = W.Kpad; a.CdPad = W.CdPad;
    a.wg_log2 = a.hg_log2 = -1;
    if (!(W.Wg & (W.Wg - 1)) && !(W.Hg & (W.Hg - 1))) { a.wg_log2 = ilog2_exact(W.Wg); a.hg_log2 = ilog2_exact(W.Hg); }
    a.bias_mode = !fused ? 0 : W.bias_from_gathered ? 2 : 1;
    a.bias_taps = W.bias_taps; a.bpart = (float*)ws + (size_t)W.S * W.Kpad * W.CdPad;
    if (W.halo) {
        WgradHaloArgs h;
        h.xg = (const bf16*)a.xg; h.d = (const bf16*)a.d; h.ws = a.ws;
        h.H = W.Hg; h.W = W.Wg; h.tiles_x = W.Wg / 32; h.tiles_y = W.Hg / 4; h.ntiles = W.halo_tiles; h.per = W.halo_per;
        h.Kpad = W.Kpad; h.CdPad = W.CdPad;
        h.bias_mode = a.bias_mode; h.bpart = a.bpart;
        const int wgs = W.S;
#define WGRAD_HALO_CASE(CX_, CD_)                                                                                  \
        if (W.CxS == CX_ && W.CdS == CD_) {                                                                        \
            static LdsOptIn lds_opt;                                                                               \
            lds_opt.ensure(reinterpret_cast<const void*>(&wgrad_halo_bf16_kernel<CX_, CD_>), WH_LDS);              \
            hipLaunchKernelGGL((wgrad_halo_bf16_kernel<CX_, CD_>), dim3(wgs), dim3(256), WH_LDS, S(st), h);        \
        }
        WGRAD_HALO_CASE(64, 64) WGRAD_HALO_CASE(32, 64) WGRAD_HALO_CASE(8, 64) WGRAD_HALO_CASE(8, 32)
#undef WGRAD_HALO_CASE
    } else if (dtype == DVAE_F32)
        hipLaunchKernelGGL(wgrad_kernel_f32, dim3(W.S, W.Kpad / 64, W.CdPad / 64), dim3(256), 0, S(st), a);
    else if (W.big)
        launch_wgrad_big(a, W.S, S(st));
    else
        hipLaunchKernelGGL(wgrad_kernel_bf16, dim3(W.S, W.Kpad / 64, W.CdPad / 64), dim3(256), 0, S(st), a);
    DVAE_CHECK_LAUNCH("wgrad");
    return DVAE_OK;
}

// the reduction of those slabs into the reference weight (and bias) layout, as a job record
static void wgrad_reduce_args(const WgradPlan& W, const dvae_conv_desc* d, const void* ws, float* dw, float* db, bool fused,
                              WreduceArgs& r) {
    memset(&r, 0, sizeof(r));
    r.ws = (const float*)ws; r.dw = dw; r.S = W.S; r.Kpad = W.Kpad; r.CdPad = W.CdPad; r.Cx = W.Cx;
    r.cx_log2 = W.cx_log2; r.Cd = W.Cd; r.ntaps = W.ntaps; r.sX = W.sX; r.sD = W.sD;
    memcpy(r.woff, W.woff, sizeof(r.woff));
    const long tot = (long)W.Kpad * W.CdPad;
    r.nmain = (int)((tot + 255) / 256);           // 256 slab elements per workgroup (wgrad_reduce_block)
    r.bpart = (const float*)ws + (size_t)W.S * W.Kpad * W.CdPad; r.db = db;
    r.bias_mode = !fused ? 0 : W.bias_from_gathered ? 2 : 1;
    r.bwidth = wgrad_bias_width(W); r.Cb = d->Cout; r.cxs = W.CxS;
    r.nbt = 0;
    for (int t = 0; t < W.ntaps && r.nbt < 4; ++t)
        if (W.bias_taps >> t & 1u) r.btap[r.nbt++] = t;
    r.nblocks = r.nmain + (fused ? (d->Cout + 15) / 16 : 0);
}

int dvae_conv_wgrad(const void* x, const void* dy, float* dw, float* db, void* ws, size_t ws_bytes,
                    const dvae_conv_desc* d, int dtype, void* st) {
    WgradPlan W;
    if (!d) DVAE_FAIL(DVAE_EINVAL, "null desc");
    int rc = build_wgrad_plan(*d, dtype, W);
    if (rc) return rc;
    if (ws_bytes < wgrad_workspace_floats(W) * sizeof(float)) DVAE_FAIL(DVAE_EWORKSPACE, "wgrad workspace too small");
    if ((uintptr_t)ws & 15) DVAE_FAIL(DVAE_EINVAL, "wgrad workspace must be 16-byte aligned");
    const bool fused = db && wgrad_bias_fused(W);
    rc = wgrad_partials(W, d, x, dy, ws, fused, dtype, st);
    if (rc) return rc;
    WreduceArgs r;
    wgrad_reduce_args(W, d, ws, dw, db, fused, r);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)r.nblocks), dim3(256), 0, S(st), r);
    DVAE_CHECK_LAUNCH("wgrad_reduce");
    if (db && !fused) {
        float* part = (float*)ws + (size_t)W.S * W.Kpad * W.CdPad;
        const void* src = dy;
        const int CS = W.bias_from_gathered ? W.CxS : W.CdS;
        const int rows = W.bias_from_gathered ? W.B * W.Hx * W.Wx : W.M;
        if (dtype == DVAE_F32)
            hipLaunchKernelGGL((colsum_partial_kernel<float>), dim3(COLSUM_BLOCKS), dim3(256), 0, S(st), (const float*)src,
                               part, rows, CS);
        else
            hipLaunchKernelGGL((colsum_partial_kernel<bf16>), dim3(COLSUM_BLOCKS), dim3(256), 0, S(st), (const bf16*)src,
                               part, rows, CS);
        DVAE_CHECK_LAUNCH("colsum_partial");
        hipLaunchKernelGGL(colsum_reduce_kernel, dim3((d->Cout + 63) / 64), dim3(256), 0, S(st), part, db, COLSUM_BLOCKS,
                           CS, d->Cout);
        DVAE_CHECK_LAUNCH("colsum_reduce");
    }
    return DVAE_OK;
}

int dvae_conv_wgrad_partial(const void* x, const void* dy, int want_bias, void* ws, size_t ws_bytes, const dvae_conv_desc* d,
                            int dtype, void* st) {
    WgradPlan W;
    if (!d) DVAE_FAIL(DVAE_EINVAL, "null desc");
    int rc = build_wgrad_plan(*d, dtype, W);
    if (rc) return rc;
    if (ws_bytes < wgrad_workspace_floats(W) * sizeof(float)) DVAE_FAIL(DVAE_EWORKSPACE, "wgrad workspace too small");
    if ((uintptr_t)ws & 15) DVAE_FAIL(DVAE_EINVAL, "wgrad workspace must be 16-byte aligned");
    if (want_bias && !wgrad_bias_fused(W)) DVAE_FAIL(DVAE_EINVAL, "wgrad_partial: this shape's bias gradient needs dvae_conv_wgrad");
    return wgrad_partials(W, d, x, dy, ws, want_bias != 0, dtype, st);
}

size_t dvae_wgrad_reduce_job_bytes(void) { return sizeof(WreduceArgs); }

int dvae_wgrad_reduce_job(const void* ws, float* dw, float* db, const dvae_conv_desc* d, int dtype, void* job_host, int* nblocks) {
    WgradPlan W;
    if (!d || !job_host || !ws || !dw) DVAE_FAIL(DVAE_EINVAL, "wgrad_reduce_job: null argument");
    int rc = build_wgrad_plan(*d, dtype, W);
    if (rc) return rc;
    if (db && !wgrad_bias_fused(W)) DVAE_FAIL(DVAE_EINVAL, "wgrad_reduce_job: this shape's bias gradient needs dvae_conv_wgrad");
    if ((uintptr_t)ws & 15) DVAE_FAIL(DVAE_EINVAL, "wgrad_reduce_job: workspace must be 16-byte aligned");
    WreduceArgs r;
    wgrad_reduce_args(W, d, ws, dw, db, db != nullptr, r);
    memcpy(job_host, &r, sizeof(r));
    if (nblocks) *nblocks = r.nblocks;
    return DVAE_OK;
}

int dvae_wgrad_reduce_batch(const void* jobs_dev, int njobs, int max_blocks, void* st) {
    if (!jobs_dev || njobs <= 0 || max_blocks <= 0) DVAE_FAIL(DVAE_EINVAL, "wgrad_reduce_batch: empty");
    hipLaunchKernelGGL(wgrad_reduce_batch_kernel, dim3((unsigned)max_blocks, (unsigned)njobs), dim3(256), 0, S(st),
                       (const WreduceArgs*)jobs_dev);
    DVAE_CHECK_LAUNCH("wgrad_reduce_batch");
    return DVAE_OK;
}

// ---------------------------------------------------------------------------------------------
size_t dvae_norm_workspace_bytes(int groups, int rows_per_group, int C) {
    if (groups <= 0 || rows_per_group <= 0 || C <= 0) return 0;
    return norm_ws_floats(norm_geom(groups, rows_per_group, C)) * sizeof(float);
}

static int norm_check(int groups, int rows_per_group, int C, int dtype) {
    const int vec = dtype == DVAE_BF16 ? 8 : 4;
    if (groups <= 0 || rows_per_group <= 0 || C < vec || (C & (C - 1)) || C / vec > 256)
        DVAE_FAIL(DVAE_EINVAL, "norm: C must be a power of two in [%d, %d]", vec, 256 * vec);
    if ((long)groups * rows_per_group * C > 0x7fffffffL * 4) DVAE_FAIL(DVAE_EINVAL, "norm: tensor too large");
    return DVAE_OK;
}

static int norm_fwd_train_impl(const void* x, const void* res, void* y, const float* gamma, const float* beta,
                               float* running_mean, float* running_var, float* save_mean, float* save_invstd, void* ws,
                               size_t ws_bytes, int groups, int rows_per_group, int C, int relu, float momentum, float eps,
                               int dtype, void* st) {
    int rc = norm_check(groups, rows_per_group, C, dtype);
    if (rc) return rc;
    const NormGeom g = norm_geom(groups, rows_per_group, C);
    if (ws_bytes < norm_ws_floats(g) * sizeof(float)) DVAE_FAIL(DVAE_EWORKSPACE, "norm workspace too small");
    // InstanceNorm of images that fit the registers of one workgroup: one launch, one read, one write
    const bool image_ok = dtype == DVAE_BF16 && groups > 1 && !res && !gamma && !beta && !running_mean;
    const ImageNormPlan ip = image_ok ? in_image_plan(rows_per_group, C) : ImageNormPlan{0, 0};
    if (ip.nv) {
        const dim3 grid(groups, ip.split);
        if (ip.nv == 4)
            hipLaunchKernelGGL((in_image_fwd_kernel<4>), grid, dim3(512), 0, S(st), (const bf16*)x, (bf16*)y, save_mean,
                               save_invstd, rows_per_group, C, eps, relu);
        else
            hipLaunchKernelGGL((in_image_fwd_kernel<8>), grid, dim3(512), 0, S(st), (const bf16*)x, (bf16*)y, save_mean,
                               save_invstd, rows_per_group, C, eps, relu);
        DVAE_CHECK_LAUNCH("in_image_fwd");
        return DVAE_OK;
    }
    float* partial = (float*)ws;
    float* scale_shift = partial + (size_t)groups * g.nchunk * 3 * C + (size_t)groups * 3 * C;
    const long nvec = (long)groups * rows_per_group * C / (dtype == DVAE_BF16 ? 8 : 4);
    DISPATCH_T(dtype,
               hipLaunchKernelGGL((norm_stats_kernel<T>), dim3(g.nchunk, groups), dim3(256), 0, S(st), (const T*)x,
                                  partial, rows_per_group, C, g.rows_per_chunk, g.nchunk);
               DVAE_CHECK_LAUNCH("norm_stats");
               hipLaunchKernelGGL(norm_finalize_kernel, dim3((C + 7) / 8, groups), dim3(256), 0, S(st), partial,
                                  g.nchunk, C, gamma, beta, running_mean, running_var, save_mean, save_invstd,
                                  scale_shift, momentum, eps);
               DVAE_CHECK_LAUNCH("norm_finalize");
               hipLaunchKernelGGL((norm_apply_kernel<T>), dim3(grid_for(nvec, 256, 4096)), dim3(256), 0, S(st),
                                  (const T*)x, (const T*)res, (T*)y, scale_shift, rows_per_group, C, nvec, relu);)
    DVAE_CHECK_LAUNCH("norm_apply");
    return DVAE_OK;
}

int dvae_norm_fwd_train(const void* x, const void* res, void* y, const float* gamma, const float* beta,
                        float* running_mean, float* running_var, float* save_mean, float* save_invstd, void* ws,
                        size_t ws_bytes, int groups, int rows_per_group, int C, int relu, float momentum, float eps,
                        int dtype, void* st) {
    return norm_fwd_train_impl(x, res, y, gamma, beta, running_mean, running_var, save_mean, save_invstd, ws, ws_bytes, groups,
                               rows_per_group, C, relu, momentum, eps, dtype, st);
}

// The same normalisation for callers that cannot allocate: the partial statistics and the scale / shift pairs go to
// `scratch`, a fixed array in this library's device image (every GPU the library runs on holds its own copy until the
// process ends).  Host code must not pass the array's name to a kernel: the name is a host-side stand-in, and the
// array's address on the current device is what hipGetSymbolAddress returns, looked up once per device.
int dvae_norm_fwd_train_static(const void* x, const void* res, void* y, const float* gamma, const float* beta,
                               float* running_mean, float* running_var, float* save_mean, float* save_invstd, int groups,
                               int rows_per_group, int C, int relu, float momentum, float eps, int dtype, void* st) {
    int rc = norm_check(groups, rows_per_group, C, dtype);
    if (rc) return rc;
    const size_t need = norm_ws_floats(norm_geom(groups, rows_per_group, C));
    if (need > DVAE_STATIC_SCRATCH_FLOATS)
        DVAE_FAIL(DVAE_EWORKSPACE, "norm_fwd_train_static: %d group(s) of %d rows x %d channels need %zu scratch floats, "
                  "the library-owned scratch holds %d (use dvae_norm_fwd_train with a workspace of your own)",
                  groups, rows_per_group, C, need, DVAE_STATIC_SCRATCH_FLOATS);
    float* ptr = static_scratch_for_current_device();
    if (!ptr) return DVAE_ELAUNCH;                        // g_err says which HIP call failed
    return norm_fwd_train_impl(x, res, y, gamma, beta, running_mean, running_var, save_mean, save_invstd, ptr,
                               (size_t)DVAE_STATIC_SCRATCH_FLOATS * sizeof(float), groups, rows_per_group, C, relu, momentum,
                               eps, dtype, st);
}

int dvae_norm_fwd_stats(const void* x, const float* gamma, const float* beta, float* save_mean, float* save_invstd,
                        float* scale_shift, void* ws, size_t ws_bytes, int groups, int rows_per_group, int C, float eps,
                        int dtype, void* st) {
    int rc = norm_check(groups, rows_per_group, C, dtype);
    if (rc) return rc;
    const NormGeom g = norm_geom(groups, rows_per_group, C);
    if (ws_bytes < norm_ws_floats(g) * sizeof(float)) DVAE_FAIL(DVAE_EWORKSPACE, "norm workspace too small");
    float* partial = (float*)ws;
    DISPATCH_T(dtype, hipLaunchKernelGGL((norm_stats_kernel<T>), dim3(g.nchunk, groups), dim3(256), 0, S(st), (const T*)x,
                                         partial, rows_per_group, C, g.rows_per_chunk, g.nchunk);)
    DVAE_CHECK_LAUNCH("norm_stats");
    hipLaunchKernelGGL(norm_finalize_kernel, dim3((C + 7) / 8, groups), dim3(256), 0, S(st), partial, g.nchunk, C, gamma,
                       beta, (float*)nullptr, (float*)nullptr, save_mean, save_invstd, scale_shift, 0.f, eps);
    DVAE_CHECK_LAUNCH("norm_finalize");
    return DVAE_OK;
}

size_t dvae_conv_out_stats_floats(const dvae_conv_desc* d, int dtype) {
    GemmPlan P;
    if (!d || dtype != DVAE_BF16 || build_plan(*d, DVAE_OP_FWD, dtype, P) || !halo_features_applicable(P)) return 0;
    return (size_t)P.B * 8 * 2 * P.NS;                    // per image: 8 chunks (one per wave of the tile walk) of H*W/8 pixels
}

int dvae_norm_stats_from_chunks(const float* chunks, int chunk_rows, const float* gamma, const float* beta, float* save_mean,
                                float* save_invstd, float* scale_shift, int groups, int rows_per_group, int C, float eps,
                                int dtype, void* st) {
    int rc = norm_check(groups, rows_per_group, C, dtype);
    if (rc) return rc;
    if (!chunks || chunk_rows <= 0 || rows_per_group % chunk_rows) DVAE_FAIL(DVAE_EINVAL, "norm_stats_from_chunks: chunking");
    const int nchunk = rows_per_group / chunk_rows;
    if (nchunk <= 256) {
        const int CB = C < 256 ? C : 256;                 // C is a power of two (norm_check)
        hipLaunchKernelGGL(norm_finalize_chunks_kernel, dim3(C / CB, groups), dim3(256), 0, S(st), chunks, nchunk, C, CB,
                           chunk_rows, gamma, beta, save_mean, save_invstd, scale_shift, eps);
    } else {
        hipLaunchKernelGGL(norm_finalize_kernel, dim3((C + 7) / 8, groups), dim3(256), 0, S(st), chunks, nchunk, C, gamma, beta,
                           (float*)nullptr, (float*)nullptr, save_mean, save_invstd, scale_shift, 0.f, eps, chunk_rows);
    }
    DVAE_CHECK_LAUNCH("norm_finalize");
    return DVAE_OK;
}

int dvae_bn_fold_eval(const float* gamma, const float* beta, const float* rm, const float* rv, float* scale,
                      float* shift, int C, float eps, void* st) {
    if (C <= 0) DVAE_FAIL(DVAE_EINVAL, "bn_fold: C <= 0");
    hipLaunchKernelGGL(bn_fold_eval_kernel, dim3((C + 63) / 64), dim3(64), 0, S(st), gamma, beta, rm, rv, scale, shift, C, eps);
    DVAE_CHECK_LAUNCH("bn_fold_eval");
    return DVAE_OK;
}

int dvae_bn_fwd_eval(const void* x, const void* res, void* y, const float* gamma, const float* beta, const float* rm,
                     const float* rv, float* scale_shift, int rows, int C, int relu, float eps, int dtype, void* st) {
    int rc = norm_check(1, rows, C, dtype);
    if (rc) return rc;
    rc = dvae_bn_fold_eval(gamma, beta, rm, rv, scale_shift, scale_shift + C, C, eps, st);
    if (rc) return rc;
    const long nvec = (long)rows * C / (dtype == DVAE_BF16 ? 8 : 4);
    DISPATCH_T(dtype, hipLaunchKernelGGL((norm_apply_kernel<T>), dim3(grid_for(nvec, 256, 4096)), dim3(256), 0, S(st),
                                         (const T*)x, (const T*)res, (T*)y, scale_shift, rows, C, nvec, relu);)
    DVAE_CHECK_LAUNCH("bn_apply_eval");
    return DVAE_OK;
}

int dvae_norm_bwd(const void* dy, const void* y, const void* x, void* dx, void* dres, const float* gamma,
                  const float* save_mean, const float* save_invstd, float* dgamma, float* dbeta, void* ws,
                  size_t ws_bytes, int groups, int rows_per_group, int C, int relu, int dtype, void* st) {
    int rc = norm_check(groups, rows_per_group, C, dtype);
    if (rc) return rc;
    if ((dgamma || dbeta) && groups != 1) DVAE_FAIL(DVAE_EINVAL, "affine gradients need a single group");
    const NormGeom g = norm_geom(groups, rows_per_group, C);
    if (ws_bytes < norm_ws_floats(g) * sizeof(float)) DVAE_FAIL(DVAE_EWORKSPACE, "norm workspace too small");
    const ImageNormPlan ip = (dtype == DVAE_BF16 && groups > 1 && !dres && !gamma) ? in_image_plan(rows_per_group, C)
                                                                                     : ImageNormPlan{0, 0};
    if (ip.nv) {
        const dim3 grid(groups, ip.split);
        if (ip.nv == 4)
            hipLaunchKernelGGL((in_image_bwd_kernel<4>), grid, dim3(512), 0, S(st), (const bf16*)dy, (const bf16*)y,
                               (const bf16*)x, (bf16*)dx, save_mean, save_invstd, rows_per_group, C, relu);
        else
            hipLaunchKernelGGL((in_image_bwd_kernel<8>), grid, dim3(512), 0, S(st), (const bf16*)dy, (const bf16*)y,
                               (const bf16*)x, (bf16*)dx, save_mean, save_invstd, rows_per_group, C, relu);
        DVAE_CHECK_LAUNCH("in_image_bwd");
        return DVAE_OK;
    }
    float* partial = (float*)ws;
    float* coef = partial + (size_t)groups * g.nchunk * 3 * C;
    const long nvec = (long)groups * rows_per_group * C / (dtype == DVAE_BF16 ? 8 : 4);
    DISPATCH_T(dtype,
               hipLaunchKernelGGL((norm_bwd_reduce_kernel<T>), dim3(g.nchunk, groups), dim3(256), 0, S(st),
                                  (const T*)dy, (const T*)y, (const T*)x, save_mean, save_invstd, partial,
                                  rows_per_group, C, g.rows_per_chunk, g.nchunk, relu);
               DVAE_CHECK_LAUNCH("norm_bwd_reduce");
               hipLaunchKernelGGL(norm_bwd_finalize_kernel, dim3((C + 7) / 8, groups), dim3(256), 0, S(st), partial,
                                  g.nchunk, C, rows_per_group, gamma, save_invstd, coef, dgamma, dbeta);
               DVAE_CHECK_LAUNCH("norm_bwd_finalize");
               hipLaunchKernelGGL((norm_bwd_apply_kernel<T>), dim3(grid_for(nvec, 256, 4096)), dim3(256), 0, S(st),
                                  (const T*)dy, (const T*)y, (const T*)x, (T*)dx, (T*)dres, save_mean, save_invstd,
                                  coef, rows_per_group, C, nvec, relu);)
    DVAE_CHECK_LAUNCH("norm_bwd_apply");
    return DVAE_OK;
}

// ---------------------------------------------------------------------------------------------
int dvae_emb_pack(const float* emb, void* table, int num_classes, int HW, int C, int spatial, int dtype, void* st) {
    if (num_classes <= 0 || HW <= 0 || C <= 0) DVAE_FAIL(DVAE_EINVAL, "emb_pack: empty");
    const long total = (long)num_classes * (spatial ? (long)HW * C : C);
    DISPATCH_T(dtype, hipLaunchKernelGGL((emb_pack_kernel<T>), dim3(grid_for(total)), dim3(256), 0, S(st), emb, (T*)table,
                                         num_classes, HW, C, spatial);)
    DVAE_CHECK_LAUNCH("emb_pack");
    return DVAE_OK;
}

int dvae_emb_add_fwd(const void* table, const int64_t* y, const void* add, void* out, int B, int HW, int C, int spatial,
                     int dtype, void* st) {
    const int vec = dtype == DVAE_BF16 ? 8 : 4;
    if (B <= 0 || HW <= 0 || C <= 0 || C % vec) DVAE_FAIL(DVAE_EINVAL, "emb_add: C must be a positive multiple of %d", vec);
    DISPATCH_T(dtype, hipLaunchKernelGGL((emb_add_kernel<T>), dim3(grid_for((long)B * HW * C / vec)), dim3(256), 0, S(st),
                                         (const T*)table, y, (const T*)add, (T*)out, B, HW, C, spatial);)
    DVAE_CHECK_LAUNCH("emb_add");
    return DVAE_OK;
}

int dvae_emb_bwd(const void* dout, const int64_t* y, float* demb, int B, int HW, int C, int num_classes, int spatial,
                 int dtype, void* st) {
    if (B <= 0 || HW <= 0 || C <= 0 || num_classes <= 0) DVAE_FAIL(DVAE_EINVAL, "emb_bwd: empty");
    const long total = (long)num_classes * (spatial ? (long)C * HW : C);
    const int vec = dtype == DVAE_BF16 ? 8 : 4;
    if (spatial && C % vec == 0 && 256 % (C / vec) == 0 && HW % (256 / (C / vec)) == 0) {
        const int blocks = num_classes * (HW / (256 / (C / vec)));
        DISPATCH_T(dtype, hipLaunchKernelGGL((emb_bwd_spatial_kernel<T>), dim3(blocks), dim3(256), 0, S(st), (const T*)dout, y,
                                             demb, B, HW, C);)
        DVAE_CHECK_LAUNCH("emb_bwd");
        return DVAE_OK;
    }
    DISPATCH_T(dtype, hipLaunchKernelGGL((emb_bwd_kernel<T>), dim3(grid_for(total, 256, 8192)), dim3(256), 0, S(st),
                                         (const T*)dout, y, demb, B, HW, C, num_classes, spatial);)
    DVAE_CHECK_LAUNCH("emb_bwd");
    return DVAE_OK;
}

// ---------------------------------------------------------------------------------------------
int dvae_philox_normal(float* out, const int32_t* index, int64_t first, int B, int per_image, uint64_t seed,
                       uint32_t stream_id, void* st) {
    if (B <= 0 || per_image <= 0 || per_image % 4) DVAE_FAIL(DVAE_EINVAL, "philox: per_image must be a positive multiple of 4");
    PhiloxKey key = {index, (long)first, (uint32_t)seed, (uint32_t)(seed >> 32), stream_id};
    hipLaunchKernelGGL(philox_normal_kernel, dim3(grid_for((long)B * per_image / 4)), dim3(256), 0, S(st), out, B, per_image, key);
    DVAE_CHECK_LAUNCH("philox_normal");
    return DVAE_OK;
}

int dvae_reparam_kl_fwd(const float* mulv, const float* eps, void* z1, float* kl_partial, int rows, int L, int dtype,
                        void* st) {
    if (rows <= 0 || L <= 0 || L % 4) DVAE_FAIL(DVAE_EINVAL, "reparam: L must be a positive multiple of 4");
    if (!eps) DVAE_FAIL(DVAE_EINVAL, "reparam: eps is NULL (use dvae_reparam_kl_fwd_philox to draw in-kernel)");
    PhiloxKey key = {nullptr, 0, 0u, 0u, 0u};
    DISPATCH_T(dtype, hipLaunchKernelGGL((reparam_kl_fwd_kernel<T, float>), dim3(DVAE_NPARTIAL), dim3(256), 0, S(st), mulv, eps,
                                         (float*)nullptr, (T*)z1, kl_partial, rows, L, rows, key);)
    DVAE_CHECK_LAUNCH("reparam_kl_fwd");
    return DVAE_OK;
}

int dvae_reparam_kl_fwd_philox(const void* mulv, int mulv_is_act, float* eps_out, void* z1, float* kl_partial, int B,
                               int rows_per_image, int L, const int32_t* index, int64_t first, uint64_t seed,
                               uint32_t stream_id, int dtype, void* st) {
    if (B <= 0 || rows_per_image <= 0 || L <= 0 || L % 4) DVAE_FAIL(DVAE_EINVAL, "reparam: L must be a positive multiple of 4");
    PhiloxKey key = {index, (long)first, (uint32_t)seed, (uint32_t)(seed >> 32), stream_id};
    if (mulv_is_act && dtype == DVAE_BF16) {
        hipLaunchKernelGGL((reparam_kl_fwd_kernel<bf16, bf16>), dim3(DVAE_NPARTIAL), dim3(256), 0, S(st), (const bf16*)mulv,
                           (const float*)nullptr, eps_out, (bf16*)z1, kl_partial, B * rows_per_image, L, rows_per_image, key);
    } else {
        DISPATCH_T(dtype, hipLaunchKernelGGL((reparam_kl_fwd_kernel<T, float>), dim3(DVAE_NPARTIAL), dim3(256), 0, S(st),
                                             (const float*)mulv, (const float*)nullptr, eps_out, (T*)z1, kl_partial,
                                             B * rows_per_image, L, rows_per_image, key);)
    }
    DVAE_CHECK_LAUNCH("reparam_kl_fwd_philox");
    return DVAE_OK;
}

int dvae_fc_reparam_supported(const dvae_conv_desc* d, int L, int dtype) {
    GemmPlan P;
    if (!d || dtype != DVAE_BF16 || L <= 0 || build_plan(*d, DVAE_OP_FWD, dtype, P)) return 0;
    return zf_applicable(P, L) ? 1 : 0;
}

int dvae_fc_reparam_fwd_philox(const void* x, const void* wp, const float* bias, void* z, const dvae_conv_desc* d, int L,
                               const int32_t* index, int64_t first, uint64_t seed, uint32_t stream_id, int dtype,
                               void* st) {
    GemmPlan P;
    if (!d || !x || !wp || !z) DVAE_FAIL(DVAE_EINVAL, "fc_reparam: null argument");
    if (dtype != DVAE_BF16) DVAE_FAIL(DVAE_EDTYPE, "fc_reparam: bf16 only (see dvae_fc_reparam_supported)");
    int rc = build_plan(*d, DVAE_OP_FWD, dtype, P);
    if (rc) return rc;
    PhiloxKey key = {index, (long)first, (uint32_t)seed, (uint32_t)(seed >> 32), stream_id};
    if (try_launch_image_zf(P, x, wp, bias, z, L, key, S(st))) {     // 16x16 maps: the image-resident kernel
        DVAE_CHECK_LAUNCH("conv_image_zf");
        return DVAE_OK;
    }
    return launch_igemm_zf(P, x, wp, bias, z, L, key, S(st));
}

int dvae_fc_reparam_post_supported(const dvae_conv_desc* d_fc, const dvae_conv_desc* d_post, int L, int dtype) {
    GemmPlan P1, P2;
    if (!d_fc || !d_post || dtype != DVAE_BF16 || build_plan(*d_fc, DVAE_OP_FWD, dtype, P1) ||
        build_plan(*d_post, DVAE_OP_FWD, dtype, P2))
        return 0;
    return head_post_applicable(P1, P2, L) ? 1 : 0;
}

int dvae_fc_reparam_post_fwd_philox(const void* x, const void* wp_fc, const float* bias_fc, const void* wp_post,
                                    const float* bias_post, void* out, const void* emb_table, const int64_t* emb_labels,
                                    void* emb_out, int emb_plain, const dvae_conv_desc* d_fc, const dvae_conv_desc* d_post,
                                    int L, const int32_t* index, int64_t first, uint64_t seed, uint32_t stream_id, int dtype,
                                    void* st) {
    GemmPlan P1, P2;
    if (!d_fc || !d_post || !x || !wp_fc || !bias_fc || !wp_post || !out) DVAE_FAIL(DVAE_EINVAL, "fc_reparam_post: null argument");
    if (emb_out && (!emb_table || !emb_labels)) DVAE_FAIL(DVAE_EINVAL, "fc_reparam_post: second output without table / labels");
    if (dtype != DVAE_BF16) DVAE_FAIL(DVAE_EDTYPE, "fc_reparam_post: bf16 only (see dvae_fc_reparam_post_supported)");
    int rc = build_plan(*d_fc, DVAE_OP_FWD, dtype, P1);
    if (rc) return rc;
    rc = build_plan(*d_post, DVAE_OP_FWD, dtype, P2);
    if (rc) return rc;
    if (!head_post_applicable(P1, P2, L)) DVAE_FAIL(DVAE_EINVAL, "fc_reparam_post: shape not supported (see dvae_fc_reparam_post_supported)");
    PhiloxKey key = {index, (long)first, (uint32_t)seed, (uint32_t)(seed >> 32), stream_id};
    launch_head_post(P1, P2, x, wp_fc, bias_fc, wp_post, bias_post, out, emb_table, emb_labels, emb_out, emb_plain, key, S(st));
    DVAE_CHECK_LAUNCH("head_post");
    return DVAE_OK;
}

int dvae_reparam_kl_bwd(const void* dz1, const float* mulv, const float* eps, const float* scal, void* dmulv, int rows,
                        int L, int dtype, void* st) {
    if (rows <= 0 || L <= 0 || L % 4) DVAE_FAIL(DVAE_EINVAL, "reparam: L must be a positive multiple of 4");
    DISPATCH_T(dtype, hipLaunchKernelGGL((reparam_kl_bwd_kernel<T>), dim3(grid_for((long)rows * L / 4)), dim3(256), 0,
                                         S(st), (const T*)dz1, mulv, eps, scal, (T*)dmulv, rows, L);)
    DVAE_CHECK_LAUNCH("reparam_kl_bwd");
    return DVAE_OK;
}

// ---------------------------------------------------------------------------------------------
int dvae_recon_loss_fwd(const float* rec_pre, const float* p, const void* x, float* rec, float* partial, int npix,
                        int dtype, void* st) {
    if (npix <= 0) DVAE_FAIL(DVAE_EINVAL, "recon_loss: empty");
    DISPATCH_T(dtype, hipLaunchKernelGGL((recon_loss_fwd_kernel<T>), dim3(DVAE_NPARTIAL), dim3(256), 0, S(st),
                                         (const float*)rec_pre, (const float*)p, (const T*)x, rec, partial, npix, act_c(3, dtype));)
    DVAE_CHECK_LAUNCH("recon_loss_fwd");
    return DVAE_OK;
}

int dvae_loss_reduce(const float* recon_partial, const float* kl_partial, float* sums, void* st) {
    hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(256), 0, S(st), recon_partial, kl_partial, sums);
    DVAE_CHECK_LAUNCH("loss_reduce");
    return DVAE_OK;
}

int dvae_loss_gate(const float* sums, float* scal, double n_elem, double rows, double local_n_elem, double local_rows,
                   float kld_target, float kld_weight, int epoch_gt_10, void* st) {
    if (n_elem <= 0 || rows <= 0 || local_n_elem <= 0 || local_rows <= 0) DVAE_FAIL(DVAE_EINVAL, "loss_gate: counts");
    hipLaunchKernelGGL(loss_gate_kernel, dim3(1), dim3(64), 0, S(st), sums, scal, n_elem, rows, local_n_elem,
                       local_rows, kld_target, kld_weight, epoch_gt_10);
    DVAE_CHECK_LAUNCH("loss_gate");
    return DVAE_OK;
}

int dvae_recon_loss_bwd(const float* rec, const float* p, const void* x, const float* scal, void* d_rec_pre, void* d_p,
                        int npix, int dtype, void* st) {
    if (npix <= 0) DVAE_FAIL(DVAE_EINVAL, "recon_loss: empty");
    DISPATCH_T(dtype, hipLaunchKernelGGL((recon_loss_bwd_kernel<T>), dim3(grid_for(npix)), dim3(256), 0, S(st), rec,
                                         (const float*)p, (const T*)x, scal, (T*)d_rec_pre, (T*)d_p, npix, act_c(3, dtype));)
    DVAE_CHECK_LAUNCH("recon_loss_bwd");
    return DVAE_OK;
}

// ---------------------------------------------------------------------------------------------
int dvae_purify_quant(const void* a, const float* b, int a_is_f32, void* v_out, uint8_t* u8, int64_t first, int B, int HW,
                      int dtype, void* st) {
    if (B <= 0 || HW <= 0) DVAE_FAIL(DVAE_EINVAL, "purify_quant: empty");
    const int CS = act_c(3, dtype);
    const int g = grid_for((long)B * HW);
    const float* bf = (const float*)b;
    const int sig = a_is_f32 == 2 ? 1 : 0;               // a is the decoder's pre-activation: sigmoid first
    if (dtype == DVAE_F32)
        hipLaunchKernelGGL((purify_quant_kernel<float, float>), dim3(g), dim3(256), 0, S(st), (const float*)a, 4, bf, CS,
                           (float*)v_out, u8, (long)first, B, HW, sig);
    else if (dtype == DVAE_BF16 && a_is_f32)
        hipLaunchKernelGGL((purify_quant_kernel<float, bf16>), dim3(g), dim3(256), 0, S(st), (const float*)a, 4, bf, CS,
                           (bf16*)v_out, u8, (long)first, B, HW, sig);
    else if (dtype == DVAE_BF16)
        hipLaunchKernelGGL((purify_quant_kernel<bf16, bf16>), dim3(g), dim3(256), 0, S(st), (const bf16*)a, CS, bf, CS,
                           (bf16*)v_out, u8, (long)first, B, HW, 0);
    else
        DVAE_FAIL(DVAE_EDTYPE, "unknown dtype %d", dtype);
    DVAE_CHECK_LAUNCH("purify_quant");
    return DVAE_OK;
}

// ---------------------------------------------------------------------------------------------
int dvae_poison_u8(const uint8_t* src, const float* pert, const int64_t* labels, const uint8_t* mask, uint8_t* dst,
                   int64_t N, int HW, int prescaled, void* st) {
    if (!src || !pert || !dst || N <= 0 || HW <= 0) DVAE_FAIL(DVAE_EINVAL, "poison_u8: null argument / empty");
    hipLaunchKernelGGL(poison_u8_kernel, dim3(grid_for((long)N * HW)), dim3(256), 0, S(st), src, pert, labels, mask, dst,
                       (long)N, HW, prescaled);
    DVAE_CHECK_LAUNCH("poison_u8");
    return DVAE_OK;
}

// ---------------------------------------------------------------------------------------------
int dvae_sqnorm_partial(const float* g, int64_t n, float* partial, void* st) {
    if (n <= 0) DVAE_FAIL(DVAE_EINVAL, "sqnorm: empty");
    hipLaunchKernelGGL(sqnorm_partial_kernel, dim3(DVAE_NPARTIAL), dim3(256), 0, S(st), g, (long)n, partial);
    DVAE_CHECK_LAUNCH("sqnorm_partial");
    return DVAE_OK;
}

int dvae_clip_adam(float* p, const float* g, float* m, float* v, int64_t n, float* hyper, const float* partial,
                   float* scal, void* st) {
    if (n <= 0) DVAE_FAIL(DVAE_EINVAL, "clip_adam: empty");
    if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) DVAE_FAIL(DVAE_EINVAL, "clip_adam: buffers must be 16-byte aligned");
    hipLaunchKernelGGL(clip_adam_kernel, dim3(grid_for((n + 3) / 4, 256, 2048)), dim3(256), 0, S(st), p, g, m, v, (long)n, hyper,
                       partial, scal);
    DVAE_CHECK_LAUNCH("clip_adam");
    hipLaunchKernelGGL(hyper_tick_kernel, dim3(1), dim3(64), 0, S(st), hyper);
    DVAE_CHECK_LAUNCH("hyper_tick");
    return DVAE_OK;
}

}  // extern "C"
