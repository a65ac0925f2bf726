// The inference latent path in one launch: fc_mu | fc_var (1x1 conv 128 -> 2 x 256), the reparameterisation with in-kernel
// Philox noise, and post_conv (1x1 conv 256 -> 128) with its optional second output (+ class embedding).  All three are
// per-pixel operations, so a tile of 64 pixels goes through them without leaving the CU: mu, log_var AND z (the largest
// tensor of the forward, 512 bytes per pixel written and read back by the two-launch form) never exist in memory.
//
// The kernel is bound by the vector ALU -- Philox4x32-10, Box-Muller and exp for 256 latents per pixel, about 85 % of the
// stand-alone head kernel -- so everything else is arranged to hide under it: eight waves, two per SIMD, all weights in
// registers (inline-assembly MFMAs with an "a" operand, as conv_wreg.hpp: wave w keeps the mu and log_var rows of latents
// 32 w .. 32 w + 31, 16 A fragments, and the post_conv rows of output channels 32 (w & 3) .., 16 more), one barrier per
// tile.  Per tile a wave (1) computes mu and log_var of its 32 latents for both 32-pixel blocks (accumulators start from
// the bias), draws eps, and writes z = eps * exp(log_var / 2) + mu as bf16 into the tile's z image in LDS; (barrier) (2)
// drains the previous tile's output patch -- adding the class embedding to the pieces on their way out for the second
// output -- and computes post_conv for pixel block w >> 2 and channel block w & 3 from the z image into the patch.  The
// arithmetic and its order are those of the two kernels it replaces (conv_image.hpp's fused head, conv_point.hpp), so
// the outputs are the same bits.
#pragma once
#include <type_traits>

#include "conv_wreg.hpp"
#include "philox.hpp"

namespace dvae {

struct HeadPostArgs {
    const void* x;           // bf16 [P][128]: the stem's output, P = B * H * W pixels
    const void* w1;          // packed fc weights [512][K1tot]: mu rows, then log_var rows
    const float* b1;         // [512]
    const void* w2;          // packed post_conv weights [128][K2tot]
    const float* b2;         // [128], or null (0)
    void* out;               // bf16 [P][128]
    void* out2;              // bf16 [P][128] = out + emb[labels[image]][pixel], or null
    const bf16* emb;         // class embedding in activation layout: [classes][emb_px ? hw : 1][128]
    const int64_t* labels;   // [B]
    int emb_px;              // table rows per class: hw (spatial table) or 0 (one row for every pixel)
    int hw, ntiles, tiles_per_wg, K1tot, K2tot;
    int tpi, one_tpi;        // tiles per image and its branch-free division: n / tpi = umulhi(n, inv_tpi) + n * one_tpi
    unsigned inv_tpi;
    PhiloxKey zkey;
};

static constexpr int HP_XPIX = 256 + 16;                       // bytes per staged input pixel
static constexpr int HP_XBYTES = 64 * HP_XPIX;
static constexpr int HP_ZPIX = 512 + 16;                       // bytes per pixel of the z image
static constexpr int HP_ZBYTES = 64 * HP_ZPIX;
static constexpr int HP_OPIX = 256 + 8;                        // bytes per pixel of the output patch
static constexpr int HP_PATCH_BYTES = 64 * HP_OPIX + 128 + 128;     // a 128-byte gap after pixel 31 (bank spread)
static constexpr int HP_WROW = 512 + 16;                       // bytes per row of a wave's weight window
static constexpr int HP_TILE_LDS = 2 * HP_XBYTES + 2 * HP_ZBYTES + 2 * HP_PATCH_BYTES + (512 + 128) * 4;
static constexpr int HP_LDS_BYTES = 8 * 32 * HP_WROW > HP_TILE_LDS ? 8 * 32 * HP_WROW : HP_TILE_LDS;

__global__ __launch_bounds__(512) void head_post_bf16_kernel(const HeadPostArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* zimg = smem + 2 * HP_XBYTES;
    unsigned char* patch = zimg + 2 * HP_ZBYTES;
    float* ssl = reinterpret_cast<float*>(patch + 2 * HP_PATCH_BYTES);       // b1[512], b2[128]
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int ob = wave & 3, pb2 = wave >> 2;              // post_conv: this wave's channel block and pixel block
    const int t0 = blockIdx.x * a.tiles_per_wg;
    const int t1 = t0 + a.tiles_per_wg < a.ntiles ? t0 + a.tiles_per_wg : a.ntiles;
    if (t0 >= t1) return;

    // ---- weights: through a private LDS window per wave, copied as contiguous row pieces, straight-line, so that the
    // fragments are plain definitions the "a" operands can pin.  Lane (r, h) supplies row r of its block, K columns
    // 16 s + 8 h .. of step s.
    u32x4 w1r[2][8];                                       // [mu | log_var][K step]: rows 32 w + r (+ 256)
    u32x4 w2r[16];                                         // rows 32 (w & 3) + r
    {
        unsigned char* win = smem + wave * (32 * HP_WROW);
#pragma unroll
        for (int j = 0; j < 2; ++j) {                      // 32 rows x 128 columns
            const unsigned char* src = reinterpret_cast<const unsigned char*>(a.w1) + (long)(256 * j + 32 * wave) * a.K1tot * 2;
            const int wrow = lane >> 4, wcol = lane & 15;
            u32x4 wt[8];
#pragma unroll
            for (int i = 0; i < 8; ++i)
                wt[i] = *reinterpret_cast<const u32x4*>(src + (unsigned)((4 * i + wrow) * a.K1tot * 2 + wcol * 16));
#pragma unroll
            for (int i = 0; i < 8; ++i) *reinterpret_cast<u32x4*>(win + (4 * i + wrow) * HP_WROW + wcol * 16) = wt[i];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int s = 0; s < 8; ++s) w1r[j][s] = *reinterpret_cast<const u32x4*>(win + r * HP_WROW + 32 * s + 16 * h);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        {                                                  // 32 rows x 256 columns
            const unsigned char* src = reinterpret_cast<const unsigned char*>(a.w2) + (long)(32 * ob) * a.K2tot * 2;
            const int wrow = lane >> 5, wcol = lane & 31;
            u32x4 wt[16];
#pragma unroll
            for (int i = 0; i < 16; ++i)
                wt[i] = *reinterpret_cast<const u32x4*>(src + (unsigned)((2 * i + wrow) * a.K2tot * 2 + wcol * 16));
#pragma unroll
            for (int i = 0; i < 16; ++i) *reinterpret_cast<u32x4*>(win + (2 * i + wrow) * HP_WROW + wcol * 16) = wt[i];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int s = 0; s < 16; ++s) w2r[s] = *reinterpret_cast<const u32x4*>(win + r * HP_WROW + 32 * s + 16 * h);
        }
    }
    __syncthreads();                                       // the windows are dead: the tile buffers take their place
    ssl[tid] = a.b1[tid];
    if (tid < 128) ssl[512 + tid] = a.b2 ? a.b2[tid] : 0.f;

    // ---- staging: a tile is 64 consecutive pixels = 16 KB contiguous: vector v = tid + 512 i, pixel v >> 4, chunk v & 15
    const unsigned char* xin = reinterpret_cast<const unsigned char*>(a.x);
    u32x4 xreg[2];
    auto load_x = [&](int tile) {
#pragma unroll
        for (int i = 0; i < 2; ++i) xreg[i] = *reinterpret_cast<const u32x4*>(xin + (long)tile * 16384 + (tid + 512 * i) * 16);
    };
    auto store_x = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
            *reinterpret_cast<u32x4*>(smem + buf * HP_XBYTES + ((tid >> 4) + 32 * i) * HP_XPIX + (tid & 15) * 16) = xreg[i];
    };
    // the output patch: pixel p at p * 264 + (p >> 5) * 128; thread (tid >> 4, chunk tid & 15) drains pixels tid >> 4 and
    // (tid >> 4) + 32 as whole 16-byte pieces (the patch's pixel pitch is 8 mod 16: two 8-byte reads)
    auto drain = [&](int tile, int pbuf) {
        const int img = (int)__umulhi((unsigned)tile, a.inv_tpi) + tile * a.one_tpi;
        const int px0 = (tile - img * a.tpi) * 64;          // first pixel of the tile inside its image
        const long lab = a.out2 ? (long)a.labels[img] : 0l;
        unsigned char* ob1 = reinterpret_cast<unsigned char*>(a.out) + (long)tile * 16384;
        unsigned char* ob2 = reinterpret_cast<unsigned char*>(a.out2) + (long)tile * 16384;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int p = (tid >> 4) + 32 * i;
            u32x4 ev = {0u, 0u, 0u, 0u};
            if (a.out2)
                ev = *reinterpret_cast<const u32x4*>(a.emb + (a.emb_px ? lab * a.hw + px0 + p : lab) * 128 + (tid & 15) * 8);
            const unsigned char* s = patch + pbuf * HP_PATCH_BYTES + p * HP_OPIX + i * 128 + (tid & 15) * 16;
            const u32x2 lo = *reinterpret_cast<const u32x2*>(s), hi = *reinterpret_cast<const u32x2*>(s + 8);
            const u32x4 v = {lo[0], lo[1], hi[0], hi[1]};
            *reinterpret_cast<u32x4*>(ob1 + (tid + 512 * i) * 16) = v;
            if (a.out2) {
                const bf16x8 x = __builtin_bit_cast(bf16x8, v), e = __builtin_bit_cast(bf16x8, ev);
                bf16x8 o2;
#pragma unroll
                for (int c = 0; c < 8; ++c) o2[c] = (bf16)((float)x[c] + (float)e[c]);   // the stored value + the embedding
                *reinterpret_cast<bf16x8*>(ob2 + (tid + 512 * i) * 16) = o2;
            }
        }
    };

    load_x(t0);
    store_x(0);
    load_x(t0 + 1 < t1 ? t0 + 1 : t0);
    __syncthreads();
    int it = 0;
#pragma nounroll
    for (int tile = t0; tile < t1; ++tile, ++it) {
        const int cur = it & 1;
        // (1) the next tile's pixels into the other buffer, the one after into registers; then mu, log_var, z
        store_x(cur ^ 1);
        load_x(tile + 2 < t1 ? tile + 2 : t1 - 1);
        const int img_i = (int)__umulhi((unsigned)tile, a.inv_tpi) + tile * a.one_tpi;
        const int px0 = (tile - img_i * a.tpi) * 64;
        const uint64_t img = a.zkey.index ? (uint64_t)a.zkey.index[img_i] : (uint64_t)(a.zkey.first + img_i);
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) {
            f32x16 acc[2];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 sh = *reinterpret_cast<const f32x4*>(ssl + 256 * j + 32 * wave + 8 * g + 4 * h);
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[j][4 * g + k] = sh[k];     // the accumulators start from the bias
                }
            const unsigned char* xrow = smem + cur * HP_XBYTES + (32 * pb + r) * HP_XPIX + 16 * h;
            u32x4 xv[8];
#pragma unroll
            for (int s = 0; s < 8; ++s) xv[s] = *reinterpret_cast<const u32x4*>(xrow + 32 * s);
            asm volatile("s_nop 4" : "+v"(acc[0]), "+v"(acc[1]));   // the bias moves -> the MFMAs' C operand (tied to the registers: nothing moves across)
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                mfma_acc_a(acc[0], w1r[0][s], xv[s]);
                mfma_acc_a(acc[1], w1r[1][s], xv[s]);
            }
            unsigned e0 = (unsigned)(px0 + 32 * pb + r) * 256u + 32u * (unsigned)wave;   // first element of this lane pair's 32
            asm volatile("" : "+v"(e0));                   // (the first Philox round would otherwise be hoisted out of the loop)
            f32x4 e[4];
            philox_lane_quads(a.zkey, img, e0, h, e);
            asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc[0]), "+v"(acc[1]));   // the last MFMAs' results -> the VALU reads below
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16x4 o;
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = (bf16)(e[g][k] * __expf(0.5f * acc[1][4 * g + k]) + acc[0][4 * g + k]);
                *reinterpret_cast<bf16x4*>(zimg + cur * HP_ZBYTES + (32 * pb + r) * HP_ZPIX + (32 * wave + 8 * g + 4 * h) * 2) = o;
            }
        }
        __syncthreads();                                   // z complete, next pixels staged, the previous patch complete
        // (2) the previous tile's patch leaves (the first time: whatever the patch holds, to this tile's own place, which
        // its real drain overwrites one barrier later: same thread, same addresses); post_conv of this tile into the other
        drain(it == 0 ? tile : tile - 1, cur ^ 1);
        {
            f32x16 acc2;
            const unsigned char* zrow = zimg + cur * HP_ZBYTES + (32 * pb2 + r) * HP_ZPIX + 16 * h;
            u32x4 zv[4];
#pragma unroll
            for (int s = 0; s < 3; ++s) zv[s] = *reinterpret_cast<const u32x4*>(zrow + 32 * s);
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                if (s + 3 < 16) zv[(s + 3) & 3] = *reinterpret_cast<const u32x4*>(zrow + 32 * (s + 3));
                if (s == 0) mfma_first_a(acc2, w2r[s], zv[s & 3]);
                else mfma_acc_a(acc2, w2r[s], zv[s & 3]);
            }
            asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc2));
            unsigned char* prow = patch + cur * HP_PATCH_BYTES + (32 * pb2 + r) * HP_OPIX + pb2 * 128 + (32 * ob + 4 * h) * 2;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 sh = *reinterpret_cast<const f32x4*>(ssl + 512 + 32 * ob + 8 * g + 4 * h);
                bf16x4 o;
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = (bf16)(acc2[4 * g + k] * 1.0f + sh[k]);
                *reinterpret_cast<bf16x4*>(prow + 16 * g) = o;
            }
        }
    }
    __syncthreads();
    drain(t1 - 1, (it - 1) & 1);
}

// true if (fc plan, post_conv plan, L) is a launch of the kernel above
static bool head_post_applicable(const GemmPlan& P1, const GemmPlan& P2, int L) {
    if (L != 256 || P1.nphase != 1 || P2.nphase != 1) return false;
    if (P1.CgS != 128 || P1.Cg != 128 || P1.N != 512 || P1.Ktot != 128 || P1.ph[0].ntaps != 1 || P1.ph[0].koff != 0) return false;
    if (P2.CgS != 256 || P2.Cg != 256 || P2.N != 128 || P2.NS != 128 || P2.Ktot != 256 || P2.ph[0].ntaps != 1 || P2.ph[0].koff != 0) return false;
    if (P1.ph[0].SY != 1 || P1.ph[0].SX != 1 || P2.ph[0].SY != 1 || P2.ph[0].SX != 1) return false;
    if (P1.Hout != P1.Hin || P1.Wout != P1.Win || P2.Hin != P1.Hin || P2.Win != P1.Win || P2.Hout != P1.Hin || P2.Wout != P1.Win || P2.B != P1.B)
        return false;
    const long hw = (long)P1.Hin * P1.Win, P = hw * P1.B;
    if (hw % 64 || hw * 256 >= 0x7fffffffL) return false;  // a tile lies inside one image; element indices are 32-bit
    return P / 64 < 0x7fffffffL / 64 && (P / 64) * (hw / 64) < (1ll << 32);
}

static void launch_head_post(const GemmPlan& P1, const GemmPlan& P2, const void* x, const void* w1, const float* b1, const void* w2,
                             const float* b2, void* out, const void* emb, const int64_t* labels, void* out2, int emb_plain,
                             const PhiloxKey& key, hipStream_t st) {
    HeadPostArgs a;
    a.x = x; a.w1 = w1; a.b1 = b1; a.w2 = w2; a.b2 = b2; a.out = out; a.out2 = out2;
    a.emb = (const bf16*)emb; a.labels = labels;
    a.hw = P1.Hin * P1.Win; a.emb_px = emb_plain ? 0 : a.hw;
    a.ntiles = (int)((long)P1.B * a.hw / 64);
    a.K1tot = P1.Ktot; a.K2tot = P2.Ktot;
    a.tpi = a.hw / 64; a.one_tpi = a.tpi == 1;
    a.inv_tpi = a.tpi > 1 ? (unsigned)(((1ull << 32) + a.tpi - 1) / a.tpi) : 0u;
    a.zkey = key;
    a.tiles_per_wg = (a.ntiles + 255) / 256;               // one workgroup per CU
    const int gx = (a.ntiles + a.tiles_per_wg - 1) / a.tiles_per_wg;
    static LdsOptIn lds_opt;
    lds_opt.ensure(reinterpret_cast<const void*>(&head_post_bf16_kernel), HP_LDS_BYTES);
    hipLaunchKernelGGL(head_post_bf16_kernel, dim3(gx), dim3(512), HP_LDS_BYTES, st, a);
}

}  // namespace dvae
