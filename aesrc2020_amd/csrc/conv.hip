// ResNet conv stack for SAR-Net on gfx950: NHWC fp32 implicit GEMM on v_mfma_f32_32x32x2_f32.
//   fwd  : y = conv(P(x), w) + bias [+ residual]          P(x) = relu(x*scale[c]+shift[c]) (the preceding
//   dgrad: dP = conv_transpose(dy, w)                            BatchNorm+ReLU folded into the A-operand load;
//   wgrad: dw += P(x)^T (*) dy   (split over output rows,        zero padding is applied AFTER P, as Keras does)
//                                  fp32 atomics into the flat gradient buffer)
// Reference: resnet.py:22-125 (_bn_relu / _bn_relu_conv / basic_block / _shortcut), TF-SAME padding per
// SURVEY.md App. A.  The 7x7/2 Cin=1 stem (resnet.py:173/191) is a direct LDS-patch kernel; MaxPooling2D
// (resnet.py:174) is fused with the stem's BN+ReLU.
#include "common.h"
#include <stdlib.h>

namespace {

// ------------------------------------------------------------------ stem 7x7/2, Cin = 1
constexpr int ST_TH = 4, ST_TW = 16;                 // output tile 4 x 16
constexpr int ST_PH = (ST_TH - 1) * 2 + 7, ST_PW = (ST_TW - 1) * 2 + 7;   // 13 x 37 input patch

constexpr int ST_PLD = ST_PW + 1;                    // 38: patch row stride
constexpr int ST_ZSLOT = ST_PH * ST_PLD;             // one zero word after the patch (k = 49 padding / invalid taps)

__device__ __forceinline__ void stem_load_patch_flat(float* patch, const float* x, int n, int H, int W, int hi0, int wi0) {
    for (int i = threadIdx.x; i < ST_PH * ST_PW; i += blockDim.x) {
        const int r = i / ST_PW, c = i - r * ST_PW;
        const int hi = hi0 + r, wi = wi0 + c;
        patch[r * ST_PLD + c] = (hi >= 0 && hi < H && wi >= 0 && wi < W) ? x[((long)n * H + hi) * W + wi] : 0.f;
    }
    if (threadIdx.x == 0) patch[ST_ZSLOT] = 0.f;
}

// y[n,ho,wo,c] = sum_{kh,kw} x[n,2ho-pt+kh,2wo-pl+kw] * w[kh,kw,0,c] + b[c];  C0 <= 64, C0 % 4 == 0.
// GEMM view per block: M = 64 output pixels (4 x 16 tile), N = 64 channels, K = 49 taps padded to 50; the A operand is
// gathered from the 13 x 37 input patch in LDS, B = the 7x7xC0 kernel in LDS; 4 waves = 2 pixel blocks x 2 channel blocks
// of v_mfma_f32_32x32x2_f32 (25 MFMAs per wave).
// stats (nullable): [stat_rows][2*C0] fp32, zeroed by the launcher -- per-channel sum and sum of squares of y, added with
// atomics to row blockIdx % stat_rows (a few adds per address); sar_bn_finalize_f32 turns them into the batch statistics of
// the BatchNorm behind the stem, which saves the separate statistics pass over the 131 MB map (B=64).
__global__ __launch_bounds__(256) void stem_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ bias, float* __restrict__ y, int B,
                                                       int H, int W, int Ho, int Wo, int C0, int pt, int pl,
                                                       float* __restrict__ stats, int stat_rows) {
    __shared__ float patch[ST_ZSLOT + 2];
    __shared__ float ws[50 * 64];
    const int tiles_w = (Wo + ST_TW - 1) / ST_TW, tiles_h = (Ho + ST_TH - 1) / ST_TH;
    int b = blockIdx.x;
    const int tw = b % tiles_w; b /= tiles_w;
    const int th = b % tiles_h; const int n = b / tiles_h;
    // all 13 kernel loads of a thread are issued before the first LDS store (clamped addresses, select afterwards): as a plain
    // loop this was 13 L2 round trips in front of every block's 25 MFMAs
    {
        float wv[13];
#pragma unroll
        for (int j = 0; j < 13; ++j) {
            const int i = threadIdx.x + 256 * j, k = i >> 6, c = i & 63;
            wv[j] = w[min(k, 48) * C0 + min(c, C0 - 1)];
        }
#pragma unroll
        for (int j = 0; j < 13; ++j) {
            const int i = threadIdx.x + 256 * j, k = i >> 6, c = i & 63;
            if (i < 50 * 64) ws[i] = (k < 49 && c < C0) ? wv[j] : 0.f;
        }
    }
    stem_load_patch_flat(patch, x, n, H, W, th * ST_TH * 2 - pt, tw * ST_TW * 2 - pl);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int mb = wave >> 1, nb = wave & 1, kr = lane >> 5, col = lane & 31;
    if (nb * 32 >= C0) return;                                      // wave-uniform (no barriers below)
    const int pix = mb * 32 + col;
    const int pbase = ((pix >> 4) * 2) * ST_PLD + (pix & 15) * 2;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
    for (int s2 = 0; s2 < 25; ++s2) {
        constexpr int dummy = 0; (void)dummy;
        const int k0 = 2 * s2, k1 = 2 * s2 + 1;
        const int offA = (k0 / 7) * ST_PLD + (k0 % 7);
        const int offB = (k1 / 7) * ST_PLD + (k1 % 7);
        const int ia = kr ? (k1 < 49 ? pbase + offB : ST_ZSLOT) : pbase + offA;
        const float a = patch[ia];
        const float bv = ws[(k0 + kr) * 64 + nb * 32 + col];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc, 0, 0, 0);
    }
    const int c = nb * 32 + col;
    if (c >= C0) return;
    const float bz = bias[c];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int prow = mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * kr;
        const int ho = th * ST_TH + (prow >> 4), wo = tw * ST_TW + (prow & 15);
        if (ho < Ho && wo < Wo) {
            const float v = acc[r] + bz;
            y[(((long)n * Ho + ho) * Wo + wo) * C0 + c] = v;
            s1 += v; s2 = fmaf(v, v, s2);
        }
    }
    if (stats) {                                                    // kernel-uniform; lanes col and col+32 hold the same channel
        s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
        if (kr == 0) {
            float* o = stats + (long)(blockIdx.x % stat_rows) * 2 * C0;
            atomicAdd(o + c, s1); atomicAdd(o + C0 + c, s2);
        }
    }
}

// dw[tap][c] += sum_pixels patch * dy ; db[c] += sum dy.  GEMM view: M = 49 taps (padded to 64), N = C0, K = pixels;
// persistent blocks walk the 4 x 16 output tiles and keep the 32x32 accumulators in registers; one atomic flush at the end.
// FUSED: dy is not read but computed per tile -- it is the gradient of [BatchNorm -> ReLU -> MaxPool 3x3/2] at the stem output
// (what sar_maxpool_bn_relu_bwd_f32's apply pass would write): dz from the (at most four) pooling windows a pixel lies in,
// then dx = scale*(dz - s1/R - xhat*s2/R).  Nothing else reads that full-resolution gradient (the network input needs
// none), so fusing it here removes a 131 MB write and a 131 MB read at B=64 (apply 83 us + wgrad 82 us before).
struct PoolBwd {
    const float* dpool; const float* out; const uint8_t* idx; const float* ys; const float* mean; const float* rstd;
    const float* scale; const double* sums; int Hp, Wp, ppt, ppl;
};
template <bool FUSED>
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                         float* __restrict__ dw, float* __restrict__ db, int B, int H,
                                                         int W, int Ho, int Wo, int C0, int pt, int pl, PoolBwd pb) {
    __shared__ float patch[ST_ZSLOT + 2];
    __shared__ __attribute__((aligned(16))) float dys[64 * 64];
    // FUSED: the 4 x 16 pixels of a tile lie in a window of 3 x 9 pooling outputs: their gradients (already masked with the
    // ReLU of the pooled value) and argmax slots are staged once per tile, and every pixel picks its <= 4 candidates from LDS.
    // (Measured alternatives: candidates straight from global memory 149 us; a scatter from the pooled window with LDS atomics
    // 159 us; this form 132 us -- against 83 + 82 us for the separate apply and weight-gradient kernels.)
    constexpr int PWH = ST_TH / 2 + 1, PWW = ST_TW / 2 + 1, PWN = PWH * PWW;
    __shared__ __attribute__((aligned(16))) float gms[FUSED ? PWN * 64 : 4];
    __shared__ uchar4 ids[FUSED ? PWN * 16 : 1];
    const int tiles_w = (Wo + ST_TW - 1) / ST_TW, tiles_h = (Ho + ST_TH - 1) / ST_TH;
    const int ntiles = B * tiles_h * tiles_w;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int mb = wave >> 1, nb = wave & 1, kr = lane >> 5, col = lane & 31;
    const int tap = mb * 32 + col;
    const bool tap_ok = tap < 49;
    const int toff = tap_ok ? (tap / 7) * ST_PLD + (tap % 7) : 0;
    const bool active = nb * 32 < C0;                               // wave-uniform
    const int C4 = C0 / 4;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float accb = 0.f;
    // FUSED: per-thread channel constants (the thread's channel vector c4 = threadIdx.x & 15 is the same for all its elements)
    float4 f_mu = make_float4(0, 0, 0, 0), f_rs = f_mu, f_sc = f_mu, f_m1 = f_mu, f_m2 = f_mu;
    if constexpr (FUSED) {
        const int c = (threadIdx.x & 15) * 4;
        if (c < C0) {
            const double invR = 1.0 / ((double)B * Ho * Wo);
            f_mu = *reinterpret_cast<const float4*>(pb.mean + c); f_rs = *reinterpret_cast<const float4*>(pb.rstd + c);
            f_sc = *reinterpret_cast<const float4*>(pb.scale + c);
            f_m1 = make_float4((float)(pb.sums[c] * invR), (float)(pb.sums[c + 1] * invR), (float)(pb.sums[c + 2] * invR), (float)(pb.sums[c + 3] * invR));
            f_m2 = make_float4((float)(pb.sums[C0 + c] * invR), (float)(pb.sums[C0 + c + 1] * invR), (float)(pb.sums[C0 + c + 2] * invR), (float)(pb.sums[C0 + c + 3] * invR));
        }
    }
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int b = tile;
        const int tw = b % tiles_w; b /= tiles_w;
        const int th = b % tiles_h; const int n = b / tiles_h;
        __syncthreads();
        stem_load_patch_flat(patch, x, n, H, W, th * ST_TH * 2 - pt, tw * ST_TW * 2 - pl);
        const int pr0 = (th * ST_TH + pb.ppt - 1) >> 1, pc0 = (tw * ST_TW + pb.ppl - 1) >> 1;     // floor: -1 at the top / left edge
        if constexpr (!FUSED) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {                            // dys[pix][0..63], float4 lanes; columns >= C0 are zero
                const int i = threadIdx.x + 256 * j, pixel = i >> 4, c4 = i & 15;
                const int ho = th * ST_TH + (pixel >> 4), wo = tw * ST_TW + (pixel & 15);
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (c4 < C4 && ho < Ho && wo < Wo) v = *reinterpret_cast<const float4*>(dy + (((long)n * Ho + ho) * Wo + wo) * C0 + c4 * 4);
                *reinterpret_cast<float4*>(&dys[pixel * 64 + c4 * 4]) = v;
            }
        } else {
            float4 xvs[4];                                           // the stem outputs of this thread's four elements: in flight
#pragma unroll                                                       // under the window staging
            for (int j = 0; j < 4; ++j) {
                const int i = threadIdx.x + 256 * j, pixel = i >> 4, c4 = i & 15;
                const int ho = th * ST_TH + (pixel >> 4), wo = tw * ST_TW + (pixel & 15);
                xvs[j] = (c4 < C4 && ho < Ho && wo < Wo) ? *reinterpret_cast<const float4*>(pb.ys + (((long)n * Ho + ho) * Wo + wo) * C0 + c4 * 4)
                                                          : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            for (int i = threadIdx.x; i < PWN * 16; i += 256) {      // stage the pooled window
                const int pp = i >> 4, c4 = i & 15;
                const int pr = pr0 + pp / PWW, pc = pc0 + pp % PWW;
                float4 gm = make_float4(0.f, 0.f, 0.f, 0.f);
                uchar4 id = make_uchar4(255, 255, 255, 255);
                if (c4 < C4 && pr >= 0 && pr < pb.Hp && pc >= 0 && pc < pb.Wp) {
                    const long o = (((long)n * pb.Hp + pr) * pb.Wp + pc) * C0 + c4 * 4;
                    id = *reinterpret_cast<const uchar4*>(pb.idx + o);
                    const float4 ov = *reinterpret_cast<const float4*>(pb.out + o), g = *reinterpret_cast<const float4*>(pb.dpool + o);
                    gm = make_float4(ov.x > 0.f ? g.x : 0.f, ov.y > 0.f ? g.y : 0.f, ov.z > 0.f ? g.z : 0.f, ov.w > 0.f ? g.w : 0.f);
                }
                *reinterpret_cast<float4*>(&gms[pp * 64 + c4 * 4]) = gm; ids[i] = id;
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = threadIdx.x + 256 * j, pixel = i >> 4, c4 = i & 15, c = c4 * 4;
                const int ho = th * ST_TH + (pixel >> 4), wo = tw * ST_TW + (pixel & 15);
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (c4 < C4 && ho < Ho && wo < Wo) {
                    // same arithmetic as maxpool_bn_relu_bwd_apply_kernel: window (a>>1) sees the pixel at row a&1 and, when a is
                    // even, window (a>>1)-1 sees it at row 2
                    const float4 xv = xvs[j];
                    const int a = ho + pb.ppt, bb = wo + pb.ppl;
                    const int ho1 = a >> 1, wo1 = bb >> 1;
                    const int hos[2] = {ho1, ho1 - 1}, khs[2] = {a & 1, 2}, wos[2] = {wo1, wo1 - 1}, kws[2] = {bb & 1, 2};
                    const bool vh[2] = {ho1 < pb.Hp, (a & 1) == 0 && ho1 >= 1 && ho1 - 1 < pb.Hp};
                    const bool vw[2] = {wo1 < pb.Wp, (bb & 1) == 0 && wo1 >= 1 && wo1 - 1 < pb.Wp};
                    float dz[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                        for (int jj = 0; jj < 2; ++jj) {
                            const int lr = min(max(hos[ii] - pr0, 0), PWH - 1), lc = min(max(wos[jj] - pc0, 0), PWW - 1);
                            const int pp = lr * PWW + lc;
                            const uchar4 id = ids[pp * 16 + c4];
                            const float4 g = *reinterpret_cast<const float4*>(&gms[pp * 64 + c]);
                            const int slot = vh[ii] && vw[jj] ? khs[ii] * 3 + kws[jj] : 254;         // 254 / 255: never equal
                            if (id.x == slot) dz[0] += g.x;
                            if (id.y == slot) dz[1] += g.y;
                            if (id.z == slot) dz[2] += g.z;
                            if (id.w == slot) dz[3] += g.w;
                        }
                    v.x = f_sc.x * (dz[0] - f_m1.x - (xv.x - f_mu.x) * f_rs.x * f_m2.x);
                    v.y = f_sc.y * (dz[1] - f_m1.y - (xv.y - f_mu.y) * f_rs.y * f_m2.y);
                    v.z = f_sc.z * (dz[2] - f_m1.z - (xv.z - f_mu.z) * f_rs.z * f_m2.z);
                    v.w = f_sc.w * (dz[3] - f_m1.w - (xv.w - f_mu.w) * f_rs.w * f_m2.w);
                }
                *reinterpret_cast<float4*>(&dys[pixel * 64 + c4 * 4]) = v;
            }
        }
        __syncthreads();
        if (active) {
#pragma unroll
            for (int s2 = 0; s2 < 32; ++s2) {
                const int p0 = 2 * s2, p1 = 2 * s2 + 1;
                const int pbA = ((p0 >> 4) * 2) * ST_PLD + (p0 & 15) * 2, pbB = ((p1 >> 4) * 2) * ST_PLD + (p1 & 15) * 2;
                const int ia = tap_ok ? (kr ? pbB : pbA) + toff : ST_ZSLOT;
                const float a = patch[ia];
                const float bv = dys[(p0 + kr) * 64 + nb * 32 + col];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc, 0, 0, 0);
            }
        }
        if (db && threadIdx.x < C0) {
            for (int pixel = 0; pixel < 64; ++pixel) accb += dys[pixel * 64 + threadIdx.x];
        }
    }
    const int c = nb * 32 + col;
    if (active && c < C0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int t = mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * kr;
            if (t < 49) atomicAdd(dw + t * C0 + c, acc[r]);
        }
    }
    if (db && threadIdx.x < C0) atomicAdd(db + threadIdx.x, accb);
}

// ------------------------------------------------------------------ BN + ReLU + MaxPool 3x3/2 SAME (fused)
// out = maxpool(relu(x*scale+shift)); idx = window position (0..8) of the first maximum.
__global__ void bn_relu_maxpool_fwd_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                           const float* __restrict__ shift, float* __restrict__ out,
                                           uint8_t* __restrict__ idx, int B, int H, int W, int C, int Ho, int Wo,
                                           int pt, int pl) {
    const int C4 = C / 4;
    const long total = (long)B * Ho * Wo * C4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4) * 4; long t = i / C4;
        const int wo = (int)(t % Wo); t /= Wo;
        const int ho = (int)(t % Ho); const int n = (int)(t / Ho);
        const float4 sc = *reinterpret_cast<const float4*>(scale + c), sh = *reinterpret_cast<const float4*>(shift + c);
        float m[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        int mi[4] = {0, 0, 0, 0};
        for (int kh = 0; kh < 3; ++kh) {
            const int hi = ho * 2 - pt + kh;
            if (hi < 0 || hi >= H) continue;
            for (int kw = 0; kw < 3; ++kw) {
                const int wi = wo * 2 - pl + kw;
                if (wi < 0 || wi >= W) continue;
                const float4 v = *reinterpret_cast<const float4*>(x + (((long)n * H + hi) * W + wi) * C + c);
                const float a[4] = {fmaxf(fmaf(v.x, sc.x, sh.x), 0.f), fmaxf(fmaf(v.y, sc.y, sh.y), 0.f),
                                    fmaxf(fmaf(v.z, sc.z, sh.z), 0.f), fmaxf(fmaf(v.w, sc.w, sh.w), 0.f)};
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (a[j] > m[j]) { m[j] = a[j]; mi[j] = kh * 3 + kw; }
            }
        }
        const long o = (((long)n * Ho + ho) * Wo + wo) * C + c;
        *reinterpret_cast<float4*>(out + o) = make_float4(m[0], m[1], m[2], m[3]);
        *reinterpret_cast<uchar4*>(idx + o) = make_uchar4(mi[0], mi[1], mi[2], mi[3]);
    }
}

// Backward pass 1: per-channel sums of dz and dz*xhat over the full-resolution tensor, where
// dz[n,h,w,c] = sum of dpool over windows whose argmax is (h,w) and whose max is > 0 (ReLU active).
// float4 lanes over channel-vectors x pixel groups; per-block partials (combined in fp64 by pool_combine_kernel).
__global__ __launch_bounds__(256) void maxpool_bn_relu_bwd_reduce_kernel(
    const float* __restrict__ dpool, const float* __restrict__ out, const uint8_t* __restrict__ idx,
    const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
    const float* __restrict__ scale, const float* __restrict__ shift,
    float* __restrict__ part, int B, int H, int W, int C, int Ho, int Wo, int pt, int pl, unsigned magic_wo, unsigned magic_ho) {
    __shared__ float4 r1[256], r2[256];
    const int tpr = C / 4;                                    // C <= 256 and 256 % C == 0 (host-checked) -> tpr | 64
    const int cv = threadIdx.x % tpr, pg = threadIdx.x / tpr, npg = 256 / tpr;
    const int c = cv * 4;
    const unsigned npix = (unsigned)(B * Ho * Wo);
    const float4 mu = *reinterpret_cast<const float4*>(mean + c), rs = *reinterpret_cast<const float4*>(rstd + c);
    const float4 scv = *reinterpret_cast<const float4*>(scale + c), sh = *reinterpret_cast<const float4*>(shift + c);
    struct B4 { bool x, y, z, w; };
    const B4 ok = {fabsf(scv.x) > 1e-6f, fabsf(scv.y) > 1e-6f, fabsf(scv.z) > 1e-6f, fabsf(scv.w) > 1e-6f};
    const float4 isc = make_float4(ok.x ? 1.f / scv.x : 0.f, ok.y ? 1.f / scv.y : 0.f, ok.z ? 1.f / scv.z : 0.f, ok.w ? 1.f / scv.w : 0.f);
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
    for (unsigned pix = blockIdx.x * npg + pg; pix < npix; pix += gridDim.x * npg) {
        const long o = (long)pix * C + c;
        const float4 a = *reinterpret_cast<const float4*>(out + o);
        const float4 g = *reinterpret_cast<const float4*>(dpool + o);
        const uchar4 k = *reinterpret_cast<const uchar4*>(idx + o);
        const unsigned t = Wo == 1 ? pix : __umulhi(pix, magic_wo), wo = pix - t * (unsigned)Wo;      // q = umulhi(m, ceil(2^32/d))
        const unsigned n = Ho == 1 ? t : __umulhi(t, magic_ho), ho = t - n * (unsigned)Ho;
        const long base = (((long)n * H + (int)ho * 2 - pt) * W + (int)wo * 2 - pl) * C + c;
        // xhat at the argmax.  Where ReLU is active the pooled value IS scale*x + shift, so x (and xhat) follow from it without
        // touching the full-resolution tensor (the gather costs a 32-byte sector per element); only a degenerate channel
        // (|scale| tiny) falls back to the gather.
        if (a.x > 0.f) { const float xv = ok.x ? (a.x - sh.x) * isc.x : x[base + ((k.x / 3) * W + (k.x % 3)) * C + 0]; const float xh = (xv - mu.x) * rs.x; s1.x += g.x; s2.x = fmaf(g.x, xh, s2.x); }
        if (a.y > 0.f) { const float xv = ok.y ? (a.y - sh.y) * isc.y : x[base + ((k.y / 3) * W + (k.y % 3)) * C + 1]; const float xh = (xv - mu.y) * rs.y; s1.y += g.y; s2.y = fmaf(g.y, xh, s2.y); }
        if (a.z > 0.f) { const float xv = ok.z ? (a.z - sh.z) * isc.z : x[base + ((k.z / 3) * W + (k.z % 3)) * C + 2]; const float xh = (xv - mu.z) * rs.z; s1.z += g.z; s2.z = fmaf(g.z, xh, s2.z); }
        if (a.w > 0.f) { const float xv = ok.w ? (a.w - sh.w) * isc.w : x[base + ((k.w / 3) * W + (k.w % 3)) * C + 3]; const float xh = (xv - mu.w) * rs.w; s1.w += g.w; s2.w = fmaf(g.w, xh, s2.w); }
    }
    r1[threadIdx.x] = s1; r2[threadIdx.x] = s2;
    __syncthreads();
    if (threadIdx.x < tpr) {
        float4 a1 = make_float4(0.f, 0.f, 0.f, 0.f), a2 = a1;
        for (int g = 0; g < npg; ++g) {
            const float4 u = r1[g * tpr + threadIdx.x], v = r2[g * tpr + threadIdx.x];
            a1.x += u.x; a1.y += u.y; a1.z += u.z; a1.w += u.w; a2.x += v.x; a2.y += v.y; a2.z += v.z; a2.w += v.w;
        }
        float* o = part + (long)blockIdx.x * 2 * C;
        *reinterpret_cast<float4*>(o + c) = a1; *reinterpret_cast<float4*>(o + C + c) = a2;
    }
}

__global__ __launch_bounds__(256) void pool_combine_kernel(const float* __restrict__ part, int nb, double* __restrict__ totals, int C2) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;     // one wave per column
    if (c >= C2) return;
    double s = 0;
    for (int b = lane; b < nb; b += 64) s += (double)part[(long)b * C2 + c];
    s = wave_sum_d(s);
    if (lane == 0) totals[c] = s;
}

// Backward pass 2: dx = scale * (dz - s1/R - xhat*s2/R) at full resolution; 4 channels per thread (float4 / uchar4).
// Thread = (pixel group, channel vector): the per-channel constants are hoisted out of the pixel loop and the pixel index is
// decomposed with two magic-number divisions (the first version spent most of its time in five 64-bit divisions per element).
__global__ __launch_bounds__(256) void maxpool_bn_relu_bwd_apply_kernel(const float* __restrict__ dpool, const float* __restrict__ out,
                                                 const uint8_t* __restrict__ idx, const float* __restrict__ x,
                                                 const float* __restrict__ mean, const float* __restrict__ rstd,
                                                 const float* __restrict__ scale, const double* __restrict__ sums,
                                                 float* __restrict__ dx, int B, int H, int W, int C, int Ho, int Wo,
                                                 int pt, int pl, unsigned magic_w, unsigned magic_h) {
    const int tpr = C / 4;                                    // C <= 256 and 256 % C == 0 (host-checked)
    const int cv = threadIdx.x % tpr, pg = threadIdx.x / tpr, npg = 256 / tpr;
    const int c = cv * 4;
    const unsigned npix = (unsigned)(B * H * W);
    const double invR = 1.0 / (double)npix;
    const float4 mu = *reinterpret_cast<const float4*>(mean + c), rs = *reinterpret_cast<const float4*>(rstd + c);
    const float4 sc = *reinterpret_cast<const float4*>(scale + c);
    const float4 m1 = make_float4((float)(sums[c] * invR), (float)(sums[c + 1] * invR), (float)(sums[c + 2] * invR), (float)(sums[c + 3] * invR));
    const float4 m2 = make_float4((float)(sums[C + c] * invR), (float)(sums[C + c + 1] * invR), (float)(sums[C + c + 2] * invR), (float)(sums[C + c + 3] * invR));
    for (unsigned pix = blockIdx.x * npg + pg; pix < npix; pix += gridDim.x * npg) {
        const unsigned t = W == 1 ? pix : __umulhi(pix, magic_w); const int w_ = (int)(pix - t * (unsigned)W);
        const unsigned n = H == 1 ? t : __umulhi(t, magic_h); const int h = (int)(t - n * (unsigned)H);
        // A pixel lies in at most 2 x 2 pooling windows (3x3, stride 2): with a = h + pt, window a>>1 sees it at row a&1 and, when a
        // is even, window (a>>1)-1 sees it at row 2.  All four candidates are fetched unconditionally from clamped addresses
        // (twelve independent loads in flight next to the x load) and masked afterwards -- the first version looped over the
        // candidates with data-dependent bounds, one dependent load group after the other.
        const long xo = (long)pix * C + c;
        const float4 xv = *reinterpret_cast<const float4*>(x + xo);
        const int a = h + pt, b = w_ + pl;
        const int ho1 = a >> 1, wo1 = b >> 1;
        const int hos[2] = {ho1, ho1 - 1}, khs[2] = {a & 1, 2}, wos[2] = {wo1, wo1 - 1}, kws[2] = {b & 1, 2};
        const bool vh[2] = {ho1 < Ho, (a & 1) == 0 && ho1 >= 1 && ho1 - 1 < Ho}, vw[2] = {wo1 < Wo, (b & 1) == 0 && wo1 >= 1 && wo1 - 1 < Wo};
        uchar4 id[4]; float4 ov[4], g[4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int hoc = min(max(hos[i], 0), Ho - 1), woc = min(max(wos[j], 0), Wo - 1);
                const long o = (((long)n * Ho + hoc) * Wo + woc) * C + c;
                id[i * 2 + j] = *reinterpret_cast<const uchar4*>(idx + o);
                ov[i * 2 + j] = *reinterpret_cast<const float4*>(out + o);
                g[i * 2 + j] = *reinterpret_cast<const float4*>(dpool + o);
            }
        float dz[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int q = i * 2 + j, slot = vh[i] && vw[j] ? khs[i] * 3 + kws[j] : 255;      // 255: never an argmax slot
                if (id[q].x == slot && ov[q].x > 0.f) dz[0] += g[q].x;
                if (id[q].y == slot && ov[q].y > 0.f) dz[1] += g[q].y;
                if (id[q].z == slot && ov[q].z > 0.f) dz[2] += g[q].z;
                if (id[q].w == slot && ov[q].w > 0.f) dz[3] += g[q].w;
            }
        float4 o4;
        o4.x = sc.x * (dz[0] - m1.x - (xv.x - mu.x) * rs.x * m2.x);
        o4.y = sc.y * (dz[1] - m1.y - (xv.y - mu.y) * rs.y * m2.y);
        o4.z = sc.z * (dz[2] - m1.z - (xv.z - mu.z) * rs.z * m2.z);
        o4.w = sc.w * (dz[3] - m1.w - (xv.w - mu.w) * rs.w * m2.w);
        *reinterpret_cast<float4*>(dx + xo) = o4;
    }
}

}  // namespace

extern "C" int sar_stem_conv7x7_fwd_f32(const float* x, const float* w, const float* bias, float* y, int B, int H,
                                        int W, int Ho, int Wo, int C0, int pad_top, int pad_left, hipStream_t stream) {
    if (!x || !w || !bias || !y || C0 > 64 || C0 % 4) return SAR_EINVAL;
    const int nblk = B * cdiv(Ho, ST_TH) * cdiv(Wo, ST_TW);
    hipLaunchKernelGGL(stem_fwd_kernel, dim3(nblk), dim3(256), 0, stream, x, w, bias, y, B, H, W, Ho, Wo, C0, pad_top, pad_left, nullptr, 1);
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}

extern "C" size_t sar_bn_workspace_bytes(int C);
extern "C" int sar_bn_finalize_f32(const float* part, int rows, const float* gamma, const float* beta, float* mean,
                                   float* rstd, float* scale, float* shift, float* moving_mean, float* moving_var, long R,
                                   int C, float eps, float momentum, int unbiased_moving_var, hipStream_t stream);

// The stem conv together with the training-mode statistics of the BatchNormalization behind it (resnet.py:173,43): y as
// sar_stem_conv7x7_fwd_f32, mean/rstd/scale/shift (+ moving averages) as sar_bn_stats_f32(y, ...) -- the sums come out of the
// conv epilogue (fp32 atomics into 512 partial rows of the workspace, cleared here first, combined in fp64).
extern "C" int sar_stem_conv7x7_fwd_bnstats_f32(const float* x, const float* w, const float* bias, float* y, int B, int H,
                                                int W, int Ho, int Wo, int C0, int pad_top, int pad_left, const float* gamma,
                                                const float* beta, double* sums, float* mean, float* rstd, float* scale,
                                                float* shift, float* moving_mean, float* moving_var, float eps,
                                                float momentum, int unbiased_moving_var, hipStream_t stream) {
    if (!x || !w || !bias || !y || C0 > 64 || C0 % 4 || !gamma || !beta || !sums || !mean || !rstd || !scale || !shift) return SAR_EINVAL;
    constexpr int ROWS = 512;
    if (sar_bn_workspace_bytes(C0) < sizeof(double) * 2 * C0 + sizeof(float) * (size_t)ROWS * 2 * C0) return SAR_EINVAL;
    float* part = reinterpret_cast<float*>(sums + 2 * C0);
    if (sar_zero_async(part, sizeof(float) * (size_t)ROWS * 2 * C0, stream) != hipSuccess) return SAR_ELAUNCH;
    const int nblk = B * cdiv(Ho, ST_TH) * cdiv(Wo, ST_TW);
    hipLaunchKernelGGL(stem_fwd_kernel, dim3(nblk), dim3(256), 0, stream, x, w, bias, y, B, H, W, Ho, Wo, C0, pad_top, pad_left, part, ROWS);
    SAR_CHECK_LAUNCH();
    return sar_bn_finalize_f32(part, ROWS, gamma, beta, mean, rstd, scale, shift, moving_mean, moving_var, (long)B * Ho * Wo, C0, eps,
                               momentum, unbiased_moving_var, stream);
}

extern "C" int sar_stem_conv7x7_wgrad_f32(const float* x, const float* dy, float* dw, float* db, int B, int H, int W,
                                          int Ho, int Wo, int C0, int pad_top, int pad_left, hipStream_t stream) {
    if (!x || !dy || !dw || C0 > 64 || C0 % 4) return SAR_EINVAL;
    const int ntiles = B * cdiv(Ho, ST_TH) * cdiv(Wo, ST_TW);
    const int nblk = ntiles < 1024 ? ntiles : 1024;
    hipLaunchKernelGGL(stem_wgrad_kernel<false>, dim3(nblk), dim3(256), 0, stream, x, dy, dw, db, B, H, W, Ho, Wo, C0, pad_top, pad_left, PoolBwd{});
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}

// Weight gradient of the stem conv straight from the gradient of the pooled map: dw += patches(x)^T * dx with
// dx = d/dy [maxpool(relu(bn(y)))] computed tile by tile inside the kernel (y = the stem output; dpool/out/idx = the pooling
// layer's incoming gradient, output and argmax slots; sums = the BatchNorm workspace AFTER sar_maxpool_bn_relu_bwd_f32(dx = NULL)
// filled its totals).  Replaces sar_maxpool_bn_relu_bwd_f32(dx) + sar_stem_conv7x7_wgrad_f32(dx) when nothing else needs dx
// (resnet.py:173-174: the stem is the first layer).
extern "C" int sar_stem_conv7x7_wgrad_poolbwd_f32(const float* x, const float* y, const float* dpool, const float* out,
                                                  const uint8_t* idx, const float* mean, const float* rstd, const float* scale,
                                                  const double* sums, float* dw, float* db, int B, int H, int W, int Ho, int Wo,
                                                  int C0, int pad_top, int pad_left, int Hp, int Wp, int pool_pad_top,
                                                  int pool_pad_left, hipStream_t stream) {
    if (!x || !y || !dpool || !out || !idx || !mean || !rstd || !scale || !sums || !dw || C0 > 64 || C0 % 4) return SAR_EINVAL;
    const int ntiles = B * cdiv(Ho, ST_TH) * cdiv(Wo, ST_TW);
    const int nblk = ntiles < 1024 ? ntiles : 1024;
    const PoolBwd pb{dpool, out, idx, y, mean, rstd, scale, sums, Hp, Wp, pool_pad_top, pool_pad_left};
    hipLaunchKernelGGL(stem_wgrad_kernel<true>, dim3(nblk), dim3(256), 0, stream, x, nullptr, dw, db, B, H, W, Ho, Wo, C0, pad_top, pad_left, pb);
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}

extern "C" int sar_bn_relu_maxpool_fwd_f32(const float* x, const float* scale, const float* shift, float* out,
                                           uint8_t* idx, int B, int H, int W, int C, int Ho, int Wo, int pad_top,
                                           int pad_left, hipStream_t stream) {
    if (!x || !scale || !shift || !out || !idx || C % 4) return SAR_EINVAL;
    const long total = (long)B * Ho * Wo * (C / 4);
    const int nblk = (int)(total / 256 + 1 < 4096 ? total / 256 + 1 : 4096);
    hipLaunchKernelGGL(bn_relu_maxpool_fwd_kernel, dim3(nblk), dim3(256), 0, stream, x, scale, shift, out, idx, B, H, W, C, Ho, Wo, pad_top, pad_left);
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}

constexpr int POOL_RED_BLOCKS = 1024;   // partial rows of the reduction (= the rows a BN workspace holds): 4 blocks per CU for a bandwidth-bound pass
// sums: the stem BN workspace (sar_bn_workspace_bytes(C)). dx may not alias x.  dx == NULL: only the two sums (the BatchNorm
// parameter gradients; sar_stem_conv7x7_wgrad_poolbwd_f32 consumes them and never materialises dx).
extern "C" int sar_maxpool_bn_relu_bwd_f32(const float* dpool, const float* out, const uint8_t* idx, const float* x,
                                           const float* mean, const float* rstd, const float* scale, const float* shift,
                                           double* sums, float* dx, int B, int H, int W, int C, int Ho, int Wo,
                                           int pad_top, int pad_left, hipStream_t stream) {
    if (!dpool || !out || !idx || !x || !mean || !rstd || !scale || !shift || !sums || C > 256 || (256 % C)) return SAR_EINVAL;
    float* part = reinterpret_cast<float*>(sums + 2 * C);          // workspace layout of sar_bn_workspace_bytes(C)
    if ((long)B * H * W * (W > H ? W : H) >= (1L << 32)) return SAR_EINVAL;         // magic-number division exactness bound
    const auto magic = [](int d) { return (unsigned)(((1UL << 32) + d - 1) / d); };
    hipLaunchKernelGGL(maxpool_bn_relu_bwd_reduce_kernel, dim3(POOL_RED_BLOCKS), dim3(256), 0, stream, dpool, out, idx, x, mean, rstd, scale, shift, part, B, H, W, C, Ho, Wo, pad_top, pad_left, magic(Wo), magic(Ho));
    hipLaunchKernelGGL(pool_combine_kernel, dim3(cdiv(2 * C, 4)), dim3(256), 0, stream, part, POOL_RED_BLOCKS, sums, 2 * C);
    if (C % 4) return SAR_EINVAL;
    if (!dx) { SAR_CHECK_LAUNCH(); return SAR_OK; }
    const long total = (long)B * H * W * (C / 4);
    const int nblk = (int)(total / 256 + 1 < 8192 ? total / 256 + 1 : 8192);
    hipLaunchKernelGGL(maxpool_bn_relu_bwd_apply_kernel, dim3(nblk), dim3(256), 0, stream, dpool, out, idx, x, mean, rstd, scale, sums, dx, B, H, W, C, Ho, Wo, pad_top, pad_left, magic(W), magic(H));
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}
