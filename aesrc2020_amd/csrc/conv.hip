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

__device__ __forceinline__ void stem_load_patch(float (*patch)[ST_PW + 1], const float* x, int n, int H, int W,
                                                int hi0, int wi0) {
    for (int i = threadIdx.x; i < ST_PH * ST_PW; i += blockDim.x) {
        const int r = i / ST_PW, c = i - r * ST_PW;
        const int hi = hi0 + r, wi = wi0 + c;
        patch[r][c] = (hi >= 0 && hi < H && wi >= 0 && wi < W) ? x[((long)n * H + hi) * W + wi] : 0.f;
    }
}

// y[n,ho,wo,c] = sum_{kh,kw} x[n,2ho-pt+kh,2wo-pl+kw] * w[kh,kw,0,c] + b[c];  C0 <= 64, C0 % 4 == 0
__global__ __launch_bounds__(256) void stem_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ bias, float* __restrict__ y, int B,
                                                       int H, int W, int Ho, int Wo, int C0, int pt, int pl) {
    __shared__ float patch[ST_PH][ST_PW + 1];
    __shared__ __attribute__((aligned(16))) float ws[49 * 64];
    const int tiles_w = (Wo + ST_TW - 1) / ST_TW, tiles_h = (Ho + ST_TH - 1) / ST_TH;
    int b = blockIdx.x;
    const int tw = b % tiles_w; b /= tiles_w;
    const int th = b % tiles_h; const int n = b / tiles_h;
    for (int i = threadIdx.x; i < 49 * C0; i += 256) ws[i] = w[i];
    stem_load_patch(patch, x, n, H, W, th * ST_TH * 2 - pt, tw * ST_TW * 2 - pl);
    __syncthreads();
    const int pix = threadIdx.x & 63, cg = threadIdx.x >> 6;      // 4 channel groups, wave-uniform
    const int cpg = C0 / 4;                                       // channels per group (<= 16)
    const int pr = pix / ST_TW, pc = pix % ST_TW;
    float acc[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) acc[c] = 0.f;
    for (int kh = 0; kh < 7; ++kh)
#pragma unroll
        for (int kw = 0; kw < 7; ++kw) {
            const float xv = patch[pr * 2 + kh][pc * 2 + kw];
            const float* wp = ws + (kh * 7 + kw) * C0 + cg * cpg;
#pragma unroll
            for (int c = 0; c < 16; ++c)
                if (c < cpg) acc[c] = fmaf(xv, wp[c], acc[c]);
        }
    const int ho = th * ST_TH + pr, wo = tw * ST_TW + pc;
    if (ho < Ho && wo < Wo) {
        float* dst = y + (((long)n * Ho + ho) * Wo + wo) * C0 + cg * cpg;
#pragma unroll
        for (int c = 0; c < 16; ++c)
            if (c < cpg) dst[c] = acc[c] + bias[cg * cpg + c];
    }
}

// dw[tap][c] += sum_pixels patch * dy ; db[c] += sum dy.  Persistent over tiles; one atomic flush per block.
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                         float* __restrict__ dw, float* __restrict__ db, int B, int H,
                                                         int W, int Ho, int Wo, int C0, int pt, int pl) {
    __shared__ float patch[ST_PH][ST_PW + 1];
    __shared__ float dys[ST_TH * ST_TW][64 + 1];
    const int tiles_w = (Wo + ST_TW - 1) / ST_TW, tiles_h = (Ho + ST_TH - 1) / ST_TH;
    const int ntiles = B * tiles_h * tiles_w;
    const int nout = 49 * C0;
    constexpr int MAXJ = 13;
    float acc[MAXJ];
    int o_tap[MAXJ], o_c[MAXJ];
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
        acc[j] = 0.f;
        const int idx = threadIdx.x + j * 256;
        o_tap[j] = idx < nout ? idx / C0 : -1; o_c[j] = idx < nout ? idx % C0 : 0;
    }
    float accb = 0.f;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int b = tile;
        const int tw = b % tiles_w; b /= tiles_w;
        const int th = b % tiles_h; const int n = b / tiles_h;
        __syncthreads();
        stem_load_patch(patch, x, n, H, W, th * ST_TH * 2 - pt, tw * ST_TW * 2 - pl);
        for (int i = threadIdx.x; i < ST_TH * ST_TW * C0; i += 256) {
            const int pix = i / C0, c = i - pix * C0;
            const int ho = th * ST_TH + pix / ST_TW, wo = tw * ST_TW + pix % ST_TW;
            dys[pix][c] = (ho < Ho && wo < Wo) ? dy[(((long)n * Ho + ho) * Wo + wo) * C0 + c] : 0.f;
        }
        __syncthreads();
        for (int pix = 0; pix < ST_TH * ST_TW; ++pix) {
            const int pr = pix / ST_TW, pc = pix % ST_TW;
#pragma unroll
            for (int j = 0; j < MAXJ; ++j)
                if (o_tap[j] >= 0) {
                    const int kh = o_tap[j] / 7, kw = o_tap[j] - kh * 7;
                    acc[j] = fmaf(patch[pr * 2 + kh][pc * 2 + kw], dys[pix][o_c[j]], acc[j]);
                }
            if (threadIdx.x < C0) accb += dys[pix][threadIdx.x];
        }
    }
#pragma unroll
    for (int j = 0; j < MAXJ; ++j)
        if (o_tap[j] >= 0) atomicAdd(dw + o_tap[j] * C0 + o_c[j], acc[j]);
    if (threadIdx.x < C0 && db) atomicAdd(db + threadIdx.x, accb);
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
__global__ __launch_bounds__(256) void maxpool_bn_relu_bwd_reduce_kernel(
    const float* __restrict__ dpool, const float* __restrict__ out, const uint8_t* __restrict__ idx,
    const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
    float* __restrict__ part, int B, int H, int W, int C, int Ho, int Wo, int pt, int pl) {
    // thread -> channel c = tid % C (C <= 256 and 256 % C == 0 assumed by the host), pixel group = tid / C
    const int c = threadIdx.x % C, pg = threadIdx.x / C, npg = blockDim.x / C;
    const long npix = (long)B * Ho * Wo;
    float s1 = 0.f, s2 = 0.f;
    const float mu = mean[c], rs = rstd[c];
    for (long pix = (long)blockIdx.x * npg + pg; pix < npix; pix += (long)gridDim.x * npg) {
        const long o = pix * C + c;
        const float a = out[o];
        if (a > 0.f) {
            const float g = dpool[o];
            const int k = idx[o];
            const int wo = (int)(pix % Wo); const long t = pix / Wo;
            const int ho = (int)(t % Ho); const int n = (int)(t / Ho);
            const int hi = ho * 2 - pt + k / 3, wi = wo * 2 - pl + k % 3;
            const float xh = (x[(((long)n * H + hi) * W + wi) * C + c] - mu) * rs;
            s1 += g; s2 += g * xh;
        }
    }
    __shared__ float r1[256], r2[256];
    r1[threadIdx.x] = s1; r2[threadIdx.x] = s2;
    __syncthreads();
    if (threadIdx.x < C) {
        float a1 = 0.f, a2 = 0.f;
        for (int g = 0; g < npg; ++g) { a1 += r1[g * C + threadIdx.x]; a2 += r2[g * C + threadIdx.x]; }
        part[(long)blockIdx.x * 2 * C + c] = a1; part[(long)blockIdx.x * 2 * C + C + c] = a2;
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

// Backward pass 2: dx = scale * (dz - s1/R - xhat*s2/R) at full resolution.
__global__ void maxpool_bn_relu_bwd_apply_kernel(const float* __restrict__ dpool, const float* __restrict__ out,
                                                 const uint8_t* __restrict__ idx, const float* __restrict__ x,
                                                 const float* __restrict__ mean, const float* __restrict__ rstd,
                                                 const float* __restrict__ scale, const double* __restrict__ sums,
                                                 float* __restrict__ dx, int B, int H, int W, int C, int Ho, int Wo,
                                                 int pt, int pl) {
    const long total = (long)B * H * W * C;
    const double invR = 1.0 / ((double)B * H * W);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C); long t = i / C;
        const int w_ = (int)(t % W); t /= W;
        const int h = (int)(t % H); const int n = (int)(t / H);
        float dz = 0.f;
        const int ho_lo = max(0, (h + pt - 1) / 2), ho_hi = min(Ho - 1, (h + pt) / 2);
        const int wo_lo = max(0, (w_ + pl - 1) / 2), wo_hi = min(Wo - 1, (w_ + pl) / 2);
        for (int ho = ho_lo; ho <= ho_hi; ++ho) {
            const int kh = h - (ho * 2 - pt);
            if (kh < 0 || kh > 2) continue;
            for (int wo = wo_lo; wo <= wo_hi; ++wo) {
                const int kw = w_ - (wo * 2 - pl);
                if (kw < 0 || kw > 2) continue;
                const long o = (((long)n * Ho + ho) * Wo + wo) * C + c;
                if (idx[o] == kh * 3 + kw && out[o] > 0.f) dz += dpool[o];
            }
        }
        const float xh = (x[i] - mean[c]) * rstd[c];
        dx[i] = scale[c] * (dz - (float)(sums[c] * invR) - xh * (float)(sums[C + c] * invR));
    }
}

}  // namespace

extern "C" int sar_stem_conv7x7_fwd_f32(const float* x, const float* w, const float* bias, float* y, int B, int H,
                                        int W, int Ho, int Wo, int C0, int pad_top, int pad_left, hipStream_t stream) {
    if (!x || !w || !bias || !y || C0 > 64 || C0 % 4) return SAR_EINVAL;
    const int nblk = B * cdiv(Ho, ST_TH) * cdiv(Wo, ST_TW);
    hipLaunchKernelGGL(stem_fwd_kernel, dim3(nblk), dim3(256), 0, stream, x, w, bias, y, B, H, W, Ho, Wo, C0, pad_top, pad_left);
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}

extern "C" int sar_stem_conv7x7_wgrad_f32(const float* x, const float* dy, float* dw, float* db, int B, int H, int W,
                                          int Ho, int Wo, int C0, int pad_top, int pad_left, hipStream_t stream) {
    if (!x || !dy || !dw || C0 > 64 || C0 % 4) return SAR_EINVAL;
    const int ntiles = B * cdiv(Ho, ST_TH) * cdiv(Wo, ST_TW);
    const int nblk = ntiles < 1024 ? ntiles : 1024;
    hipLaunchKernelGGL(stem_wgrad_kernel, dim3(nblk), dim3(256), 0, stream, x, dy, dw, db, B, H, W, Ho, Wo, C0, pad_top, pad_left);
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

// sums: the stem BN workspace (sar_bn_workspace_bytes(C)). dx may not alias x.
extern "C" int sar_maxpool_bn_relu_bwd_f32(const float* dpool, const float* out, const uint8_t* idx, const float* x,
                                           const float* mean, const float* rstd, const float* scale, double* sums,
                                           float* dx, int B, int H, int W, int C, int Ho, int Wo, int pad_top,
                                           int pad_left, hipStream_t stream) {
    if (!dpool || !out || !idx || !x || !mean || !rstd || !scale || !sums || !dx || C > 256 || (256 % C)) return SAR_EINVAL;
    float* part = reinterpret_cast<float*>(sums + 2 * C);          // workspace layout of sar_bn_workspace_bytes(C)
    hipLaunchKernelGGL(maxpool_bn_relu_bwd_reduce_kernel, dim3(256), dim3(256), 0, stream, dpool, out, idx, x, mean, rstd, part, B, H, W, C, Ho, Wo, pad_top, pad_left);
    hipLaunchKernelGGL(pool_combine_kernel, dim3(cdiv(2 * C, 4)), dim3(256), 0, stream, part, 256, sums, 2 * C);
    const long total = (long)B * H * W * C;
    const int nblk = (int)(total / 256 + 1 < 8192 ? total / 256 + 1 : 8192);
    hipLaunchKernelGGL(maxpool_bn_relu_bwd_apply_kernel, dim3(nblk), dim3(256), 0, stream, dpool, out, idx, x, mean, rstd, scale, sums, dx, B, H, W, C, Ho, Wo, pad_top, pad_left);
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}
