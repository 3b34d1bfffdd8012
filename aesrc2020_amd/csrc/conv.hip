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

struct ConvP {
    const float* x; const float* w; const float* bias; const float* scale; const float* shift; const float* res;
    float* y;
    int B, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pt, pl, splitk;
};

constexpr int CK = 16;  // k-tile

// ------------------------------------------------------------------ forward
template <int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN) void conv_fwd_kernel(ConvP p) {
    constexpr int BMt = 32 * WM, BNt = 32 * WN, NT = 64 * WM * WN;
    constexpr int NA = (BMt * 4 + NT - 1) / NT;   // float4 per thread for the A tile
    constexpr int NB = (BNt * 4 + NT - 1) / NT;
    __shared__ __attribute__((aligned(16))) float As[CK][BMt + 4];
    __shared__ __attribute__((aligned(16))) float Bs[CK][BNt + 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const long M = (long)p.B * p.Ho * p.Wo;
    const int K = p.KH * p.KW * p.Cin;
    const long m0 = (long)blockIdx.x * BMt;
    const int n0 = blockIdx.y * BNt;

    // per-thread A rows (fixed across the k loop)
    int an[NA], ah[NA], aw[NA];
    const int kq = (tid & 3) * 4;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int q = tid + i * NT;
        const long m = m0 + (q >> 2);
        if (q < BMt * 4 && m < M) {
            const int wo = (int)(m % p.Wo); const long t = m / p.Wo;
            const int ho = (int)(t % p.Ho); an[i] = (int)(t / p.Ho);
            ah[i] = ho * p.stride - p.pt; aw[i] = wo * p.stride - p.pl;
        } else {
            an[i] = -1; ah[i] = 0; aw[i] = 0;
        }
    }
    float4 ra[NA], rb[NB];
    auto load = [&](int kt) {
        const int k0 = kt * CK;
        const int tap = k0 / p.Cin, ci = k0 - tap * p.Cin + kq;
        const int kh = tap / p.KW, kw = tap - kh * p.KW;
        float4 sc = make_float4(1, 1, 1, 1), sh = make_float4(0, 0, 0, 0);
        if (p.scale) { sc = *reinterpret_cast<const float4*>(p.scale + ci); sh = *reinterpret_cast<const float4*>(p.shift + ci); }
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            float4 v = make_float4(0, 0, 0, 0);
            const int hi = ah[i] + kh, wi = aw[i] + kw;
            if (an[i] >= 0 && hi >= 0 && hi < p.H && wi >= 0 && wi < p.W) {
                v = *reinterpret_cast<const float4*>(p.x + (((long)an[i] * p.H + hi) * p.W + wi) * p.Cin + ci);
                if (p.scale) {
                    v.x = fmaxf(fmaf(v.x, sc.x, sh.x), 0.f); v.y = fmaxf(fmaf(v.y, sc.y, sh.y), 0.f);
                    v.z = fmaxf(fmaf(v.z, sc.z, sh.z), 0.f); v.w = fmaxf(fmaf(v.w, sc.w, sh.w), 0.f);
                }
            }
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int q = tid + i * NT;
            const int k = q / (BNt / 4), n4 = (q % (BNt / 4)) * 4;
            float4 v = make_float4(0, 0, 0, 0);
            if (q < BNt * 4 && n0 + n4 < p.Cout) v = *reinterpret_cast<const float4*>(p.w + (long)(k0 + k) * p.Cout + n0 + n4);
            rb[i] = v;
        }
    };
    auto store = [&]() {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int q = tid + i * NT;
            if (q < BMt * 4) {
                const int r = q >> 2;
                As[kq + 0][r] = ra[i].x; As[kq + 1][r] = ra[i].y; As[kq + 2][r] = ra[i].z; As[kq + 3][r] = ra[i].w;
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int q = tid + i * NT;
            if (q < BNt * 4) *reinterpret_cast<float4*>(&Bs[q / (BNt / 4)][(q % (BNt / 4)) * 4]) = rb[i];
        }
    };

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    // split-K: slice z owns k-tiles [kt0, kt1); small-M layers (stages 2-3) need it to fill 256 CUs
    const int nkt_all = K / CK, per = (nkt_all + p.splitk - 1) / p.splitk;
    const int kt0 = blockIdx.z * per, nkt = min(nkt_all, kt0 + per);
    if (kt0 < nkt) load(kt0);
    for (int kt = kt0; kt < nkt; ++kt) {
        store();
        __syncthreads();
        if (kt + 1 < nkt) load(kt + 1);
#pragma unroll
        for (int k = 0; k < CK; k += 2) {
            const float a = As[k + (lane >> 5)][wm * 32 + (lane & 31)];
            const float b = Bs[k + (lane >> 5)][wn * 32 + (lane & 31)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    const int col = n0 + wn * 32 + (lane & 31);
    if (col >= p.Cout) return;
    const bool lead = blockIdx.z == 0;
    const float bv = (p.bias && lead) ? p.bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const long row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row >= M) continue;
        float v = acc[r] + bv;
        if (p.res && lead) v += p.res[row * p.Cout + col];
        if (p.splitk > 1) atomicAdd(p.y + row * p.Cout + col, v);       // y was zeroed by the launcher
        else p.y[row * p.Cout + col] = v;
    }
}

// ------------------------------------------------------------------ dgrad
struct DgradP {
    const float* dy; const float* w; float* dx;
    int B, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pt, pl, accumulate, splitk;
};

template <int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN) void conv_dgrad_kernel(DgradP p) {
    constexpr int BMt = 32 * WM, BNt = 32 * WN, NT = 64 * WM * WN;
    constexpr int NA = (BMt * 4 + NT - 1) / NT;
    constexpr int NB = (BNt * 4 + NT - 1) / NT;
    __shared__ __attribute__((aligned(16))) float As[CK][BMt + 4];
    __shared__ __attribute__((aligned(16))) float Bs[CK][BNt + 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const long M = (long)p.B * p.H * p.W;           // input pixels
    const int K = p.KH * p.KW * p.Cout;
    const long m0 = (long)blockIdx.x * BMt;
    const int n0 = blockIdx.y * BNt;                // ci tile
    int an[NA], ah[NA], aw[NA];
    const int kq = (tid & 3) * 4;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int q = tid + i * NT;
        const long m = m0 + (q >> 2);
        if (q < BMt * 4 && m < M) {
            const int w_ = (int)(m % p.W); const long t = m / p.W;
            an[i] = (int)(t / p.H); ah[i] = (int)(t % p.H) + p.pt; aw[i] = w_ + p.pl;
        } else { an[i] = -1; ah[i] = 0; aw[i] = 0; }
    }
    float4 ra[NA], rb[NB];
    auto load = [&](int kt) {
        const int k0 = kt * CK;
        const int tap = k0 / p.Cout, co = k0 - tap * p.Cout + kq;
        const int kh = tap / p.KW, kw = tap - kh * p.KW;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            float4 v = make_float4(0, 0, 0, 0);
            const int hn = ah[i] - kh, wn_ = aw[i] - kw;
            if (an[i] >= 0 && hn >= 0 && wn_ >= 0 && (hn % p.stride) == 0 && (wn_ % p.stride) == 0) {
                const int ho = hn / p.stride, wo = wn_ / p.stride;
                if (ho < p.Ho && wo < p.Wo)
                    v = *reinterpret_cast<const float4*>(p.dy + (((long)an[i] * p.Ho + ho) * p.Wo + wo) * p.Cout + co);
            }
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {   // B[k=co][n=ci] = w[tap][ci][co]  (k contiguous)
            const int q = tid + i * NT;
            const int n = q >> 2, kk = (q & 3) * 4;
            float4 v = make_float4(0, 0, 0, 0);
            if (q < BNt * 4 && n0 + n < p.Cin)
                v = *reinterpret_cast<const float4*>(p.w + ((long)tap * p.Cin + n0 + n) * p.Cout + (k0 - tap * p.Cout) + kk);
            rb[i] = v;
        }
    };
    auto store = [&]() {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int q = tid + i * NT;
            if (q < BMt * 4) {
                const int r = q >> 2;
                As[kq + 0][r] = ra[i].x; As[kq + 1][r] = ra[i].y; As[kq + 2][r] = ra[i].z; As[kq + 3][r] = ra[i].w;
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int q = tid + i * NT;
            if (q < BNt * 4) {
                const int n = q >> 2, kk = (q & 3) * 4;
                Bs[kk + 0][n] = rb[i].x; Bs[kk + 1][n] = rb[i].y; Bs[kk + 2][n] = rb[i].z; Bs[kk + 3][n] = rb[i].w;
            }
        }
    };
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const int nkt_all = K / CK, per = (nkt_all + p.splitk - 1) / p.splitk;
    const int kt0 = blockIdx.z * per, nkt = min(nkt_all, kt0 + per);
    if (kt0 < nkt) load(kt0);
    for (int kt = kt0; kt < nkt; ++kt) {
        store();
        __syncthreads();
        if (kt + 1 < nkt) load(kt + 1);
#pragma unroll
        for (int k = 0; k < CK; k += 2) {
            const float a = As[k + (lane >> 5)][wm * 32 + (lane & 31)];
            const float b = Bs[k + (lane >> 5)][wn * 32 + (lane & 31)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    const int col = n0 + wn * 32 + (lane & 31);
    if (col >= p.Cin) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const long row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row >= M) continue;
        float* d = p.dx + row * p.Cin + col;
        if (p.splitk > 1) atomicAdd(d, acc[r]);                         // dx zeroed by the launcher unless accumulating
        else *d = p.accumulate ? (*d + acc[r]) : acc[r];
    }
}

// ------------------------------------------------------------------ wgrad
struct WgradP {
    const float* x; const float* dy; const float* scale; const float* shift; float* dw;
    int B, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pt, pl, rows_per_slice;
};

template <int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN) void conv_wgrad_kernel(WgradP p) {
    constexpr int BMt = 32 * WM, BNt = 32 * WN, NT = 64 * WM * WN;
    constexpr int NA = (BMt * 4 + NT - 1) / NT;
    constexpr int NB = (BNt * 4 + NT - 1) / NT;
    __shared__ __attribute__((aligned(16))) float As[CK][BMt + 4];
    __shared__ __attribute__((aligned(16))) float Bs[CK][BNt + 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const long R = (long)p.B * p.Ho * p.Wo;
    const int Mp = p.KH * p.KW * p.Cin;
    const int m0 = blockIdx.x * BMt, n0 = blockIdx.y * BNt;
    const long r_begin = (long)blockIdx.z * p.rows_per_slice;
    const long r_end = min(R, r_begin + p.rows_per_slice);
    // per-thread A columns (m' = (tap, ci)) are fixed; rows move
    int a_kh[NA], a_kw[NA], a_ci[NA], a_r[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int q = tid + i * NT;
        const int mp = m0 + (q % (BMt / 4)) * 4;
        a_r[i] = q / (BMt / 4);
        if (q < BMt * 4 && mp < Mp) {
            const int tap = mp / p.Cin; a_ci[i] = mp - tap * p.Cin;
            a_kh[i] = tap / p.KW; a_kw[i] = tap - a_kh[i] * p.KW;
        } else { a_ci[i] = -1; a_kh[i] = 0; a_kw[i] = 0; }
    }
    float4 ra[NA], rb[NB];
    auto load = [&](long r0) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            float4 v = make_float4(0, 0, 0, 0);
            const long m = r0 + a_r[i];
            if (a_ci[i] >= 0 && m < r_end) {
                const int wo = (int)(m % p.Wo); const long t = m / p.Wo;
                const int ho = (int)(t % p.Ho); const int n = (int)(t / p.Ho);
                const int hi = ho * p.stride - p.pt + a_kh[i], wi = wo * p.stride - p.pl + a_kw[i];
                if (hi >= 0 && hi < p.H && wi >= 0 && wi < p.W) {
                    v = *reinterpret_cast<const float4*>(p.x + (((long)n * p.H + hi) * p.W + wi) * p.Cin + a_ci[i]);
                    if (p.scale) {
                        const float4 sc = *reinterpret_cast<const float4*>(p.scale + a_ci[i]);
                        const float4 sh = *reinterpret_cast<const float4*>(p.shift + a_ci[i]);
                        v.x = fmaxf(fmaf(v.x, sc.x, sh.x), 0.f); v.y = fmaxf(fmaf(v.y, sc.y, sh.y), 0.f);
                        v.z = fmaxf(fmaf(v.z, sc.z, sh.z), 0.f); v.w = fmaxf(fmaf(v.w, sc.w, sh.w), 0.f);
                    }
                }
            }
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int q = tid + i * NT;
            const int r = q / (BNt / 4), n4 = (q % (BNt / 4)) * 4;
            float4 v = make_float4(0, 0, 0, 0);
            if (q < BNt * 4 && r0 + r < r_end && n0 + n4 < p.Cout)
                v = *reinterpret_cast<const float4*>(p.dy + (r0 + r) * p.Cout + n0 + n4);
            rb[i] = v;
        }
    };
    auto store = [&]() {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int q = tid + i * NT;
            if (q < BMt * 4) *reinterpret_cast<float4*>(&As[q / (BMt / 4)][(q % (BMt / 4)) * 4]) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int q = tid + i * NT;
            if (q < BNt * 4) *reinterpret_cast<float4*>(&Bs[q / (BNt / 4)][(q % (BNt / 4)) * 4]) = rb[i];
        }
    };
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    if (r_begin < r_end) load(r_begin);
    for (long r0 = r_begin; r0 < r_end; r0 += CK) {
        store();
        __syncthreads();
        if (r0 + CK < r_end) load(r0 + CK);
#pragma unroll
        for (int k = 0; k < CK; k += 2) {
            const float a = As[k + (lane >> 5)][wm * 32 + (lane & 31)];
            const float b = Bs[k + (lane >> 5)][wn * 32 + (lane & 31)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    const int col = n0 + wn * 32 + (lane & 31);
    if (col >= p.Cout) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < Mp) atomicAdd(p.dw + (long)row * p.Cout + col, acc[r]);
    }
}

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

static bool conv_args_ok(int Cin, int Cout) { return (Cin % 16 == 0) && (Cout % 16 == 0); }

// Blocks are 256 threads with ~9-11 KB of LDS: aim for >= 4 blocks per CU (1024 blocks) so that global-load latency of
// one block's k-tile hides behind the MFMA phase of another; each slice keeps >= 4 k-tiles.
static int pick_conv_splitk(int tiles, int nkt) {
    static const bool off = getenv("SAR_NO_SPLITK") != nullptr;      // debugging aid
    if (off || tiles >= 768) return 1;
    int s = (1024 + tiles - 1) / tiles;
    if (s > nkt / 4) s = nkt / 4;
    if (s > 16) s = 16;
    return s < 1 ? 1 : s;
}

extern "C" int sar_conv2d_fwd_f32(const float* x, const float* w, const float* bias, const float* scale,
                                  const float* shift, const float* residual, float* y, int B, int H, int W, int Cin,
                                  int Ho, int Wo, int Cout, int KH, int KW, int stride, int pad_top, int pad_left,
                                  hipStream_t stream) {
    if (!x || !w || !y || !conv_args_ok(Cin, Cout) || (scale && !shift)) return SAR_EINVAL;
    ConvP p{x, w, bias, scale, shift, residual, y, B, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad_top, pad_left, 1};
    const long M = (long)B * Ho * Wo;
    const int tiles = Cout <= 32 ? cdiv(M, 128) * cdiv(Cout, 32) : cdiv(M, 64) * cdiv(Cout, 64);
    p.splitk = pick_conv_splitk(tiles, KH * KW * Cin / CK);
    if (p.splitk > 1 && sar_zero_async(y, sizeof(float) * M * Cout, stream) != hipSuccess) return SAR_ELAUNCH;
    if (Cout <= 32) {
        dim3 grid(cdiv(M, 128), cdiv(Cout, 32), p.splitk);
        hipLaunchKernelGGL((conv_fwd_kernel<4, 1>), grid, dim3(256), 0, stream, p);
    } else {
        dim3 grid(cdiv(M, 64), cdiv(Cout, 64), p.splitk);
        hipLaunchKernelGGL((conv_fwd_kernel<2, 2>), grid, dim3(256), 0, stream, p);
    }
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}

extern "C" int sar_conv2d_dgrad_f32(const float* dy, const float* w, float* dx, int B, int H, int W, int Cin, int Ho,
                                    int Wo, int Cout, int KH, int KW, int stride, int pad_top, int pad_left,
                                    int accumulate, hipStream_t stream) {
    if (!dy || !w || !dx || !conv_args_ok(Cin, Cout)) return SAR_EINVAL;
    DgradP p{dy, w, dx, B, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad_top, pad_left, accumulate, 1};
    const long M = (long)B * H * W;
    const int tiles = Cin <= 32 ? cdiv(M, 128) * cdiv(Cin, 32) : cdiv(M, 64) * cdiv(Cin, 64);
    p.splitk = pick_conv_splitk(tiles, KH * KW * Cout / CK);
    if (p.splitk > 1 && !accumulate && sar_zero_async(dx, sizeof(float) * M * Cin, stream) != hipSuccess) return SAR_ELAUNCH;
    if (Cin <= 32) {
        dim3 grid(cdiv(M, 128), cdiv(Cin, 32), p.splitk);
        hipLaunchKernelGGL((conv_dgrad_kernel<4, 1>), grid, dim3(256), 0, stream, p);
    } else {
        dim3 grid(cdiv(M, 64), cdiv(Cin, 64), p.splitk);
        hipLaunchKernelGGL((conv_dgrad_kernel<2, 2>), grid, dim3(256), 0, stream, p);
    }
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}

extern "C" int sar_conv2d_wgrad_f32(const float* x, const float* dy, const float* scale, const float* shift,
                                    float* dw, int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW,
                                    int stride, int pad_top, int pad_left, hipStream_t stream) {
    if (!x || !dy || !dw || !conv_args_ok(Cin, Cout) || (scale && !shift)) return SAR_EINVAL;
    const long R = (long)B * Ho * Wo;
    const int Mp = KH * KW * Cin;
    // split the reduction so the grid fills 256 CUs a few times over
    const int tiles = cdiv(Mp, 64) * cdiv(Cout, Cout <= 32 ? 32 : 64);
    int splits = (int)((1024 + tiles - 1) / tiles);
    int rows = (int)((R + splits - 1) / splits);
    rows = ((rows + CK - 1) / CK) * CK;
    if (rows < 256) rows = 256;
    splits = cdiv(R, rows);
    WgradP p{x, dy, scale, shift, dw, B, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad_top, pad_left, rows};
    if (Cout <= 32) {
        dim3 grid(cdiv(Mp, 64), cdiv(Cout, 32), splits);
        hipLaunchKernelGGL((conv_wgrad_kernel<2, 1>), grid, dim3(128), 0, stream, p);
    } else {
        dim3 grid(cdiv(Mp, 64), cdiv(Cout, 64), splits);
        hipLaunchKernelGGL((conv_wgrad_kernel<2, 2>), grid, dim3(256), 0, stream, p);
    }
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}

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
