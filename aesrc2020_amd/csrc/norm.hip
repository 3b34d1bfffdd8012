// Normalisation kernels for SAR-Net on gfx950 (all HBM-bound: float4 loads, fp64 cross-block sums).
//  * training-mode BatchNormalization over (R, C) views, eps = 1e-3 (resnet.py:25; model.py:29-30; App. A):
//      stats -> finalize (mean, rstd, scale=gamma*rstd, shift=beta-mean*scale; moving averages)
//      the normalise+ReLU itself is folded into the consuming conv (conv.hip) or done by bn_apply
//      backward = reduce (sum dz, sum dz*xhat) + apply; dz = dact * (x*scale+shift > 0) when relu is set
//  * BatchNormalization on 2-D (B, C) inputs (model.py:286,289): thread-per-channel
//  * LayerNormalization (keras_layer_normalization, eps = 1e-14, model.py:32-33): wave-per-row
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int BN_MAX_BLOCKS = 256;    // grid cap of the reduction kernels in this file
constexpr int BN_PART_ROWS = 1024;    // partial rows the workspace holds (conv epilogues write one row per row tile)

// ---- per-channel sum / sum-of-squares of x (R, C), C % 4 == 0, C <= 1024 -----------------------------
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ x, float* __restrict__ part, long R,
                                                       int C, int tpr /* pow2 >= C/4, <= 256 */) {
    __shared__ float4 s1[256], s2[256];
    const int cv = threadIdx.x % tpr, rg = threadIdx.x / tpr, nrg = 256 / tpr;
    float4 a = make_float4(0, 0, 0, 0), b = a;
    if (cv * 4 < C) {
        for (long r = (long)blockIdx.x * nrg + rg; r < R; r += (long)gridDim.x * nrg) {
            const float4 v = *reinterpret_cast<const float4*>(x + r * C + cv * 4);
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
            b.x = fmaf(v.x, v.x, b.x); b.y = fmaf(v.y, v.y, b.y); b.z = fmaf(v.z, v.z, b.z); b.w = fmaf(v.w, v.w, b.w);
        }
    }
    s1[threadIdx.x] = a; s2[threadIdx.x] = b;
    __syncthreads();
    if (threadIdx.x < tpr && threadIdx.x * 4 < C) {      // per-block partials, combined in fp64 by the next kernel
        float4 d1 = make_float4(0, 0, 0, 0), d2 = d1;
        for (int g = 0; g < nrg; ++g) {
            const float4 u = s1[g * tpr + threadIdx.x], w = s2[g * tpr + threadIdx.x];
            d1.x += u.x; d1.y += u.y; d1.z += u.z; d1.w += u.w;
            d2.x += w.x; d2.y += w.y; d2.z += w.z; d2.w += w.w;
        }
        float* o = part + (long)blockIdx.x * 2 * C;
        *reinterpret_cast<float4*>(o + threadIdx.x * 4) = d1;
        *reinterpret_cast<float4*>(o + C + threadIdx.x * 4) = d2;
    }
}

// One wave per column: lanes stride over the per-block partials (independent loads in flight), fp64 wave reduction.
__device__ __forceinline__ double combine_column(const float* __restrict__ part, int nb, int stride, int c, int lane) {
    double s = 0;
    for (int b = lane; b < nb; b += 64) s += (double)part[(long)b * stride + c];
    return wave_sum_d(s);
}

// totals[c] = sum over blocks of part[b][c], c < C2 (fp64).  grid = ceil(C2/4), block = 256 (4 waves = 4 columns)
__global__ __launch_bounds__(256) void bn_combine_kernel(const float* __restrict__ part, int nb, double* __restrict__ totals, int C2) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= C2) return;
    const double s = combine_column(part, nb, C2, c, lane);
    if (lane == 0) totals[c] = s;
}

// combines the per-block partials in fp64 and folds gamma/beta into scale/shift (one launch)
// Block = FIN_CH channels x FIN_RG row groups: each thread walks the partial rows of its group reading FIN_CH consecutive
// channels (32-byte segments; the first version's wave-per-channel walk read one float per cache line), four rows per trip so
// that eight loads are in flight per thread (the kernel is a chain of L2 round trips: with 16 groups and two rows per trip
// it took 7 us on <= 1 MB of partials), fp64 accumulation, LDS reduce.
constexpr int FIN_CH = 8, FIN_RG = 32;
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ part, int nb, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float* __restrict__ mean, float* __restrict__ rstd,
                                   float* __restrict__ scale, float* __restrict__ shift, float* __restrict__ mov_mean,
                                   float* __restrict__ mov_var, long R, int C, float eps, float momentum,
                                   int unbiased_moving_var) {
    __shared__ double r1[FIN_RG][FIN_CH + 1], r2[FIN_RG][FIN_CH + 1];
    const int cl = threadIdx.x % FIN_CH, rg = threadIdx.x / FIN_CH;
    const int c = blockIdx.x * FIN_CH + cl;
    double a1 = 0, a2 = 0;
    if (c < C) {
        const float* p = part + c;
        int row = rg;
        for (; row + 3 * FIN_RG < nb; row += 4 * FIN_RG) {
            float u[4], v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { u[j] = p[(long)(row + j * FIN_RG) * 2 * C]; v[j] = p[(long)(row + j * FIN_RG) * 2 * C + C]; }
            a1 += ((double)u[0] + (double)u[1]) + ((double)u[2] + (double)u[3]);
            a2 += ((double)v[0] + (double)v[1]) + ((double)v[2] + (double)v[3]);
        }
        for (; row < nb; row += FIN_RG) { a1 += (double)p[(long)row * 2 * C]; a2 += (double)p[(long)row * 2 * C + C]; }
    }
    r1[rg][cl] = a1; r2[rg][cl] = a2;
    __syncthreads();
    if (rg != 0 || c >= C) return;
    double s1 = 0, s2 = 0;
#pragma unroll
    for (int g = 0; g < FIN_RG; ++g) { s1 += r1[g][cl]; s2 += r2[g][cl]; }
    const double mu = s1 / (double)R;
    double var = s2 / (double)R - mu * mu;
    if (var < 0) var = 0;
    const double rs = 1.0 / sqrt(var + (double)eps);
    mean[c] = (float)mu; rstd[c] = (float)rs;
    const float sc = gamma[c] * (float)rs;
    scale[c] = sc; shift[c] = beta[c] - (float)mu * sc;
    if (mov_mean) {
        const double v = unbiased_moving_var && R > 1 ? var * (double)R / (double)(R - 1) : var;
        mov_mean[c] = momentum * mov_mean[c] + (1.f - momentum) * (float)mu;
        mov_var[c] = momentum * mov_var[c] + (1.f - momentum) * (float)v;
    }
}

__global__ void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                const float* __restrict__ shift, float* __restrict__ y, long n4, int C, int relu) {
    const int C4 = C / 4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4) * 4;
        float4 v = reinterpret_cast<const float4*>(x)[i];
        const float4 sc = *reinterpret_cast<const float4*>(scale + c), sh = *reinterpret_cast<const float4*>(shift + c);
        v.x = fmaf(v.x, sc.x, sh.x); v.y = fmaf(v.y, sc.y, sh.y); v.z = fmaf(v.z, sc.z, sh.z); v.w = fmaf(v.w, sc.w, sh.w);
        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        reinterpret_cast<float4*>(y)[i] = v;
    }
}

// sums[0:C] = sum dz ; sums[C:2C] = sum dz*xhat
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ dact, const float* __restrict__ x,
                                                            const float* __restrict__ scale, const float* __restrict__ shift,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            float* __restrict__ part, long R, int C, int tpr, int relu) {
    __shared__ float4 s1[256], s2[256];
    const int cv = threadIdx.x % tpr, rg = threadIdx.x / tpr, nrg = 256 / tpr;
    float4 a = make_float4(0, 0, 0, 0), b = a;
    if (cv * 4 < C) {
        const float4 sc = *reinterpret_cast<const float4*>(scale + cv * 4), sh = *reinterpret_cast<const float4*>(shift + cv * 4);
        const float4 mu = *reinterpret_cast<const float4*>(mean + cv * 4), rs = *reinterpret_cast<const float4*>(rstd + cv * 4);
        auto step = [&](const float4 v, float4 g) {
            if (relu) {
                if (fmaf(v.x, sc.x, sh.x) <= 0.f) g.x = 0.f;
                if (fmaf(v.y, sc.y, sh.y) <= 0.f) g.y = 0.f;
                if (fmaf(v.z, sc.z, sh.z) <= 0.f) g.z = 0.f;
                if (fmaf(v.w, sc.w, sh.w) <= 0.f) g.w = 0.f;
            }
            a.x += g.x; a.y += g.y; a.z += g.z; a.w += g.w;
            b.x = fmaf(g.x, (v.x - mu.x) * rs.x, b.x); b.y = fmaf(g.y, (v.y - mu.y) * rs.y, b.y);
            b.z = fmaf(g.z, (v.z - mu.z) * rs.z, b.z); b.w = fmaf(g.w, (v.w - mu.w) * rs.w, b.w);
        };
        // four rows per trip, all eight loads issued before the first use: the kernel is a handful of memory round trips per
        // thread (4-16 rows), so loads in flight are what counts
        const long S = (long)gridDim.x * nrg;
        long r = (long)blockIdx.x * nrg + rg;
        const float* xp = x + cv * 4; const float* gp = dact + cv * 4;
        for (; r + 3 * S < R; r += 4 * S) {
            float4 v[4], g[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[j] = *reinterpret_cast<const float4*>(xp + (r + j * S) * C); g[j] = *reinterpret_cast<const float4*>(gp + (r + j * S) * C); }
#pragma unroll
            for (int j = 0; j < 4; ++j) step(v[j], g[j]);
        }
        for (; r < R; r += S) step(*reinterpret_cast<const float4*>(xp + r * C), *reinterpret_cast<const float4*>(gp + r * C));
    }
    s1[threadIdx.x] = a; s2[threadIdx.x] = b;
    __syncthreads();
    if (threadIdx.x < tpr && threadIdx.x * 4 < C) {      // per-block partials, combined in fp64 by the next kernel
        float4 d1 = make_float4(0, 0, 0, 0), d2 = d1;
        for (int g = 0; g < nrg; ++g) {
            const float4 u = s1[g * tpr + threadIdx.x], w = s2[g * tpr + threadIdx.x];
            d1.x += u.x; d1.y += u.y; d1.z += u.z; d1.w += u.w;
            d2.x += w.x; d2.y += w.y; d2.z += w.z; d2.w += w.w;
        }
        float* o = part + (long)blockIdx.x * 2 * C;
        *reinterpret_cast<float4*>(o + threadIdx.x * 4) = d1;
        *reinterpret_cast<float4*>(o + C + threadIdx.x * 4) = d2;
    }
}

// dx = scale*(dz - s1/R - xhat*s2/R) [+ add];  block 0 also accumulates dgamma += s2, dbeta += s1
__global__ void bn_bwd_apply_kernel(const float* __restrict__ dact, const float* __restrict__ x,
                                    const float* __restrict__ scale, const float* __restrict__ shift,
                                    const float* __restrict__ mean, const float* __restrict__ rstd,
                                    const double* __restrict__ sums, const float* __restrict__ add,
                                    float* __restrict__ dx, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                    long R, int C, int relu) {
    const int C4 = C / 4;
    const long n4 = R * C4;
    const double invR = 1.0 / (double)R;
    if (blockIdx.x == 0 && dgamma) {
        for (int c = threadIdx.x; c < C; c += blockDim.x) { dgamma[c] += (float)sums[C + c]; dbeta[c] += (float)sums[c]; }
    }
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4) * 4;
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        float4 g = reinterpret_cast<const float4*>(dact)[i];
        const float4 sc = *reinterpret_cast<const float4*>(scale + c), sh = *reinterpret_cast<const float4*>(shift + c);
        const float4 mu = *reinterpret_cast<const float4*>(mean + c), rs = *reinterpret_cast<const float4*>(rstd + c);
        if (relu) {
            if (fmaf(v.x, sc.x, sh.x) <= 0.f) g.x = 0.f;
            if (fmaf(v.y, sc.y, sh.y) <= 0.f) g.y = 0.f;
            if (fmaf(v.z, sc.z, sh.z) <= 0.f) g.z = 0.f;
            if (fmaf(v.w, sc.w, sh.w) <= 0.f) g.w = 0.f;
        }
        float4 o;
        o.x = sc.x * (g.x - (float)(sums[c + 0] * invR) - (v.x - mu.x) * rs.x * (float)(sums[C + c + 0] * invR));
        o.y = sc.y * (g.y - (float)(sums[c + 1] * invR) - (v.y - mu.y) * rs.y * (float)(sums[C + c + 1] * invR));
        o.z = sc.z * (g.z - (float)(sums[c + 2] * invR) - (v.z - mu.z) * rs.z * (float)(sums[C + c + 2] * invR));
        o.w = sc.w * (g.w - (float)(sums[c + 3] * invR) - (v.w - mu.w) * rs.w * (float)(sums[C + c + 3] * invR));
        if (add) { const float4 e = reinterpret_cast<const float4*>(add)[i]; o.x += e.x; o.y += e.y; o.z += e.z; o.w += e.w; }
        reinterpret_cast<float4*>(dx)[i] = o;
    }
}

__global__ void bn_param_grads_kernel(const double* __restrict__ sums, float* __restrict__ dgamma,
                                      float* __restrict__ dbeta, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) { dgamma[c] += (float)sums[C + c]; dbeta[c] += (float)sums[c]; }
}

// ---- 2-D BatchNorm: x (B, C).  Block = 32 channels x 8 row groups: coalesced 128-byte rows, B/8 sequential loads per
//      thread (the tensors are tiny -- 64 x 8192 at most -- so the kernel is latency-bound: parallelism over rows matters)
constexpr int BN1_CH = 32, BN1_RG = 8;
__device__ __forceinline__ float bn1d_colsum(float v, float (*red)[BN1_CH], int rg, int cl) {
    __syncthreads();                                  // protects `red` against the previous use
    red[rg][cl] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < BN1_RG; ++g) t += red[g][cl];
    return t;
}
// (round 6: a thread's rows are loaded ONCE, all in flight, and kept in registers -- B <= BN1_RG * BN1_RM, the model's batches; the three passes over
// x (sum, squared deviations, normalise) were three dependent round trips of a kernel that is nothing but round trips.  Larger batches re-read.)
constexpr int BN1_RM = 16;
__global__ __launch_bounds__(256) void bn1d_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                const float* __restrict__ beta, float* __restrict__ y, float* __restrict__ mean,
                                float* __restrict__ rstd, float* __restrict__ mov_mean, float* __restrict__ mov_var,
                                int B, int C, float eps, float momentum) {
    __shared__ float red[BN1_RG][BN1_CH];
    const int cl = threadIdx.x % BN1_CH, rg = threadIdx.x / BN1_CH;
    const int c = blockIdx.x * BN1_CH + cl;
    const bool ok = c < C;
    const bool held = B <= BN1_RG * BN1_RM;                        // (kernel-uniform)
    float xv[BN1_RM];
    float s = 0.f;
    if (held) {
#pragma unroll
        for (int i = 0; i < BN1_RM; ++i) { const int b = rg + i * BN1_RG; xv[i] = (ok && b < B) ? x[(long)b * C + c] : 0.f; }
#pragma unroll
        for (int i = 0; i < BN1_RM; ++i) s += xv[i];
    } else if (ok) for (int b = rg; b < B; b += BN1_RG) s += x[(long)b * C + c];
    const float g = ok ? gamma[c] : 0.f, be = ok ? beta[c] : 0.f;  // (requested here: behind the two reductions they were one more round trip)
    const float mu = bn1d_colsum(s, red, rg, cl) / B;
    float q = 0.f;
    if (held) {
#pragma unroll
        for (int i = 0; i < BN1_RM; ++i) { const float d = (rg + i * BN1_RG < B) ? xv[i] - mu : 0.f; q = fmaf(d, d, q); }
    } else if (ok) for (int b = rg; b < B; b += BN1_RG) { const float d = x[(long)b * C + c] - mu; q = fmaf(d, d, q); }
    const float var = bn1d_colsum(q, red, rg, cl) / B;
    if (!ok) return;
    const float rs = 1.f / sqrtf(var + eps);
    if (held) {
#pragma unroll
        for (int i = 0; i < BN1_RM; ++i) { const int b = rg + i * BN1_RG; if (b < B) y[(long)b * C + c] = (xv[i] - mu) * rs * g + be; }
    } else for (int b = rg; b < B; b += BN1_RG) y[(long)b * C + c] = (x[(long)b * C + c] - mu) * rs * g + be;
    if (rg == 0) {
        mean[c] = mu; rstd[c] = rs;
        if (mov_mean) {
            mov_mean[c] = momentum * mov_mean[c] + (1.f - momentum) * mu;
            // Keras 2.2.x BatchNormalization.call feeds the moving average the sample variance
            // var * n / (n - (1 + epsilon)) ("unbiased estimator of population variance"), not the biased batch variance
            const float nb = (float)B, corr = nb > 1.f + eps ? nb / (nb - (1.f + eps)) : 1.f;
            mov_var[c] = momentum * mov_var[c] + (1.f - momentum) * var * corr;
        }
    }
}

__global__ __launch_bounds__(256) void bn1d_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                const float* __restrict__ gamma, const float* __restrict__ mean,
                                const float* __restrict__ rstd, float* __restrict__ dx, float* __restrict__ dgamma,
                                float* __restrict__ dbeta, int B, int C) {
    __shared__ float red[BN1_RG][BN1_CH];
    const int cl = threadIdx.x % BN1_CH, rg = threadIdx.x / BN1_CH;
    const int c = blockIdx.x * BN1_CH + cl;
    const bool ok = c < C;
    const bool held = B <= BN1_RG * BN1_RM;                        // (kernel-uniform)
    const float mu = ok ? mean[c] : 0.f, rs = ok ? rstd[c] : 0.f, gm = ok ? gamma[c] : 0.f;
    float gv[BN1_RM], xh[BN1_RM];
    float a1 = 0.f, a2 = 0.f;
    if (held) {
#pragma unroll
        for (int i = 0; i < BN1_RM; ++i) {
            const int b = rg + i * BN1_RG; const bool in = ok && b < B;
            gv[i] = in ? dy[(long)b * C + c] : 0.f; xh[i] = in ? x[(long)b * C + c] : mu;
        }
#pragma unroll
        for (int i = 0; i < BN1_RM; ++i) { xh[i] = (xh[i] - mu) * rs; a1 += gv[i]; a2 = fmaf(gv[i], xh[i], a2); }
    } else if (ok)
        for (int b = rg; b < B; b += BN1_RG) {
            const float g = dy[(long)b * C + c];
            a1 += g; a2 = fmaf(g, (x[(long)b * C + c] - mu) * rs, a2);
        }
    const float s1 = bn1d_colsum(a1, red, rg, cl), s2 = bn1d_colsum(a2, red, rg, cl);
    if (!ok) return;
    if (rg == 0) { dgamma[c] += s2; dbeta[c] += s1; }
    const float k = gm * rs, m1 = s1 / B, m2 = s2 / B;
    if (held) {
#pragma unroll
        for (int i = 0; i < BN1_RM; ++i) { const int b = rg + i * BN1_RG; if (b < B) dx[(long)b * C + c] = k * (gv[i] - m1 - xh[i] * m2); }
    } else
        for (int b = rg; b < B; b += BN1_RG) {
            const float xhh = (x[(long)b * C + c] - mu) * rs;
            dx[(long)b * C + c] = k * (dy[(long)b * C + c] - m1 - xhh * m2);
        }
}

// ---- LayerNorm: x (R, C), one wave per row, C % 4 == 0, C <= 2048 -------------------------------------
constexpr int LN_MAXV = 8;   // float4 per lane

__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float* __restrict__ y,
                                                     float* __restrict__ mean, float* __restrict__ rstd, long R, int C,
                                                     float eps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int C4 = C / 4;
    for (long r = (long)blockIdx.x * 4 + wave; r < R; r += (long)gridDim.x * 4) {
        float4 v[LN_MAXV];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < LN_MAXV; ++j) {
            const int c4 = lane + j * 64;
            if (c4 < C4) { v[j] = reinterpret_cast<const float4*>(x + r * C)[c4]; s += v[j].x + v[j].y + v[j].z + v[j].w; }
        }
        const float mu = wave_sum(s) / C;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < LN_MAXV; ++j) {
            const int c4 = lane + j * 64;
            if (c4 < C4) {
                const float a = v[j].x - mu, b = v[j].y - mu, c = v[j].z - mu, d = v[j].w - mu;
                q += a * a + b * b + c * c + d * d;
            }
        }
        const float rs = 1.f / sqrtf(wave_sum(q) / C + eps);
        if (lane == 0) { mean[r] = mu; rstd[r] = rs; }
#pragma unroll
        for (int j = 0; j < LN_MAXV; ++j) {
            const int c4 = lane + j * 64;
            if (c4 < C4) {
                const float4 g = reinterpret_cast<const float4*>(gamma)[c4], be = reinterpret_cast<const float4*>(beta)[c4];
                float4 o;
                o.x = (v[j].x - mu) * rs * g.x + be.x; o.y = (v[j].y - mu) * rs * g.y + be.y;
                o.z = (v[j].z - mu) * rs * g.z + be.z; o.w = (v[j].w - mu) * rs * g.w + be.w;
                reinterpret_cast<float4*>(y + r * C)[c4] = o;
            }
        }
    }
}

template <int NW>
__global__ __launch_bounds__(NW * 64) void ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, float* __restrict__ dx,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta, long R,
                                                     int C, int accumulate, int act) {
    extern __shared__ __attribute__((aligned(16))) float sm[];       // NW waves x 2 x C
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int C4 = C / 4;
    float4 ag[LN_MAXV], ab[LN_MAXV];
#pragma unroll
    for (int j = 0; j < LN_MAXV; ++j) { ag[j] = make_float4(0, 0, 0, 0); ab[j] = ag[j]; }
    for (long r = (long)blockIdx.x * NW + wave; r < R; r += (long)gridDim.x * NW) {
        const float mu = mean[r], rs = rstd[r];
        float4 xh[LN_MAXV], g[LN_MAXV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < LN_MAXV; ++j) {
            const int c4 = lane + j * 64;
            if (c4 < C4) {
                const float4 v = reinterpret_cast<const float4*>(x + r * C)[c4];
                const float4 d = reinterpret_cast<const float4*>(dy + r * C)[c4];
                const float4 gm = reinterpret_cast<const float4*>(gamma)[c4];
                xh[j] = make_float4((v.x - mu) * rs, (v.y - mu) * rs, (v.z - mu) * rs, (v.w - mu) * rs);
                g[j] = make_float4(d.x * gm.x, d.y * gm.y, d.z * gm.z, d.w * gm.w);
                s1 += g[j].x + g[j].y + g[j].z + g[j].w;
                s2 += g[j].x * xh[j].x + g[j].y * xh[j].y + g[j].z * xh[j].z + g[j].w * xh[j].w;
                ag[j].x = fmaf(d.x, xh[j].x, ag[j].x); ag[j].y = fmaf(d.y, xh[j].y, ag[j].y);
                ag[j].z = fmaf(d.z, xh[j].z, ag[j].z); ag[j].w = fmaf(d.w, xh[j].w, ag[j].w);
                ab[j].x += d.x; ab[j].y += d.y; ab[j].z += d.z; ab[j].w += d.w;
            }
        }
        const float m1 = wave_sum(s1) / C, m2 = wave_sum(s2) / C;
#pragma unroll
        for (int j = 0; j < LN_MAXV; ++j) {
            const int c4 = lane + j * 64;
            if (c4 < C4) {
                float4 o;
                o.x = rs * (g[j].x - m1 - xh[j].x * m2); o.y = rs * (g[j].y - m1 - xh[j].y * m2);
                o.z = rs * (g[j].z - m1 - xh[j].z * m2); o.w = rs * (g[j].w - m1 - xh[j].w * m2);
                if (act == ACT_TANH) {                              // x = tanh(z) of the Dense in front: hand back d/dz (kernel-uniform)
                    const float ir = 1.f / rs;
                    const float v0 = fmaf(xh[j].x, ir, mu), v1 = fmaf(xh[j].y, ir, mu), v2 = fmaf(xh[j].z, ir, mu), v3 = fmaf(xh[j].w, ir, mu);
                    o.x *= 1.f - v0 * v0; o.y *= 1.f - v1 * v1; o.z *= 1.f - v2 * v2; o.w *= 1.f - v3 * v3;
                }
                float4* d = reinterpret_cast<float4*>(dx + r * C) + c4;
                if (accumulate) { const float4 e = *d; o.x += e.x; o.y += e.y; o.z += e.z; o.w += e.w; }
                *d = o;
            }
        }
    }
    // cross-wave reduce of the parameter-gradient partials, then one atomic per column per block
    float* sg = sm + wave * 2 * C; float* sb = sg + C;
#pragma unroll
    for (int j = 0; j < LN_MAXV; ++j) {
        const int c4 = lane + j * 64;
        if (c4 < C4) { reinterpret_cast<float4*>(sg)[c4] = ag[j]; reinterpret_cast<float4*>(sb)[c4] = ab[j]; }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += NW * 64) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) { a += sm[w * 2 * C + c]; b += sm[w * 2 * C + C + c]; }
        atomicAdd(dgamma + c, a); atomicAdd(dbeta + c, b);
    }
}

}  // namespace

// BN reduction workspace: double totals[2C] followed by float partials[BN_PART_ROWS][2C]
extern "C" size_t sar_bn_workspace_bytes(int C) { return sizeof(double) * 2 * C + sizeof(float) * (size_t)BN_PART_ROWS * 2 * C; }

static int pow2_tpr(int C) { int v = C / 4, p = 1; while (p < v) p <<= 1; return p; }

// sums: workspace of sar_bn_workspace_bytes(C) (fp64 totals + per-block partials; no zeroing, no atomics).
// Computes batch statistics of x (R,C) and the folded scale/shift.
extern "C" int sar_bn_stats_f32(const float* x, const float* gamma, const float* beta, double* sums, float* mean,
                                float* rstd, float* scale, float* shift, float* moving_mean, float* moving_var, long R,
                                int C, float eps, float momentum, int unbiased_moving_var, hipStream_t stream) {
    if (!x || !gamma || !beta || !sums || !mean || !rstd || !scale || !shift || C % 4 || C > 1024 || R <= 0) return SAR_EINVAL;
    const int tpr = pow2_tpr(C), nrg = 256 / tpr;
    long nb = (R + nrg * 4 - 1) / (nrg * 4);
    if (nb > BN_MAX_BLOCKS) nb = BN_MAX_BLOCKS;
    if (nb < 1) nb = 1;
    float* part = reinterpret_cast<float*>(sums + 2 * C);
    hipLaunchKernelGGL(bn_stats_kernel, dim3((int)nb), dim3(256), 0, stream, x, part, R, C, tpr);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, FIN_CH)), dim3(256), 0, stream, part, (int)nb, gamma, beta, mean, rstd, scale,
                       shift, moving_mean, moving_var, R, C, eps, momentum, unbiased_moving_var);
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}

// Second half of sar_bn_stats_f32 for producers that wrote the per-block partials themselves (conv epilogues):
// part = [rows][2C] (sum | sum of squares); combines in fp64, folds gamma/beta, updates the moving averages.
extern "C" int sar_bn_finalize_f32(const float* part, int rows, const float* gamma, const float* beta, float* mean,
                                   float* rstd, float* scale, float* shift, float* moving_mean, float* moving_var, long R,
                                   int C, float eps, float momentum, int unbiased_moving_var, hipStream_t stream) {
    if (!part || rows < 1 || rows > BN_PART_ROWS || !gamma || !beta || !mean || !rstd || !scale || !shift || C < 1 || R <= 0) return SAR_EINVAL;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, FIN_CH)), dim3(256), 0, stream, part, rows, gamma, beta, mean, rstd, scale,
                       shift, moving_mean, moving_var, R, C, eps, momentum, unbiased_moving_var);
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}

// ---- inference mode (Keras learning_phase 0): moving averages instead of batch statistics
__global__ void bn_fold_kernel(const float* gamma, const float* beta, const float* mm, const float* mv, float* scale,
                               float* shift, int C, float eps) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float sc = gamma[c] * rsqrtf(mv[c] + eps);
    scale[c] = sc; shift[c] = beta[c] - mm[c] * sc;
}
__global__ void bn_infer_kernel(const float* x, const float* gamma, const float* beta, const float* mm, const float* mv,
                                float* y, long n, int C, float eps, int relu) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const float sc = gamma[c] * rsqrtf(mv[c] + eps);
        float v = fmaf(x[i], sc, beta[c] - mm[c] * sc);
        y[i] = relu ? fmaxf(v, 0.f) : v;
    }
}
extern "C" int sar_bn_fold_f32(const float* gamma, const float* beta, const float* moving_mean, const float* moving_var,
                               float* scale, float* shift, int C, float eps, hipStream_t stream) {
    if (!gamma || !beta || !moving_mean || !moving_var || !scale || !shift || C < 1) return SAR_EINVAL;
    hipLaunchKernelGGL(bn_fold_kernel, dim3(cdiv(C, 256)), dim3(256), 0, stream, gamma, beta, moving_mean, moving_var, scale, shift, C, eps);
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}
extern "C" int sar_bn_infer_f32(const float* x, const float* gamma, const float* beta, const float* moving_mean,
                                const float* moving_var, float* y, long R, int C, float eps, int relu, hipStream_t stream) {
    if (!x || !gamma || !beta || !moving_mean || !moving_var || !y || C < 1 || R < 0) return SAR_EINVAL;
    if (R == 0) return SAR_OK;
    const long n = R * C;
    const int nb = (int)(n / 256 + 1 < 4096 ? n / 256 + 1 : 4096);
    hipLaunchKernelGGL(bn_infer_kernel, dim3(nb), dim3(256), 0, stream, x, gamma, beta, moving_mean, moving_var, y, n, C, eps, relu);
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}

extern "C" int sar_bn_apply_f32(const float* x, const float* scale, const float* shift, float* y, long R, int C, int relu,
                                hipStream_t stream) {
    if (!x || !scale || !shift || !y || C % 4) return SAR_EINVAL;
    const long n4 = R * (C / 4);
    const int nb = (int)(n4 / 256 + 1 < 8192 ? n4 / 256 + 1 : 8192);
    hipLaunchKernelGGL(bn_apply_kernel, dim3(nb), dim3(256), 0, stream, x, scale, shift, y, n4, C, relu);
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}

// Backward of y = [relu](bn(x)) given dact = dL/dy.  dx (+)= ...; dgamma/dbeta accumulate.  `add` (optional, same
// shape) is summed into dx (residual/shortcut gradient).  dx may alias dact or add.
extern "C" int sar_bn_bwd_f32(const float* dact, const float* x, const float* scale, const float* shift,
                              const float* mean, const float* rstd, double* sums, const float* add, float* dx,
                              float* dgamma, float* dbeta, long R, int C, int relu, hipStream_t stream) {
    if (!dact || !x || !scale || !shift || !mean || !rstd || !sums || !dx || C % 4 || C > 1024) return SAR_EINVAL;
    const int tpr = pow2_tpr(C), nrg = 256 / tpr;
    long nb = (R + nrg * 4 - 1) / (nrg * 4);
    if (nb > BN_MAX_BLOCKS) nb = BN_MAX_BLOCKS;
    if (nb < 1) nb = 1;
    float* part = reinterpret_cast<float*>(sums + 2 * C);
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3((int)nb), dim3(256), 0, stream, dact, x, scale, shift, mean, rstd, part, R, C, tpr, relu);
    hipLaunchKernelGGL(bn_combine_kernel, dim3(cdiv(2 * C, 4)), dim3(256), 0, stream, part, (int)nb, sums, 2 * C);
    const long n4 = R * (C / 4);
    const int nb2 = (int)(n4 / 256 + 1 < 8192 ? n4 / 256 + 1 : 8192);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(nb2), dim3(256), 0, stream, dact, x, scale, shift, mean, rstd, sums, add, dx, dgamma, dbeta, R, C, relu);
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}

extern "C" int sar_bn_param_grads_f32(const double* sums, float* dgamma, float* dbeta, int C, hipStream_t stream) {
    if (!sums || !dgamma || !dbeta) return SAR_EINVAL;
    hipLaunchKernelGGL(bn_param_grads_kernel, dim3(cdiv(C, 256)), dim3(256), 0, stream, sums, dgamma, dbeta, C);
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}

extern "C" int sar_bn1d_fwd_f32(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                                float* moving_mean, float* moving_var, int B, int C, float eps, float momentum,
                                hipStream_t stream) {
    if (!x || !gamma || !beta || !y || !mean || !rstd || B <= 0) return SAR_EINVAL;
    hipLaunchKernelGGL(bn1d_fwd_kernel, dim3(cdiv(C, BN1_CH)), dim3(256), 0, stream, x, gamma, beta, y, mean, rstd, moving_mean, moving_var, B, C, eps, momentum);
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}

extern "C" int sar_bn1d_bwd_f32(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                                float* dx, float* dgamma, float* dbeta, int B, int C, hipStream_t stream) {
    if (!dy || !x || !gamma || !mean || !rstd || !dx || !dgamma || !dbeta) return SAR_EINVAL;
    hipLaunchKernelGGL(bn1d_bwd_kernel, dim3(cdiv(C, BN1_CH)), dim3(256), 0, stream, dy, x, gamma, mean, rstd, dx, dgamma, dbeta, B, C);
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}

extern "C" int sar_layernorm_fwd_f32(const float* x, const float* gamma, const float* beta, float* y, float* mean,
                                     float* rstd, long R, int C, float eps, hipStream_t stream) {
    if (!x || !gamma || !beta || !y || !mean || !rstd || C % 4 || C > 64 * 4 * LN_MAXV) return SAR_EINVAL;
    const int nb = (int)((R + 3) / 4 < 2048 ? (R + 3) / 4 : 2048);
    hipLaunchKernelGGL(ln_fwd_kernel, dim3(nb), dim3(256), 0, stream, x, gamma, beta, y, mean, rstd, R, C, eps);
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}

// Blocks x waves of the backward launch: every block ends in 2C atomics (R = 2496, C = 256, four waves: 32 / 64 / 128 / 256 / 512 blocks = 20.6 /
// 12.5 / 9.9 / 11.2 / 16.6 us), and with four waves 128 blocks are two waves per CU on half of the chip -- each walking 16 rows one dependent
// load-reduce-store round after the other.  Round 5: 512-thread blocks, the same number of atomics for twice the waves -- step A/B (three
// launches per step): 128 x 4 waves 2.0189 ms, 128 x 8 2.0122, 256 x 8 2.0226, 128 / 256 x 16 2.036 (SAR_LN_BWD_WAVES, SAR_LN_BWD_BLOCKS).
static int ln_bwd_launch(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd, float* dx, float* dgamma,
                         float* dbeta, long R, int C, int accumulate, int act, hipStream_t stream) {
    static const int ln_cap = getenv("SAR_LN_BWD_BLOCKS") ? atoi(getenv("SAR_LN_BWD_BLOCKS")) : 128;
    static const int nw_env = getenv("SAR_LN_BWD_WAVES") ? atoi(getenv("SAR_LN_BWD_WAVES")) : 8;
    const int nw = (nw_env == 8 || nw_env == 16) && sizeof(float) * 2 * nw_env * C <= 48 * 1024 ? nw_env : 4;   // (the cross-wave reduction's LDS)
    const long want = (R + nw - 1) / nw;
    const int nb = (int)(want < ln_cap ? want : ln_cap);
    if (nw == 16) hipLaunchKernelGGL(ln_bwd_kernel<16>, dim3(nb), dim3(1024), sizeof(float) * 32 * C, stream, dy, x, gamma, mean, rstd, dx, dgamma, dbeta, R, C, accumulate, act);
    else if (nw == 8) hipLaunchKernelGGL(ln_bwd_kernel<8>, dim3(nb), dim3(512), sizeof(float) * 16 * C, stream, dy, x, gamma, mean, rstd, dx, dgamma, dbeta, R, C, accumulate, act);
    else hipLaunchKernelGGL(ln_bwd_kernel<4>, dim3(nb), dim3(256), sizeof(float) * 8 * C, stream, dy, x, gamma, mean, rstd, dx, dgamma, dbeta, R, C, accumulate, act);
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}

extern "C" int sar_layernorm_bwd_f32(const float* dy, const float* x, const float* gamma, const float* mean,
                                     const float* rstd, float* dx, float* dgamma, float* dbeta, long R, int C,
                                     int accumulate, hipStream_t stream) {
    if (!dy || !x || !gamma || !mean || !rstd || !dx || !dgamma || !dbeta || C % 4 || C > 64 * 4 * LN_MAXV) return SAR_EINVAL;
    return ln_bwd_launch(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, R, C, accumulate, ACT_NONE, stream);
}

// LayerNormalization backward fused with the backward of the activation in FRONT of it: the reference's DS(..., 'tanh') -> LN pairs
// (model.py:253-254, 265-266, 279-280).  x = act(z) is the LayerNorm's input, so dx = d/dz = LN'(dy) * act'(x) -- what
// sar_layernorm_bwd_f32 followed by sar_act_bwd_f32 computes, without the second launch.  act: SAR_ACT_NONE / SAR_ACT_TANH.
extern "C" int sar_layernorm_bwd_act_f32(const float* dy, const float* x, const float* gamma, const float* mean,
                                         const float* rstd, float* dx, float* dgamma, float* dbeta, long R, int C,
                                         int accumulate, int act, hipStream_t stream) {
    if (!dy || !x || !gamma || !mean || !rstd || !dx || !dgamma || !dbeta || C % 4 || C > 64 * 4 * LN_MAXV) return SAR_EINVAL;
    if (act != ACT_NONE && act != ACT_TANH) return SAR_EINVAL;
    return ln_bwd_launch(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, R, C, accumulate, act, stream);
}
