// Accent-classification heads and losses of SAR-Net on gfx950 (all fp32, latency-bound: one wave per sample).
//  * SphereFace / CosFace / ArcFace .call([x, y]) -> softmax probabilities     losses.py:28-47 / 74-91 / 121-144
//  * circle_loss(y_true, y_pred, gamma, margin) -> per-sample loss              losses.py:157-172
//  * Keras categorical_crossentropy on probabilities (normalise, clip 1e-7)     model.py:350,354
#include "heads_margin.h"

namespace {

// grid = B, block = 64.  Saves inv_x (B) [negative = clamped], inv_w (A), cosine c (B,A).
__global__ __launch_bounds__(64) void margin_fwd_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                        const float* __restrict__ W, float* __restrict__ probs,
                                                        float* __restrict__ cosv, float* __restrict__ inv_x,
                                                        float* __restrict__ inv_w, int D, int A, int mode, float s,
                                                        float m) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const float* xr = x + (long)b * D;
    float sx = 0.f;
    for (int d = lane; d < D; d += 64) sx = fmaf(xr[d], xr[d], sx);
    sx = wave_sum(sx);
    const float ix = 1.f / sqrtf(fmaxf(sx, 1e-12f));
    float my_logit = -INFINITY, my_c = 0.f;
    for (int a = 0; a < A; ++a) {
        float sw = 0.f, dot = 0.f;
        for (int d = lane; d < D; d += 64) { const float w = W[(long)d * A + a]; sw = fmaf(w, w, sw); dot = fmaf(xr[d], w, dot); }
        sw = wave_sum(sw); dot = wave_sum(dot);
        const float iw = 1.f / sqrtf(fmaxf(sw, 1e-12f));
        const float c = dot * ix * iw;
        if (lane == a) {
            my_c = c;
            float t;
            if (mode == HEAD_COS) t = c - m;
            else {
                const float th = acosf(fminf(fmaxf(c, -1.f + KEPS), 1.f - KEPS));
                t = mode == HEAD_ARC ? cosf(th + m) : cosf(m * th);
            }
            const float yy = y[(long)b * A + a];
            my_logit = (c * (1.f - yy) + t * yy) * s;
            cosv[(long)b * A + a] = c;
            if (b == 0) inv_w[a] = sw >= 1e-12f ? iw : -iw;
        }
    }
    const float mx = wave_max(my_logit);
    const float e = lane < A ? expf(my_logit - mx) : 0.f;
    const float sum = wave_sum(e);
    if (lane < A) probs[(long)b * A + lane] = e / sum;
    if (lane == 0) inv_x[b] = sx >= 1e-12f ? ix : -ix;
    (void)my_c;
}

// grid = B, block = 64.  dwn (D,A) accumulates xn^T dc (atomics; zeroed by the launcher).
__global__ __launch_bounds__(64) void margin_bwd_kernel(const float* __restrict__ dprobs, const float* __restrict__ probs,
                                                        const float* __restrict__ cosv, const float* __restrict__ x,
                                                        const float* __restrict__ y, const float* __restrict__ W,
                                                        const float* __restrict__ inv_x, const float* __restrict__ inv_w,
                                                        float* __restrict__ dx, float* __restrict__ dwn, int D, int A,
                                                        int mode, float s, float m, int accumulate_dx) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const float p = lane < A ? probs[(long)b * A + lane] : 0.f;
    const float dp = lane < A ? dprobs[(long)b * A + lane] : 0.f;
    const float sd = wave_sum(p * dp);
    float dc = 0.f;
    if (lane < A) {
        const float dlog = p * (dp - sd);
        const float c = cosv[(long)b * A + lane], yy = y[(long)b * A + lane];
        float dt = 1.f;
        if (mode != HEAD_COS) {
            const float lo = -1.f + KEPS, hi = 1.f - KEPS;
            if (c < lo || c > hi) dt = 0.f;
            else {
                const float th = acosf(c);
                dt = mode == HEAD_ARC ? sinf(th + m) / sinf(th) : m * sinf(m * th) / sinf(th);
            }
        }
        dc = s * dlog * ((1.f - yy) + yy * dt);
    }
    const float ixs = inv_x[b], ix = fabsf(ixs);
    const float* xr = x + (long)b * D;
    // dxn[d] = sum_a dc[a] wn[d][a];  proj = xn . dxn
    float proj = 0.f;
    float dxn_l[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) dxn_l[i] = 0.f;
    for (int a = 0; a < A; ++a) {
        const float dca = __shfl(dc, a, 64);
        const float iw = fabsf(inv_w[a]);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int d = lane + i * 64;
            if (d < D) {
                const float wn = W[(long)d * A + a] * iw;
                dxn_l[i] = fmaf(dca, wn, dxn_l[i]);
                atomicAdd(dwn + (long)d * A + a, xr[d] * ix * dca);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) { const int d = lane + i * 64; if (d < D) proj = fmaf(xr[d] * ix, dxn_l[i], proj); }
    proj = wave_sum(proj);
    if (ixs < 0.f) proj = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int d = lane + i * 64;
        if (d < D) {
            const float g = (dxn_l[i] - xr[d] * ix * proj) * ix;
            float* dst = dx + (long)b * D + d;
            *dst = accumulate_dx ? *dst + g : g;
        }
    }
}

// (bodies: heads_margin.h)
template <bool CACHE>
__global__ __launch_bounds__(64) void margin_train_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                          const float* __restrict__ y_true, const float* __restrict__ dloss,
                                                          const float* __restrict__ W, float* __restrict__ probs, float* __restrict__ loss,
                                                          float* __restrict__ inv_w, float* __restrict__ dx, float* __restrict__ dwn,
                                                          int D, int A, int mode, float s, float m, int accumulate_dx) {
    margin_train_body<CACHE>(blockIdx.x, threadIdx.x, x, y, y_true, dloss, W, probs, loss, inv_w, dx + (long)blockIdx.x * D, dwn, D, A, mode, s, m, accumulate_dx);
}
template <bool CLEAR>
__global__ __launch_bounds__(64) void margin_wbwd_kernel(float* __restrict__ dwn, const float* __restrict__ W,
                                                         const float* __restrict__ inv_w, float* __restrict__ dW, int D,
                                                         int A) {
    margin_wbwd_body<CLEAR>(blockIdx.x, threadIdx.x, dwn, W, inv_w, dW, D, A);
}

// thread per sample
__global__ void circle_fwd_kernel(const float* __restrict__ y, const float* __restrict__ sp, float* __restrict__ loss,
                                  int B, int A, float gamma, float margin) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float mx = -INFINITY, ysum = 0.f, yz = 0.f;
    for (int a = 0; a < A; ++a) {
        const float s = sp[(long)b * A + a], yy = y[(long)b * A + a];
        const float ap = fmaxf(1.f + margin - s, 0.f), an = fmaxf(s + margin, 0.f);
        const float z = (yy * (ap * (s - (1.f - margin))) + (1.f - yy) * (an * (s - margin))) * gamma;
        mx = fmaxf(mx, z); ysum += yy; yz = fmaf(yy, z, yz);
    }
    float se = 0.f;
    for (int a = 0; a < A; ++a) {
        const float s = sp[(long)b * A + a], yy = y[(long)b * A + a];
        const float ap = fmaxf(1.f + margin - s, 0.f), an = fmaxf(s + margin, 0.f);
        const float z = (yy * (ap * (s - (1.f - margin))) + (1.f - yy) * (an * (s - margin))) * gamma;
        se += expf(z - mx);
    }
    loss[b] = ysum * (mx + logf(se)) - yz;
}
__global__ void circle_bwd_kernel(const float* __restrict__ y, const float* __restrict__ sp,
                                  const float* __restrict__ dloss, float* __restrict__ ds, int B, int A, float gamma,
                                  float margin) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float mx = -INFINITY, ysum = 0.f;
    for (int a = 0; a < A; ++a) {
        const float s = sp[(long)b * A + a], yy = y[(long)b * A + a];
        const float ap = fmaxf(1.f + margin - s, 0.f), an = fmaxf(s + margin, 0.f);
        const float z = (yy * (ap * (s - (1.f - margin))) + (1.f - yy) * (an * (s - margin))) * gamma;
        mx = fmaxf(mx, z); ysum += yy;
    }
    float se = 0.f;
    for (int a = 0; a < A; ++a) {
        const float s = sp[(long)b * A + a], yy = y[(long)b * A + a];
        const float ap = fmaxf(1.f + margin - s, 0.f), an = fmaxf(s + margin, 0.f);
        const float z = (yy * (ap * (s - (1.f - margin))) + (1.f - yy) * (an * (s - margin))) * gamma;
        se += expf(z - mx);
    }
    const float g = dloss[b];
    for (int a = 0; a < A; ++a) {
        const float s = sp[(long)b * A + a], yy = y[(long)b * A + a];
        const float ap = fmaxf(1.f + margin - s, 0.f), an = fmaxf(s + margin, 0.f);
        const float z = (yy * (ap * (s - (1.f - margin))) + (1.f - yy) * (an * (s - margin))) * gamma;
        const float dz = expf(z - mx) / se * ysum - yy;
        ds[(long)b * A + a] = g * dz * gamma * (yy * ap + (1.f - yy) * an);
    }
}

__global__ void cce_fwd_kernel(const float* __restrict__ y, const float* __restrict__ p, float* __restrict__ loss, int B, int A) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float S = 0.f;
    for (int a = 0; a < A; ++a) S += p[(long)b * A + a];
    float l = 0.f;
    for (int a = 0; a < A; ++a) {
        const float q = fminf(fmaxf(p[(long)b * A + a] / S, KEPS), 1.f - KEPS);
        l -= y[(long)b * A + a] * logf(q);
    }
    loss[b] = l;
}
__global__ void cce_bwd_kernel(const float* __restrict__ y, const float* __restrict__ p, const float* __restrict__ dloss,
                               float* __restrict__ dp, int B, int A) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float S = 0.f;
    for (int a = 0; a < A; ++a) S += p[(long)b * A + a];
    float acc = 0.f;
    for (int a = 0; a < A; ++a) {
        const float q = p[(long)b * A + a] / S;
        const float dq = (q >= KEPS && q <= 1.f - KEPS) ? -y[(long)b * A + a] / q : 0.f;
        acc = fmaf(dq, q, acc);
    }
    const float g = dloss[b];
    for (int a = 0; a < A; ++a) {
        const float q = p[(long)b * A + a] / S;
        const float dq = (q >= KEPS && q <= 1.f - KEPS) ? -y[(long)b * A + a] / q : 0.f;
        dp[(long)b * A + a] = g * (dq - acc) / S;
    }
}

}  // namespace

extern "C" int sar_margin_head_fwd_f32(const float* x, const float* y, const float* W, float* probs, float* cosv,
                                       float* inv_x, float* inv_w, int B, int D, int A, int mode, float s, float m,
                                       hipStream_t stream) {
    if (!x || !y || !W || !probs || !cosv || !inv_x || !inv_w || A > 64 || A < 1 || mode < 0 || mode > 2) return SAR_EINVAL;
    hipLaunchKernelGGL(margin_fwd_kernel, dim3(B), dim3(64), 0, stream, x, y, W, probs, cosv, inv_x, inv_w, D, A, mode, s, m);
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}

// dwn_ws: float[D*A] workspace. dW accumulates; dx written (or accumulated).
extern "C" int sar_margin_head_bwd_f32(const float* dprobs, const float* probs, const float* cosv, const float* x,
                                       const float* y, const float* W, const float* inv_x, const float* inv_w,
                                       float* dx, float* dW, float* dwn_ws, int B, int D, int A, int mode, float s,
                                       float m, int accumulate_dx, hipStream_t stream) {
    if (!dprobs || !probs || !cosv || !x || !y || !W || !inv_x || !inv_w || !dx || !dW || !dwn_ws || A > 64 || D > 512) return SAR_EINVAL;
    if (sar_zero_async(dwn_ws, sizeof(float) * D * A, stream) != hipSuccess) return SAR_ELAUNCH;
    hipLaunchKernelGGL(margin_bwd_kernel, dim3(B), dim3(64), 0, stream, dprobs, probs, cosv, x, y, W, inv_x, inv_w, dx, dwn_ws, D, A, mode, s, m, accumulate_dx);
    hipLaunchKernelGGL(margin_wbwd_kernel<false>, dim3(A), dim3(64), 0, stream, dwn_ws, W, inv_w, dW, D, A);
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}

// The training step of a margin head in two launches: forward (probs), categorical_crossentropy(y_true, probs) (loss), and the backward
// of both -- dx (written or accumulated), dW += -- per sample in one wave, then the column-normalisation backward.  y = the one-hot
// labels the margin is applied to (losses.py: the layer's second input), y_true = the loss target, dloss (B) = d total / d loss_b.
// inv_w: float[A] scratch.  dwn_ws: float[D*A], ZERO on entry and zero again on return (clear it once after allocating it).
extern "C" int sar_margin_head_train_f32(const float* x, const float* y, const float* y_true, const float* dloss, const float* W,
                                         float* probs, float* loss, float* inv_w, float* dx, float* dW, float* dwn_ws, int B, int D,
                                         int A, int mode, float s, float m, int accumulate_dx, hipStream_t stream) {
    if (!x || !y || !y_true || !dloss || !W || !probs || !loss || !inv_w || !dx || !dW || !dwn_ws || A > 64 || A < 1 || D > 512 || mode < 0 || mode > 2)
        return SAR_EINVAL;
    // register-cached form: A <= 8 (two 16-byte loads per kernel row: rows 16-byte aligned when A % 4 == 0), D <= 256
    if (A <= 8 && A % 4 == 0 && D <= 256 && (reinterpret_cast<uintptr_t>(W) & 15) == 0)
        hipLaunchKernelGGL(margin_train_kernel<true>, dim3(B), dim3(64), 0, stream, x, y, y_true, dloss, W, probs, loss, inv_w, dx, dwn_ws, D, A, mode, s, m, accumulate_dx);
    else
        hipLaunchKernelGGL(margin_train_kernel<false>, dim3(B), dim3(64), 0, stream, x, y, y_true, dloss, W, probs, loss, inv_w, dx, dwn_ws, D, A, mode, s, m, accumulate_dx);
    hipLaunchKernelGGL(margin_wbwd_kernel<true>, dim3(A), dim3(64), 0, stream, dwn_ws, W, inv_w, dW, D, A);
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}

extern "C" int sar_circle_loss_fwd_f32(const float* y, const float* s, float* loss, int B, int A, float gamma, float margin, hipStream_t stream) {
    if (!y || !s || !loss) return SAR_EINVAL;
    hipLaunchKernelGGL(circle_fwd_kernel, dim3(cdiv(B, 64)), dim3(64), 0, stream, y, s, loss, B, A, gamma, margin);
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}
extern "C" int sar_circle_loss_bwd_f32(const float* y, const float* s, const float* dloss, float* ds, int B, int A, float gamma, float margin, hipStream_t stream) {
    if (!y || !s || !dloss || !ds) return SAR_EINVAL;
    hipLaunchKernelGGL(circle_bwd_kernel, dim3(cdiv(B, 64)), dim3(64), 0, stream, y, s, dloss, ds, B, A, gamma, margin);
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}
extern "C" int sar_cce_fwd_f32(const float* y, const float* p, float* loss, int B, int A, hipStream_t stream) {
    if (!y || !p || !loss) return SAR_EINVAL;
    hipLaunchKernelGGL(cce_fwd_kernel, dim3(cdiv(B, 64)), dim3(64), 0, stream, y, p, loss, B, A);
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}
extern "C" int sar_cce_bwd_f32(const float* y, const float* p, const float* dloss, float* dp, int B, int A, hipStream_t stream) {
    if (!y || !p || !dloss || !dp) return SAR_EINVAL;
    hipLaunchKernelGGL(cce_bwd_kernel, dim3(cdiv(B, 64)), dim3(64), 0, stream, y, p, dloss, dp, B, A);
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}
