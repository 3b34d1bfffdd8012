// Margin-softmax head bodies shared by heads.hip (one launch per piece) and heads_fused.hip (the accent classifier and the margin head's
// training step in ONE launch, round 5): losses.py:28-47 / 74-91 / 121-144 + Keras categorical_crossentropy (model.py:354).
#pragma once
#include "common.h"

namespace {

enum { HEAD_SPHERE = 0, HEAD_COS = 1, HEAD_ARC = 2 };
constexpr float KEPS = 1e-7f;

// One sample's class-level arithmetic on one wave (lane a < A owns class a; c = its cosine): margin logits -> softmax -> Keras
// categorical_crossentropy -> the backward of all three; writes probs / loss, returns d loss / d cos of the lane's class.
__device__ __forceinline__ float margin_row_math(const int b, const int lane, const int A, const int mode, const float s, const float m,
                                                 const float c, const float yy, const float yt, const float gl,
                                                 float* __restrict__ probs, float* __restrict__ loss) {
    float my_logit = -INFINITY;
    if (lane < A) {
        float t;
        if (mode == HEAD_COS) t = c - m;
        else {
            const float th = acosf(fminf(fmaxf(c, -1.f + KEPS), 1.f - KEPS));
            t = mode == HEAD_ARC ? cosf(th + m) : cosf(m * th);
        }
        my_logit = (c * (1.f - yy) + t * yy) * s;
    }
    const float mx = wave_max(my_logit);
    const float e = lane < A ? expf(my_logit - mx) : 0.f;
    const float p = e / wave_sum(e);
    if (lane < A) probs[(long)b * A + lane] = p;
    // categorical_crossentropy(y_true, p): renormalise, clip (cce_fwd_kernel / cce_bwd_kernel)
    const float S = wave_sum(p), q = p / S;
    const float l = wave_sum(lane < A ? -yt * logf(fminf(fmaxf(q, KEPS), 1.f - KEPS)) : 0.f);
    if (lane == 0) loss[b] = l;
    const float dq = (lane < A && q >= KEPS && q <= 1.f - KEPS) ? -yt / q : 0.f;
    const float acc = wave_sum(dq * q);
    const float dp = lane < A ? gl * (dq - acc) / S : 0.f;
    // margin_bwd_kernel
    const float sd = wave_sum(p * dp);
    float dc = 0.f;
    if (lane < A) {
        const float dlog = p * (dp - sd);
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
    return dc;
}

// Forward + Keras categorical_crossentropy + backward of one sample in ONE wave (margin_fwd_kernel, cce_fwd / cce_bwd, margin_bwd_kernel
// back to back on the same lanes: lane a < A owns class a): the training step's six launches around the head become two (this + the
// column-normalisation backward).  grid = B, block = 64.  dwn (D,A): zero on entry (margin_wbwd_kernel<true> leaves it zero again).
// CACHE: the wave keeps its slice of W in registers (A <= 8, D <= 256: the reference's 8 accents on a 256-wide embedding) -- every
// operand of the sample is requested in ONE memory phase; looping over the classes with the loads inside was ~15 dependent round trips
// (29 us for 4 KB of arithmetic).
// dxrow: the sample's row of the input gradient; dwn: the (D,A) sum over the samples of the normalised-kernel gradient, added with fp32
// atomics -- both may be LDS (heads_fused.hip, round 5: the accent classifier's workgroup runs the margin head of its rows on spare waves).
template <bool CACHE>
__device__ __forceinline__ void margin_train_body(const int b, const int lane, const float* __restrict__ x, const float* __restrict__ y,
                                                  const float* __restrict__ y_true, const float* __restrict__ dloss,
                                                  const float* __restrict__ W, float* __restrict__ probs, float* __restrict__ loss,
                                                  float* __restrict__ inv_w, float* dxrow, float* dwn,
                                                  int D, int A, int mode, float s, float m, int accumulate_dx) {
    const float* xr = x + (long)b * D;
    float xv[4], wv[4][8], dxo[4];
    if constexpr (CACHE) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int d = min(lane + 64 * i, D - 1);
            xv[i] = lane + 64 * i < D ? xr[d] : 0.f;
            dxo[i] = accumulate_dx ? dxrow[d] : 0.f;
            const float4 w0 = *reinterpret_cast<const float4*>(W + (long)d * A), w1 = A > 4 ? *reinterpret_cast<const float4*>(W + (long)d * A + 4) : make_float4(0, 0, 0, 0);
            const float ws[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
            for (int a = 0; a < 8; ++a) wv[i][a] = (lane + 64 * i < D && a < A) ? ws[a] : 0.f;
        }
    }
    const float yy = lane < A ? y[(long)b * A + lane] : 0.f;
    const float yt = lane < A ? y_true[(long)b * A + lane] : 0.f;
    const float gl = dloss[b];
    float sx = 0.f;
    if constexpr (CACHE) { for (int i = 0; i < 4; ++i) sx = fmaf(xv[i], xv[i], sx); }
    else for (int d = lane; d < D; d += 64) sx = fmaf(xr[d], xr[d], sx);
    sx = wave_sum(sx);
    const float ix = 1.f / sqrtf(fmaxf(sx, 1e-12f));
    float c = 0.f, my_iw = 0.f;
    auto fwd_class = [&](int a, float sw, float dot) __attribute__((always_inline)) {
        sw = wave_sum(sw); dot = wave_sum(dot);
        const float iw = 1.f / sqrtf(fmaxf(sw, 1e-12f));
        if (lane == a) {
            c = dot * ix * iw; my_iw = iw;
            if (b == 0) inv_w[a] = sw >= 1e-12f ? iw : -iw;
        }
    };
    if constexpr (CACHE) {
#pragma unroll
        for (int a = 0; a < 8; ++a) {
            float sw = 0.f, dot = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) { sw = fmaf(wv[i][a], wv[i][a], sw); dot = fmaf(xv[i], wv[i][a], dot); }
            if (a < A) fwd_class(a, sw, dot);                       // wave-uniform
        }
    } else {
        for (int a = 0; a < A; ++a) {
            float sw = 0.f, dot = 0.f;
            for (int d = lane; d < D; d += 64) { const float w = W[(long)d * A + a]; sw = fmaf(w, w, sw); dot = fmaf(xr[d], w, dot); }
            fwd_class(a, sw, dot);
        }
    }
    const float dc = margin_row_math(b, lane, A, mode, s, m, c, yy, yt, gl, probs, loss);
    float proj = 0.f;
    float dxn_l[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) dxn_l[i] = 0.f;
    if constexpr (CACHE) {
#pragma unroll
        for (int a = 0; a < 8; ++a) {
            if (a < A) {                                            // wave-uniform
                const float dca = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dc), a));
                const float iw = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_iw), a));
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int d = lane + i * 64;
                    if (d < D) {
                        dxn_l[i] = fmaf(dca, wv[i][a] * iw, dxn_l[i]);
                        atomicAdd(dwn + (long)d * A + a, xv[i] * ix * dca);
                    }
                }
            }
        }
    } else {
        for (int a = 0; a < A; ++a) {
            const float dca = __shfl(dc, a, 64), iw = __shfl(my_iw, a, 64);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int d = lane + i * 64;
                if (d < D) {
                    dxn_l[i] = fmaf(dca, W[(long)d * A + a] * iw, dxn_l[i]);
                    atomicAdd(dwn + (long)d * A + a, xr[d] * ix * dca);
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < (CACHE ? 4 : 8); ++i) { const int d = lane + i * 64; if (d < D) proj = fmaf((CACHE ? xv[i < 4 ? i : 0] : xr[d]) * ix, dxn_l[i], proj); }
    proj = wave_sum(proj);
    if (sx < 1e-12f) proj = 0.f;
#pragma unroll
    for (int i = 0; i < (CACHE ? 4 : 8); ++i) {
        const int d = lane + i * 64;
        if (d < D) {
            const float xd = CACHE ? xv[i < 4 ? i : 0] : xr[d];
            const float g = (dxn_l[i] - xd * ix * proj) * ix;
            float* dst = dxrow + d;
            if (CACHE) *dst = dxo[i < 4 ? i : 0] + g;
            else *dst = accumulate_dx ? *dst + g : g;
        }
    }
}

// column-normalisation backward: dW[:,a] += (dwn[:,a] - wn[:,a] (wn[:,a].dwn[:,a])) * inv_w[a];  grid = A, block = 64
// CLEAR: the workspace is left zero for the next step (saves the zeroing launch in front of the next margin_train_kernel)
template <bool CLEAR>
__device__ __forceinline__ void margin_wbwd_body(const int a, const int lane, float* __restrict__ dwn, const float* __restrict__ W,
                                                 const float* __restrict__ inv_w, float* __restrict__ dW, int D, int A) {
    const float iws = inv_w[a], iw = fabsf(iws);
    float proj = 0.f;
    for (int d = lane; d < D; d += 64) proj = fmaf(W[(long)d * A + a] * iw, dwn[(long)d * A + a], proj);
    proj = wave_sum(proj);
    if (iws < 0.f) proj = 0.f;
    for (int d = lane; d < D; d += 64) {
        dW[(long)d * A + a] += (dwn[(long)d * A + a] - W[(long)d * A + a] * iw * proj) * iw;
        if (CLEAR) dwn[(long)d * A + a] = 0.f;
    }
}

}  // namespace
