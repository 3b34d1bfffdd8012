// The accent classifier of SAR-Net in ONE launch, forward + loss + backward:
//   AR_CF_DS1 = Dense(64, relu) -> AR_CF_DS2 = Dense(64, relu) -> y_accent = Dense(A, softmax)   (model.py:150-152 via DS, model.py:35-42)
//   loss = Keras categorical_crossentropy(y_true, y_accent)                                      (model.py:350)
// On a 64-utterance batch these are 64 x 256 x 64, 64 x 64 x 64 and 64 x 64 x 8 products: as separate launches (3 GEMMs, softmax,
// cross-entropy forward; cross-entropy / softmax backward, 6 GEMMs, 2 activation backwards) they were 15 launches of ~5-14 us each
// around ~1 us of arithmetic -- 110 us of a 2.4 ms step spent on dispatch latency.  Here a workgroup owns four rows and walks the whole
// chain with the intermediates in LDS; the weight gradients of the rows leave as fp32 atomics.
// fp32 FMA arithmetic throughout (also in the "bf16" mode, whose GEMMs would round these operands).
#include "common.h"

namespace {

constexpr int FH_ROWS = 4;      // rows per workgroup: 16 workgroups for the reference's batch of 64 -- ONE workgroup of 64 rows ran 180 us
                                // (3.5 M multiply-adds at the 64 per cycle of a single CU, operands from L2), slower than the launches it replaced
constexpr int FH_DMAX = 64;     // max width of the two hidden layers
constexpr int FH_AMAX = 32;     // max classes
constexpr int FH_NT = 1024;    // sixteen waves: the loops are chains of L2 loads, and waves are what hides them on a CU that runs one block
constexpr float FH_KEPS = 1e-7f;

struct FusedHeadP {
    const float* x; const float* W1; const float* b1; const float* W2; const float* b2; const float* W3; const float* b3;
    const float* y_true; const float* dloss;
    float* h1; float* h2; float* probs; float* loss;
    float* dW1; float* db1; float* dW2; float* db2; float* dW3; float* db3; float* dx;
    int B, H, D1, D2, A;
};

__global__ __launch_bounds__(FH_NT) void accent_classifier_kernel(const FusedHeadP p) {
    __shared__ float h1s[FH_ROWS][FH_DMAX + 1], h2s[FH_ROWS][FH_DMAX + 1], g1s[FH_ROWS][FH_DMAX + 1], g2s[FH_ROWS][FH_DMAX + 1];
    __shared__ float z3s[FH_ROWS][FH_AMAX + 1], g3s[FH_ROWS][FH_AMAX + 1];
    __shared__ float pq[4][FH_ROWS][FH_DMAX + 1];                   // the four k-quarters of the first product
    extern __shared__ __attribute__((aligned(16))) float dyn[];
    const int H = p.H, D1 = p.D1, D2 = p.D2, A = p.A, P1 = D1 + 1, P2 = D2 + 1, P3 = A + 1;
    float* W1s = dyn;                  // [H][D1 + 1]
    float* W2s = W1s + H * P1;         // [D1][D2 + 1]
    float* W3s = W2s + D1 * P2;        // [D2][A + 1]
    float* xs = W3s + D2 * P3;         // [FH_ROWS][H]: the block's input rows
    const int tid = threadIdx.x;
    const int r0 = blockIdx.x * FH_ROWS, nr = min(FH_ROWS, p.B - r0);
    const bool train = p.y_true != nullptr && p.dloss != nullptr;

    // Everything the chain reads from global memory, once, with all loads in flight: the three kernels (82 KB) and the block's rows.
    // (Read where they are used -- in the phase loops, from L2 -- every phase began with a memory round trip nothing could hide on a CU
    // that runs one block: 129 us with one load pair per trip, 51 us unrolled 16 deep.)
    for (int i = tid; i < H * D1; i += FH_NT) W1s[(i / D1) * P1 + i % D1] = p.W1[i];
    for (int i = tid; i < D1 * D2; i += FH_NT) W2s[(i / D2) * P2 + i % D2] = p.W2[i];
    for (int i = tid; i < D2 * A; i += FH_NT) W3s[(i / A) * P3 + i % A] = p.W3[i];
    for (int i = tid; i < FH_ROWS * H; i += FH_NT) xs[i] = p.x[(long)(r0 + min(i / H, nr - 1)) * H + i % H];
    __syncthreads();
    // ---- h1 = relu(x W1 + b1): thread = (column j, quarter q of the k range), all four rows per thread; the quarters meet in LDS
    for (int o = tid; o < D1 * 4; o += FH_NT) {
        const int j = o % D1, q = o / D1;
        const int k0 = (H * q) / 4, k1 = (H * (q + 1)) / 4;
        float a[FH_ROWS];
#pragma unroll
        for (int r = 0; r < FH_ROWS; ++r) a[r] = 0.f;
#pragma unroll 8
        for (int k = k0; k < k1; ++k) {
            const float wv = W1s[k * P1 + j];
#pragma unroll
            for (int r = 0; r < FH_ROWS; ++r) a[r] = fmaf(xs[r * H + k], wv, a[r]);
        }
#pragma unroll
        for (int r = 0; r < FH_ROWS; ++r) pq[q][r][j] = a[r];
    }
    __syncthreads();
    for (int o = tid; o < D1 * FH_ROWS; o += FH_NT) {
        const int j = o % D1, r = o / D1;
        h1s[r][j] = fmaxf((pq[0][r][j] + pq[1][r][j]) + (pq[2][r][j] + pq[3][r][j]) + p.b1[j], 0.f);
    }
    __syncthreads();
    // ---- h2 = relu(h1 W2 + b2)
    for (int o = tid; o < D2 * FH_ROWS; o += FH_NT) {
        const int j = o % D2, r = o / D2;
        float a = p.b2[j];
#pragma unroll 8
        for (int k = 0; k < D1; ++k) a = fmaf(h1s[r][k], W2s[k * P2 + j], a);
        h2s[r][j] = fmaxf(a, 0.f);
    }
    __syncthreads();
    // ---- logits, softmax, cross-entropy (+ its gradient w.r.t. the logits)
    for (int o = tid; o < A * FH_ROWS; o += FH_NT) {
        const int a_ = o % A, r = o / A;
        float a = p.b3[a_];
#pragma unroll 8
        for (int k = 0; k < D2; ++k) a = fmaf(h2s[r][k], W3s[k * P3 + a_], a);
        z3s[r][a_] = a;
    }
    for (int o = tid; o < FH_ROWS * D1; o += FH_NT) if (o / D1 < nr) p.h1[(long)(r0 + o / D1) * D1 + o % D1] = h1s[o / D1][o % D1];
    for (int o = tid; o < FH_ROWS * D2; o += FH_NT) if (o / D2 < nr) p.h2[(long)(r0 + o / D2) * D2 + o % D2] = h2s[o / D2][o % D2];
    __syncthreads();
    if (tid < nr) {
        const int r = tid;
        float* z = z3s[r];
        float m = -INFINITY;
        for (int a = 0; a < A; ++a) m = fmaxf(m, z[a]);
        float s = 0.f;
        for (int a = 0; a < A; ++a) { const float e = expf(z[a] - m); z[a] = e; s += e; }
        const float inv = 1.f / s;
        float S = 0.f;
        for (int a = 0; a < A; ++a) { z[a] *= inv; p.probs[(long)(r0 + r) * A + a] = z[a]; S += z[a]; }
        if (p.y_true != nullptr) {
            const float* y = p.y_true + (long)(r0 + r) * A;
            // Keras categorical_crossentropy on probabilities: renormalise, clip to [1e-7, 1 - 1e-7] (heads.hip: cce_fwd / cce_bwd)
            float l = 0.f, acc = 0.f;
            for (int a = 0; a < A; ++a) {
                const float q = z[a] / S;
                l -= y[a] * logf(fminf(fmaxf(q, FH_KEPS), 1.f - FH_KEPS));
                const float dq = (q >= FH_KEPS && q <= 1.f - FH_KEPS) ? -y[a] / q : 0.f;
                acc = fmaf(dq, q, acc);
            }
            p.loss[r0 + r] = l;
            if (train) {
                const float g = p.dloss[r0 + r];
                float sd = 0.f;
                for (int a = 0; a < A; ++a) {
                    const float q = z[a] / S;
                    const float dq = (q >= FH_KEPS && q <= 1.f - FH_KEPS) ? -y[a] / q : 0.f;
                    const float dpa = g * (dq - acc) / S;
                    g3s[r][a] = dpa;
                    sd = fmaf(dpa, z[a], sd);
                }
                for (int a = 0; a < A; ++a) g3s[r][a] = z[a] * (g3s[r][a] - sd);       // softmax backward
            }
        }
    } else if (tid < FH_ROWS && train) {
        for (int a = 0; a < A; ++a) g3s[tid][a] = 0.f;                              // rows past the batch
    }
    if (!train) return;
    __syncthreads();
    // ---- y_accent: dW3 += h2^T dz3, db3 += colsum dz3; dz2 = (dz3 W3^T) * [h2 > 0]
    for (int o = tid; o < D2 * A; o += FH_NT) {
        const int a_ = o % A, k = o / A;
        float s = 0.f;
        for (int r = 0; r < nr; ++r) s = fmaf(h2s[r][k], g3s[r][a_], s);
        atomicAdd(p.dW3 + k * A + a_, s);
    }
    if (tid < A) {
        float s = 0.f;
        for (int r = 0; r < nr; ++r) s += g3s[r][tid];
        atomicAdd(p.db3 + tid, s);
    }
    for (int o = tid; o < FH_ROWS * D2; o += FH_NT) {
        const int k = o % D2, r = o / D2;
        float s = 0.f;
        for (int a = 0; a < A; ++a) s = fmaf(g3s[r][a], W3s[k * P3 + a], s);
        g2s[r][k] = h2s[r][k] > 0.f ? s : 0.f;
    }
    __syncthreads();
    // ---- AR_CF_DS2: dW2 += h1^T dz2, db2; dz1 = (dz2 W2^T) * [h1 > 0]
    for (int o = tid; o < D1 * D2; o += FH_NT) {
        const int j = o % D2, k = o / D2;
        float s = 0.f;
        for (int r = 0; r < nr; ++r) s = fmaf(h1s[r][k], g2s[r][j], s);
        atomicAdd(p.dW2 + k * D2 + j, s);
    }
    if (tid < D2) {
        float s = 0.f;
        for (int r = 0; r < nr; ++r) s += g2s[r][tid];
        atomicAdd(p.db2 + tid, s);
    }
    for (int o = tid; o < FH_ROWS * D1; o += FH_NT) {
        const int k = o % D1, r = o / D1;
        float s = 0.f;
#pragma unroll 8
        for (int j = 0; j < D2; ++j) s = fmaf(g2s[r][j], W2s[k * P2 + j], s);
        g1s[r][k] = (r < nr && h1s[r][k] > 0.f) ? s : 0.f;
    }
    __syncthreads();
    // ---- AR_CF_DS1: dW1 += x^T dz1, db1; dx = dz1 W1^T (written, not accumulated: it opens the gradient of the embedding)
    for (int o = tid; o < H * D1; o += FH_NT) {
        const int j = o % D1, k = o / D1;
        float s = 0.f;
        for (int r = 0; r < nr; ++r) s = fmaf(xs[r * H + k], g1s[r][j], s);
        atomicAdd(p.dW1 + k * D1 + j, s);
    }
    if (tid < D1) {
        float s = 0.f;
        for (int r = 0; r < nr; ++r) s += g1s[r][tid];
        atomicAdd(p.db1 + tid, s);
    }
    for (int o = tid; o < nr * H; o += FH_NT) {
        const int k = o % H, r = o / H;
        float s = 0.f;
#pragma unroll 8
        for (int j = 0; j < D1; ++j) s = fmaf(g1s[r][j], W1s[k * P1 + j], s);         // row k at pitch D1 + 1: conflict-free across the lanes
        p.dx[(long)(r0 + r) * H + k] = s;
    }
}

}  // namespace

// include/sarnet.h.  y_true NULL: forward only (h1, h2, probs); dloss NULL: forward + per-sample loss; both: + backward (parameter
// gradients accumulate, dx is written).  D1, D2 <= 64, A <= 32.
extern "C" int sar_accent_classifier_f32(const float* x, const float* W1, const float* b1, const float* W2, const float* b2, const float* W3,
                                         const float* b3, const float* y_true, const float* dloss, float* h1, float* h2, float* probs,
                                         float* loss, float* dW1, float* db1, float* dW2, float* db2, float* dW3, float* db3, float* dx,
                                         int B, int H, int D1, int D2, int A, hipStream_t stream) {
    if (!x || !W1 || !b1 || !W2 || !b2 || !W3 || !b3 || !h1 || !h2 || !probs || B < 1) return SAR_EINVAL;
    if (H < 1 || D1 < 1 || D1 > FH_DMAX || D2 < 1 || D2 > FH_DMAX || A < 1 || A > FH_AMAX) return SAR_EINVAL;
    if (y_true && !loss) return SAR_EINVAL;
    if (y_true && dloss && (!dW1 || !db1 || !dW2 || !db2 || !dW3 || !db3 || !dx)) return SAR_EINVAL;
    FusedHeadP p{x, W1, b1, W2, b2, W3, b3, y_true, dloss, h1, h2, probs, loss, dW1, db1, dW2, db2, dW3, db3, dx, B, H, D1, D2, A};
    const size_t lds = ((size_t)H * (D1 + 1) + (size_t)D1 * (D2 + 1) + (size_t)D2 * (A + 1) + (size_t)FH_ROWS * H) * sizeof(float);
    if (lds > 140 * 1024) return SAR_EINVAL;               // H <= 480 at D1 = 64
    static PerDeviceFlag attr;
    if (!attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(accent_classifier_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024) != hipSuccess)
            return SAR_ELAUNCH;
        attr = true;
    }
    hipLaunchKernelGGL(accent_classifier_kernel, dim3(cdiv(B, FH_ROWS)), dim3(FH_NT), lds, stream, p);
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}
