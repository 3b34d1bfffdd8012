// Fused L2-regularisation + Keras-Adam over the flat parameter buffer (model.py:197 Adam(lr, decay=2e-4);
// l2(1e-4) regularisers at model.py:35-48,87-104 and resnet.py:36,56,87,117; unit_norm at model.py:163).
// HBM-bound sweep: 16 B/param read + 12 B/param written.  Parameters are padded to 256-element chunks so a
// per-chunk table carries the L2 coefficient; step-dependent scalars live in a device-side state block so
// the launch can be replayed from a HIP graph.
#include "common.h"

struct SarAdamState {      // mirrored by aesrc2020_amd/_lib.py (8 x 4 bytes)
    float lr, decay, beta1, beta2, eps, lr_t;
    int iterations;
    float grad_scale;
};

namespace {

__global__ void adam_prepare_kernel(SarAdamState* st) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const float it = (float)st->iterations;
        const float lr = st->lr / (1.f + st->decay * it);
        const float t = it + 1.f;
        st->lr_t = lr * sqrtf(1.f - powf(st->beta2, t)) / (1.f - powf(st->beta1, t));
        st->iterations += 1;
    }
}

__global__ __launch_bounds__(256) void adam_l2_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                      float* __restrict__ m, float* __restrict__ v,
                                                      const float* __restrict__ chunk_l2, const SarAdamState* st,
                                                      float* __restrict__ l2_loss, long nchunks) {
    __shared__ float red[16];
    const float lr_t = st->lr_t, b1 = st->beta1, b2 = st->beta2, eps = st->eps, gs = st->grad_scale;
    float lacc = 0.f;
    for (long c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const float l2 = chunk_l2[c];
        const long i = c * 256 + threadIdx.x;
        const float w = p[i];
        const float gr = fmaf(2.f * l2, w, g[i] * gs);
        lacc = fmaf(l2 * w, w, lacc);
        const float mm = b1 * m[i] + (1.f - b1) * gr;
        const float vv = b2 * v[i] + (1.f - b2) * gr * gr;
        m[i] = mm; v[i] = vv;
        p[i] = w - lr_t * mm / (sqrtf(vv) + eps);
    }
    if (l2_loss) {
        lacc = block_sum(lacc, red);
        if (threadIdx.x == 0) atomicAdd(l2_loss, lacc);
    }
}

// sum of the l2(1e-4) regularisers for the current weights, without an update (evaluate(): Keras adds them to `loss`)
__global__ __launch_bounds__(256) void l2_penalty_kernel(const float* __restrict__ p, const float* __restrict__ chunk_l2,
                                                         float* __restrict__ out, long nchunks) {
    __shared__ float red[16];
    float lacc = 0.f;
    for (long c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const float w = p[c * 256 + threadIdx.x];
        lacc = fmaf(chunk_l2[c] * w, w, lacc);
    }
    lacc = block_sum(lacc, red);
    if (threadIdx.x == 0) atomicAdd(out, lacc);
}

// w (D, A): w[:,a] /= (1e-7 + ||w[:,a]||)
__global__ __launch_bounds__(64) void unit_norm_kernel(float* __restrict__ w, int D, int A) {
    const int a = blockIdx.x, lane = threadIdx.x;
    float s = 0.f;
    for (int d = lane; d < D; d += 64) s = fmaf(w[(long)d * A + a], w[(long)d * A + a], s);
    s = wave_sum(s);
    const float k = 1.f / (1e-7f + sqrtf(s));
    for (int d = lane; d < D; d += 64) w[(long)d * A + a] *= k;
}

}  // namespace

extern "C" int sar_adam_prepare(void* state, hipStream_t stream) {
    if (!state) return SAR_EINVAL;
    hipLaunchKernelGGL(adam_prepare_kernel, dim3(1), dim3(64), 0, stream, static_cast<SarAdamState*>(state));
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}

// n must be a multiple of 256 (the flat buffers are padded); chunk_l2 has n/256 entries.
extern "C" int sar_adam_l2_step_f32(float* params, const float* grads, float* m, float* v, const float* chunk_l2,
                                    const void* state, float* l2_loss, long n, hipStream_t stream) {
    if (!params || !grads || !m || !v || !chunk_l2 || !state || (n % 256)) return SAR_EINVAL;
    const long nchunks = n / 256;
    const int nb = (int)(nchunks < 4096 ? nchunks : 4096);
    hipLaunchKernelGGL(adam_l2_kernel, dim3(nb), dim3(256), 0, stream, params, grads, m, v, chunk_l2,
                       static_cast<const SarAdamState*>(state), l2_loss, nchunks);
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}

extern "C" int sar_l2_penalty_f32(const float* params, const float* chunk_l2, float* out, long n, hipStream_t stream) {
    if (!params || !chunk_l2 || !out || (n % 256)) return SAR_EINVAL;
    const long nchunks = n / 256;
    if (sar_zero_async(out, sizeof(float), stream) != hipSuccess) return SAR_ELAUNCH;
    if (nchunks == 0) return SAR_OK;
    hipLaunchKernelGGL(l2_penalty_kernel, dim3((int)(nchunks < 1024 ? nchunks : 1024)), dim3(256), 0, stream, params, chunk_l2, out, nchunks);
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}

extern "C" int sar_unit_norm_f32(float* w, int D, int A, hipStream_t stream) {
    if (!w) return SAR_EINVAL;
    hipLaunchKernelGGL(unit_norm_kernel, dim3(A), dim3(64), 0, stream, w, D, A);
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}
