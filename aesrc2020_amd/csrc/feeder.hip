// Data-feeder kernel: the per-utterance feature normalisation + padding the reference does on the host
// (utils.py:35-36 feat_norm = sklearn MinMaxScaler().fit_transform over the utterance's frames, per feature column;
//  utils.py:39-46 feat_reshape = truncate / zero-pad to max_input_len), run on the ragged batch in HBM.
//
// HBM-bound byte shuffling: one pass for the column extrema (L2-resident: an utterance is ~128 KB), one pass that writes
// the (B, T, D) batch.  Each utterance is handled by `nsplit` blocks that all recompute the extrema (cheap, from L2)
// and each write 1/nsplit of the output rows, so a 64-utterance batch still fills the chip.
#include "common.h"

// sklearn rounds the product and the sum separately.  HIP's __fmul_rn/__fadd_rn are header inlines compiled under
// -ffp-contract=fast (they fuse into v_fma_f32 after inlining), so the kernel uses plain operators under an explicit
// `#pragma clang fp contract(off)` instead.

namespace {

// Arithmetic follows sklearn's float32 path exactly: scale_ = 1 / range (range == 0 -> 1), min_ = 0 - data_min * scale_,
// X * scale_ + min_ as two separately rounded operations (no fma contraction).
__global__ __launch_bounds__(256) void feat_minmax_pad_kernel(const float* __restrict__ frames, const int* __restrict__ offsets,
                                                              float* __restrict__ out, int T, int D, int nsplit) {
#pragma clang fp contract(off)
    extern __shared__ float sm[];                    // [2][nrg][D] partial extrema, then [2][D] scale / offset
    const int b = blockIdx.x, s = blockIdx.y, tid = threadIdx.x;
    const int beg = offsets[b], n = offsets[b + 1] - beg;
    const float* src = frames + (long)beg * D;
    const int nrg = 256 / D, rg = tid / D, c = tid - rg * D;
    const bool act = rg < nrg;
    float mn = INFINITY, mx = -INFINITY;
    if (act)
        for (int r = rg; r < n; r += nrg) {
            const float v = src[(long)r * D + c];
            mn = fminf(mn, v); mx = fmaxf(mx, v);
        }
    float* smin = sm; float* smax = sm + nrg * D;
    if (act) { smin[rg * D + c] = mn; smax[rg * D + c] = mx; }
    __syncthreads();
    float* scale = sm + 2 * nrg * D; float* offs = scale + D;
    if (tid < D) {
        float a = smin[tid], z = smax[tid];
        for (int g = 1; g < nrg; ++g) { a = fminf(a, smin[g * D + tid]); z = fmaxf(z, smax[g * D + tid]); }
        float range = z - a;
        if (range == 0.f) range = 1.f;
        const float sc = 1.f / range;                 // correctly rounded (hipcc default: IEEE fp32 division)
        const float prod = a * sc;
        scale[tid] = sc; offs[tid] = 0.f - prod;
    }
    __syncthreads();
    // rows [t0, t1) of the padded output
    const int per = (T + nsplit - 1) / nsplit, t0 = s * per, t1 = min(T, t0 + per);
    float* dst = out + (long)b * T * D;
    const int keep = min(n, T);
    for (long i = (long)t0 * D + tid; i < (long)t1 * D; i += 256) {
        const int r = (int)(i / D), cc = (int)(i - (long)r * D);
        float v = 0.f;
        if (r < keep) { const float prod = src[i] * scale[cc]; v = prod + offs[cc]; }
        dst[i] = v;
    }
}

}  // namespace

extern "C" int sar_feat_minmax_pad_f32(const float* frames, const int* offsets, float* out, int B, int T, int D,
                                       hipStream_t stream) {
    if (!frames || !offsets || !out || B < 0 || T < 1 || D < 1 || D > 256) return SAR_EINVAL;
    if (B == 0) return SAR_OK;
    const int nsplit = B >= 512 ? 1 : (B >= 128 ? 2 : 4);
    const int nrg = 256 / D;
    const size_t lds = sizeof(float) * (2 * nrg * D + 2 * D);
    hipLaunchKernelGGL(feat_minmax_pad_kernel, dim3(B, nsplit), dim3(256), lds, stream, frames, offsets, out, T, D, nsplit);
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}
