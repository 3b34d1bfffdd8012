// VladPooling.call (VLAD.py:26-49) on gfx950: NetVLAD / GhostVLAD soft-assignment residual pooling.
// One workgroup per utterance; the (L, K+G) assignment matrix is staged in LDS; the reference's
// B x L x (K+G) x D broadcast tensor (VLAD.py:40-41) is never materialised:
//     V[k,d] = sum_l A[l,k] x[l,d]  -  (sum_l A[l,k]) c[k,d]        (only the K non-ghost clusters)
//     out[k,:] = V[k,:] * rsqrt(max(|V[k,:]|^2, 1e-12))               (tf.nn.l2_normalize, VLAD.py:47)
// Saved for backward: A (B,L,K+G), inv (B,K) (negative sign flags the clamped branch).
#include "common.h"

namespace {

constexpr int KC = 16;      // clusters accumulated per register pass
constexpr int XB = 13;      // feature rows fetched per batch (independent loads in flight)

// Stage the (L, K+G) score tile into LDS with coalesced loads, soft-max each row in LDS (one thread per row; rows are
// short), then write the assignment matrix back coalesced.  (v1 read the scores with per-thread serial global loads:
// ~100 dependent L2 round trips per row.)
__device__ __forceinline__ void softmax_rows_to_lds(const float* __restrict__ score, float* __restrict__ A_s,
                                                    float* __restrict__ A_out, int L, int KG) {
    for (int i = threadIdx.x; i < L * KG; i += blockDim.x) A_s[i] = score[i];
    __syncthreads();
    for (int l = threadIdx.x; l < L; l += blockDim.x) {
        float* s = A_s + l * KG;
        float m = -INFINITY;
        for (int k = 0; k < KG; ++k) m = fmaxf(m, s[k]);
        float sum = 0.f;
        for (int k = 0; k < KG; ++k) { const float e = expf(s[k] - m); s[k] = e; sum += e; }
        const float inv = 1.f / sum;
        for (int k = 0; k < KG; ++k) s[k] *= inv;
    }
    __syncthreads();
    if (A_out) for (int i = threadIdx.x; i < L * KG; i += blockDim.x) A_out[i] = A_s[i];
}

__global__ __launch_bounds__(256) void vlad_fwd_kernel(const float* __restrict__ feat, const float* __restrict__ score,
                                                       const float* __restrict__ centers, float* __restrict__ out,
                                                       float* __restrict__ A_out, float* __restrict__ inv_out, int L,
                                                       int D, int K, int KG) {
    extern __shared__ float sm[];
    float* A_s = sm;                   // [L][KG]
    float* asum = A_s + L * KG;        // [KG]
    float* ss = asum + KG;             // [K]
    float* res_s = ss + K;             // [K][D] un-normalised residuals (written once to HBM after the norms are known)
    const int b = blockIdx.x;
    const float* x = feat + (long)b * L * D;
    softmax_rows_to_lds(score + (long)b * L * KG, A_s, A_out + (long)b * L * KG, L, KG);
    for (int k = threadIdx.x; k < K; k += blockDim.x) ss[k] = 0.f;
    __syncthreads();
    for (int k = threadIdx.x; k < KG; k += blockDim.x) {
        float s = 0.f;
        for (int l = 0; l < L; ++l) s += A_s[l * KG + k];
        asum[k] = s;
    }
    __syncthreads();
    float* o = out + (long)b * K * D;
    for (int k0 = 0; k0 < K; k0 += KC) {
        for (int d0 = 0; d0 < D; d0 += blockDim.x) {           // wave-uniform trip count (shuffles below)
            const int d = d0 + threadIdx.x;
            const bool d_ok = d < D;
            float acc[KC];
#pragma unroll
            for (int i = 0; i < KC; ++i) acc[i] = 0.f;
            for (int l0 = 0; l0 < L; l0 += XB) {                 // XB independent loads in flight per batch
                float xv[XB];
#pragma unroll
                for (int j = 0; j < XB; ++j) xv[j] = (d_ok && l0 + j < L) ? x[(long)(l0 + j) * D + d] : 0.f;
#pragma unroll
                for (int j = 0; j < XB; ++j)
                    if (l0 + j < L) {
#pragma unroll
                        for (int i = 0; i < KC; ++i)
                            if (k0 + i < K) acc[i] = fmaf(A_s[(l0 + j) * KG + k0 + i], xv[j], acc[i]);
                    }
            }
#pragma unroll
            for (int i = 0; i < KC; ++i)
                if (d_ok && k0 + i < K) {
                    const float r = acc[i] - asum[k0 + i] * centers[(long)(k0 + i) * D + d];
                    res_s[(k0 + i) * D + d] = r;
                    acc[i] = r * r;
                } else acc[i] = 0.f;
            // one LDS atomic per wave and cluster (256 same-address float atomics per cluster serialise badly)
#pragma unroll
            for (int off = 32; off > 0; off >>= 1)
#pragma unroll
                for (int i = 0; i < KC; ++i) acc[i] += __shfl_xor(acc[i], off, 64);
            if ((threadIdx.x & 63) == 0)
#pragma unroll
                for (int i = 0; i < KC; ++i)
                    if (k0 + i < K) atomicAdd(&ss[k0 + i], acc[i]);
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        const float s = ss[k];
        const float iv = 1.f / sqrtf(fmaxf(s, 1e-12f));
        inv_out[(long)b * K + k] = s >= 1e-12f ? iv : -iv;
    }
    for (int i = threadIdx.x; i < K * D; i += blockDim.x) {
        const float iv = 1.f / sqrtf(fmaxf(ss[i / D], 1e-12f));
        o[i] = res_s[i] * iv;
    }
}

__global__ __launch_bounds__(256) void vlad_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ feat,
                                                       const float* __restrict__ A, const float* __restrict__ centers,
                                                       const float* __restrict__ out, const float* __restrict__ inv,
                                                       float* __restrict__ dfeat, float* __restrict__ dscore,
                                                       float* __restrict__ dcenters, int L, int D, int K, int KG,
                                                       int accumulate_dfeat) {
    extern __shared__ float sm[];
    const int DP = D + 1;
    float* A_s = sm;                    // [L][KG]
    float* dres = A_s + L * KG;         // [K][D+1]
    float* asum = dres + K * DP;        // [KG]
    float* dot = asum + KG;             // [K]   n.do
    float* cd = dot + K;                // [K]   dres.c
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* x = feat + (long)b * L * D;
    const float* dO = dout + (long)b * K * D;
    const float* n = out + (long)b * K * D;
    for (int i = tid; i < L * KG; i += 256) A_s[i] = A[(long)b * L * KG + i];
    for (int k = tid; k < K; k += 256) { dot[k] = 0.f; cd[k] = 0.f; }
    __syncthreads();
    for (int k = tid; k < KG; k += 256) {
        float s = 0.f;
        for (int l = 0; l < L; ++l) s += A_s[l * KG + k];
        asum[k] = s;
    }
    // n . dO per cluster (wave per cluster)
    for (int k = wave; k < K; k += 4) {
        float s = 0.f;
        for (int d = lane; d < D; d += 64) s = fmaf(n[(long)k * D + d], dO[(long)k * D + d], s);
        s = wave_sum(s);
        if (lane == 0) dot[k] = s;
    }
    __syncthreads();
    // dres = l2-normalise backward; dcenters; dres.c
    for (int k = wave; k < K; k += 4) {
        const float iv = inv[(long)b * K + k];
        const float aiv = fabsf(iv), dt = iv > 0.f ? dot[k] : 0.f;
        float s = 0.f;
        for (int d = lane; d < D; d += 64) {
            const float g = (dO[(long)k * D + d] - n[(long)k * D + d] * dt) * aiv;
            dres[k * DP + d] = g;
            const float c = centers[(long)k * D + d];
            s = fmaf(g, c, s);
            atomicAdd(dcenters + (long)k * D + d, -asum[k] * g);
        }
        s = wave_sum(s);
        if (lane == 0) cd[k] = s;
    }
    __syncthreads();
    // dA row by row (wave per row), then softmax backward -> dscore.  lane owns clusters lane and lane+64.
    for (int l = wave; l < L; l += 4) {
        float xr[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { const int d = lane + i * 64; xr[i] = d < D ? x[(long)l * D + d] : 0.f; }
        float dA0 = 0.f, dA1 = 0.f;
        for (int k0 = 0; k0 < K; k0 += 16) {                        // 16 independent butterfly chains per batch
            float part[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                float s = 0.f;
                if (k0 + j < K) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) { const int d = lane + i * 64; if (d < D) s = fmaf(dres[(k0 + j) * DP + d], xr[i], s); }
                }
                part[j] = s;
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1)
#pragma unroll
                for (int j = 0; j < 16; ++j) part[j] += __shfl_xor(part[j], off, 64);
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if (k0 + j < K) {
                    const float s = part[j] - cd[k0 + j];
                    if (k0 + j == lane) dA0 = s;
                    if (k0 + j == lane + 64) dA1 = s;
                }
        }
        const float a0 = lane < KG ? A_s[l * KG + lane] : 0.f;
        const float a1 = lane + 64 < KG ? A_s[l * KG + lane + 64] : 0.f;
        const float t = wave_sum(dA0 * a0 + dA1 * a1);
        if (lane < KG) dscore[((long)b * L + l) * KG + lane] = a0 * (dA0 - t);
        if (lane + 64 < KG) dscore[((long)b * L + l) * KG + lane + 64] = a1 * (dA1 - t);
    }
    // dfeat[l,d] = sum_k A[l,k] dres[k,d]
    for (int d = tid; d < D; d += 256) {
        for (int l = 0; l < L; ++l) {
            float s = 0.f;
            for (int k = 0; k < K; ++k) s = fmaf(A_s[l * KG + k], dres[k * DP + d], s);
            float* dst = dfeat + ((long)b * L + l) * D + d;
            *dst = accumulate_dfeat ? *dst + s : s;
        }
    }
}

}  // namespace

extern "C" int sar_vlad_fwd_f32(const float* feat, const float* score, const float* centers, float* out, float* A,
                                float* inv, int B, int L, int D, int K, int G, hipStream_t stream) {
    const int KG = K + G;
    if (!feat || !score || !centers || !out || !A || !inv || K < 1 || G < 0) return SAR_EINVAL;
    const size_t lds = sizeof(float) * ((size_t)L * KG + KG + K + (size_t)K * D);
    if (lds > 160 * 1024) return SAR_EINVAL;
    static size_t lds_set = 0;     // raise the dynamic-LDS cap once per size (host-side attribute, not a stream op)
    if (lds > 48 * 1024 && lds > lds_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(vlad_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return SAR_ELAUNCH;
        lds_set = 160 * 1024;
    }
    hipLaunchKernelGGL(vlad_fwd_kernel, dim3(B), dim3(256), lds, stream, feat, score, centers, out, A, inv, L, D, K, KG);
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}

extern "C" int sar_vlad_bwd_f32(const float* dout, const float* feat, const float* A, const float* centers,
                                const float* out, const float* inv, float* dfeat, float* dscore, float* dcenters, int B,
                                int L, int D, int K, int G, int accumulate_dfeat, hipStream_t stream) {
    const int KG = K + G;
    if (!dout || !feat || !A || !centers || !out || !inv || !dfeat || !dscore || !dcenters || KG > 128 || D > 512) return SAR_EINVAL;
    const size_t lds = sizeof(float) * ((size_t)L * KG + (size_t)K * (D + 1) + KG + 2 * K);
    if (lds > 160 * 1024) return SAR_EINVAL;
    static size_t lds_set = 0;
    if (lds > 48 * 1024 && lds > lds_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(vlad_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return SAR_ELAUNCH;
        lds_set = 160 * 1024;
    }
    hipLaunchKernelGGL(vlad_bwd_kernel, dim3(B), dim3(256), lds, stream, dout, feat, A, centers, out, inv, dfeat, dscore, dcenters, L, D, K, KG, accumulate_dfeat);
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}
