// K.ctc_batch_cost (model.py:62-64) on gfx950: one workgroup per utterance.
// Semantics: logits = log(y_pred + 1e-7) -> tf.nn.ctc_loss with its internal log-softmax,
// blank = V-1, ctc_merge_repeated=True.  So the effective emission is y_t(k) = (p_t(k)+eps) / (sum_j p_t(j) + V*eps).
// fwd : loss[b] = -log P(label | y);  log-alpha (B, L, Smax) and log-denominators (B, L) go to the workspace
// bwd : dp[b,t,k] = dloss[b] * ( 1/den_t - occ_t(k) / (p_t(k)+eps) ),  occ = state posteriors summed per symbol;
//       frames t >= in_len[b] get zero gradient.  Labels arrive as float32 (utils.py:107), cast here.
// Infeasible labelings (TF raises) produce loss = +inf and a zero gradient.
#include "common.h"

namespace {

constexpr float CTC_EPS = 1e-7f;
constexpr int CTC_SMAX = 512;      // 2*max_label_len+1 (reference: 2*72+1 = 145)

__device__ __forceinline__ int ext_label(const float* lab, int s, int blank) {
    return (s & 1) ? (int)lab[s >> 1] : blank;
}

__global__ __launch_bounds__(256) void ctc_fwd_kernel(const float* __restrict__ probs, const float* __restrict__ labels,
                                                      const int* __restrict__ in_len, const int* __restrict__ lab_len,
                                                      float* __restrict__ loss, float* __restrict__ alpha_ws,
                                                      float* __restrict__ logden_ws, int L, int V, int maxU, int Smax) {
    __shared__ float red[16];
    __shared__ float al[2][CTC_SMAX];
    const int b = blockIdx.x, tid = threadIdx.x;
    int T = in_len[b]; if (T > L) T = L;
    const int U = min(lab_len[b], maxU), S = 2 * U + 1, blank = V - 1;
    const float* p = probs + (long)b * L * V;
    const float* lab = labels + (long)b * maxU;
    float* aw = alpha_ws + (long)b * L * Smax;
    float* ld = logden_ws + (long)b * L;
    // log denominators
    for (int t = 0; t < T; ++t) {
        float s = 0.f;
        for (int v = tid; v < V; v += 256) s += p[(long)t * V + v] + CTC_EPS;
        s = block_sum(s, red);
        if (tid == 0) ld[t] = logf(s);
    }
    __syncthreads();
    // state-local constants (each thread owns states tid, tid+256)
    for (int s = tid; s < S; s += 256) {
        float a = -INFINITY;
        if (s < 2 && T > 0) a = logf(p[ext_label(lab, s, blank)] + CTC_EPS) - ld[0];
        al[0][s] = a; aw[s] = a;
    }
    __syncthreads();
    for (int t = 1; t < T; ++t) {
        const int cur = t & 1, prv = cur ^ 1;
        const float ldt = ld[t];
        for (int s = tid; s < S; s += 256) {
            const int k = ext_label(lab, s, blank);
            float a = al[prv][s];
            if (s >= 1) a = lse2(a, al[prv][s - 1]);
            if (s >= 2 && k != blank && k != ext_label(lab, s - 2, blank)) a = lse2(a, al[prv][s - 2]);
            a += logf(p[(long)t * V + k] + CTC_EPS) - ldt;
            al[cur][s] = a; aw[(long)t * Smax + s] = a;
        }
        __syncthreads();
    }
    if (tid == 0) {
        float ll = -INFINITY;
        if (T > 0) {
            const int cur = (T - 1) & 1;
            ll = al[cur][S - 1];
            if (S > 1) ll = lse2(ll, al[cur][S - 2]);
        }
        loss[b] = -ll;
    }
}

__global__ __launch_bounds__(256) void ctc_bwd_kernel(const float* __restrict__ probs, const float* __restrict__ labels,
                                                      const int* __restrict__ in_len, const int* __restrict__ lab_len,
                                                      const float* __restrict__ loss, const float* __restrict__ dloss,
                                                      const float* __restrict__ alpha_ws,
                                                      const float* __restrict__ logden_ws, float* __restrict__ dprobs,
                                                      int L, int V, int maxU, int Smax) {
    extern __shared__ float occ[];               // [V]
    __shared__ float be[2][CTC_SMAX];            // beta (excludes the emission at t)
    const int b = blockIdx.x, tid = threadIdx.x;
    int T = in_len[b]; if (T > L) T = L;
    const int U = min(lab_len[b], maxU), S = 2 * U + 1, blank = V - 1;
    const float* p = probs + (long)b * L * V;
    const float* lab = labels + (long)b * maxU;
    const float* aw = alpha_ws + (long)b * L * Smax;
    const float* ld = logden_ws + (long)b * L;
    float* dp = dprobs + (long)b * L * V;
    const float ll = -loss[b];
    const float g = dloss[b];
    const bool feasible = (ll > -INFINITY) && (ll < INFINITY);
    // frames beyond the input length (and everything when infeasible) get zero gradient
    for (long i = (long)(feasible ? T : 0) * V + tid; i < (long)L * V; i += 256) dp[i] = 0.f;
    if (!feasible) return;
    for (int t = T - 1; t >= 0; --t) {
        const int cur = t & 1, nxt = cur ^ 1;
        for (int v = tid; v < V; v += 256) occ[v] = 0.f;
        for (int s = tid; s < S; s += 256) {
            float bb;
            if (t == T - 1) {
                bb = (s >= S - 2) ? 0.f : -INFINITY;
            } else {
                const float ldn = ld[t + 1];
                const int k0 = ext_label(lab, s, blank);
                bb = be[nxt][s] + logf(p[(long)(t + 1) * V + k0] + CTC_EPS) - ldn;
                if (s + 1 < S) {
                    const int k1 = ext_label(lab, s + 1, blank);
                    bb = lse2(bb, be[nxt][s + 1] + logf(p[(long)(t + 1) * V + k1] + CTC_EPS) - ldn);
                }
                if (s + 2 < S) {
                    const int k2 = ext_label(lab, s + 2, blank);
                    if (k2 != blank && k2 != k0) bb = lse2(bb, be[nxt][s + 2] + logf(p[(long)(t + 1) * V + k2] + CTC_EPS) - ldn);
                }
            }
            be[cur][s] = bb;
        }
        __syncthreads();                          // occ zeroed, beta[cur] complete
        for (int s = tid; s < S; s += 256) {
            const float w = expf(aw[(long)t * Smax + s] + be[cur][s] - ll);
            if (w > 0.f) atomicAdd(&occ[ext_label(lab, s, blank)], w);
        }
        __syncthreads();
        const float inv_den = expf(-ld[t]);
        for (int v = tid; v < V; v += 256) dp[(long)t * V + v] = g * (inv_den - occ[v] / (p[(long)t * V + v] + CTC_EPS));
        __syncthreads();
    }
}

// K.ctc_decode(greedy=True) (model.py:385-389): argmax per frame, merge repeats, drop blank.  One wave per utterance.
__global__ void ctc_greedy_kernel(const float* __restrict__ probs, const int* __restrict__ in_len,
                                  int* __restrict__ decoded, int* __restrict__ dec_len, int L, int V) {
    const int b = blockIdx.x, lane = threadIdx.x;
    int T = in_len[b]; if (T > L) T = L;
    int prev = -1, n = 0;
    for (int t = 0; t < T; ++t) {
        const float* p = probs + ((long)b * L + t) * V;
        float best = -INFINITY; int bi = 0;
        for (int v = lane; v < V; v += 64) { const float x = p[v]; if (x > best) { best = x; bi = v; } }
        for (int o = 32; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        if (bi != prev && bi != V - 1) { if (lane == 0) decoded[(long)b * L + n] = bi; ++n; }
        prev = bi;
    }
    if (lane == 0) { dec_len[b] = n; for (int i = n; i < L; ++i) decoded[(long)b * L + i] = -1; }
}

}  // namespace

extern "C" size_t sar_ctc_workspace_bytes(int B, int L, int max_label_len) {
    const int Smax = 2 * max_label_len + 1;
    return sizeof(float) * ((size_t)B * L * Smax + (size_t)B * L);
}

extern "C" int sar_ctc_batch_cost_fwd_f32(const float* probs, const float* labels, const int* in_len, const int* lab_len,
                                          float* loss, void* workspace, size_t workspace_bytes, int B, int L, int V,
                                          int max_label_len, hipStream_t stream) {
    const int Smax = 2 * max_label_len + 1;
    if (!probs || !labels || !in_len || !lab_len || !loss || !workspace || Smax > CTC_SMAX ||
        workspace_bytes < sar_ctc_workspace_bytes(B, L, max_label_len))
        return SAR_EINVAL;
    float* alpha = static_cast<float*>(workspace);
    float* logden = alpha + (size_t)B * L * Smax;
    hipLaunchKernelGGL(ctc_fwd_kernel, dim3(B), dim3(256), 0, stream, probs, labels, in_len, lab_len, loss, alpha, logden, L, V, max_label_len, Smax);
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}

extern "C" int sar_ctc_batch_cost_bwd_f32(const float* probs, const float* labels, const int* in_len, const int* lab_len,
                                          const float* loss, const float* dloss, const void* workspace,
                                          size_t workspace_bytes, float* dprobs, int B, int L, int V, int max_label_len,
                                          hipStream_t stream) {
    const int Smax = 2 * max_label_len + 1;
    if (!probs || !labels || !in_len || !lab_len || !loss || !dloss || !workspace || !dprobs || Smax > CTC_SMAX ||
        workspace_bytes < sar_ctc_workspace_bytes(B, L, max_label_len) || V > 8192)
        return SAR_EINVAL;
    const float* alpha = static_cast<const float*>(workspace);
    const float* logden = alpha + (size_t)B * L * Smax;
    hipLaunchKernelGGL(ctc_bwd_kernel, dim3(B), dim3(256), sizeof(float) * V, stream, probs, labels, in_len, lab_len, loss, dloss, alpha, logden, dprobs, L, V, max_label_len, Smax);
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}

extern "C" int sar_ctc_greedy_decode_f32(const float* probs, const int* in_len, int* decoded, int* decoded_len, int B,
                                         int L, int V, hipStream_t stream) {
    if (!probs || !in_len || !decoded || !decoded_len) return SAR_EINVAL;
    hipLaunchKernelGGL(ctc_greedy_kernel, dim3(B), dim3(64), 0, stream, probs, in_len, decoded, decoded_len, L, V);
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}
