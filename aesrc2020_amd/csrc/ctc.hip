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
// dynamic LDS for the e / alpha staging.  Round 4: up to 144 KB (the static arrays are < 12 KB; past 64 KB the launchers raise the kernels'
// limit): the reference's shipped configuration -- 1200 frames, L = 114, 145 states: 66 KB forward, 132 KB backward -- ran the recursions on
// the L2-resident workspace, one round trip per frame in their critical path: ctc_fwd 103 -> 64 us, ctc_bwd 166 -> 91 us at B = 32.
constexpr size_t CTC_LDS_CAP = 144 * 1024;

// log(exp(a) + exp(b)) for the recursions: m + log(1 + exp(-|a - b|)) on v_exp_f32 / v_log_f32 (one of each; ~1e-7 relative in the
// probability domain).  common.h's lse2 is two libm expf and a logf, ~120 instructions -- with up to two of them per state and frame
// that arithmetic, not memory, was the recursion's 1.1 us per frame.
__device__ __forceinline__ float lse2f(float a, float b) {
    const float m = fmaxf(a, b);
    if (m == -INFINITY) return -INFINITY;
    return m + __logf(1.f + __expf(-fabsf(a - b)));
}

__device__ __forceinline__ int ext_label(const float* lab, int s, int blank) {
    return (s & 1) ? (int)lab[s >> 1] : blank;
}

// Both kernels split the work into a PARALLEL part (denominators, emission log-probabilities e[t][s] of the extended label,
// the dense part of the gradient) and the inherently SEQUENTIAL alpha / beta recursion.  The recursion used to issue a
// dependent global load of the 10 MB probability tensor and two block barriers per frame (~2 us x L of pure latency); now it
// reads e (and alpha, backward) from LDS when T*S fits (always for the reference shapes: 39 x 33), else from the
// L2-resident workspace.
constexpr int CTC_NT = 1024;       // threads per workgroup: one workgroup walks an utterance's L x V probabilities, so the parallel phases want the loads of many waves in flight
constexpr int CTC_NF = 4;           // frames a wave reduces at once (independent loads in flight)

template <bool LDS_E>
__global__ __launch_bounds__(CTC_NT) void ctc_fwd_kernel(const float* __restrict__ probs, const float* __restrict__ labels,
                                                      const int* __restrict__ in_len, const int* __restrict__ lab_len,
                                                      float* __restrict__ loss, float* __restrict__ alpha_ws,
                                                      float* __restrict__ logden_ws, float* __restrict__ e_ws, int L, int V,
                                                      int maxU, int Smax) {
    extern __shared__ float e_s[];               // LDS_E: [L][Smax]
    __shared__ float al[2][CTC_SMAX];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int T = in_len[b]; if (T > L) T = L;
    const int U = min(lab_len[b], maxU), S = 2 * U + 1, blank = V - 1;
    const float* p = probs + (long)b * L * V;
    const float* lab = labels + (long)b * maxU;
    // the extended label sequence l' (blank, l1, blank, l2, ...) once into LDS: the recursion below reads it every frame, and a
    // global (even L1-resident) load per frame sits in its critical path
    __shared__ int xl[CTC_SMAX];
    for (int s_ = threadIdx.x; s_ < 2 * min(lab_len[b], maxU) + 1; s_ += CTC_NT) xl[s_] = ext_label(lab, s_, V - 1);
    float* aw = alpha_ws + (long)b * L * Smax;
    float* ew = e_ws + (long)b * L * Smax;
    float* ld = logden_ws + (long)b * L;
    // (1) log denominators: each wave reduces CTC_NF frames at a time
    for (int t0 = wave * CTC_NF; t0 < T; t0 += (CTC_NT / 64) * CTC_NF) {
        float acc[CTC_NF];
#pragma unroll
        for (int j = 0; j < CTC_NF; ++j) acc[j] = 0.f;
        for (int v = lane; v < V; v += 64) {
#pragma unroll
            for (int j = 0; j < CTC_NF; ++j) acc[j] += p[(long)min(t0 + j, T - 1) * V + v] + CTC_EPS;
        }
#pragma unroll
        for (int j = 0; j < CTC_NF; ++j) {
            const float sum = wave_sum(acc[j]);
            if (lane == 0 && t0 + j < T) ld[t0 + j] = logf(sum);
        }
    }
    __syncthreads();
    // (2) emissions of the extended label: e[t][s] = log y_t(l'_s), all (t, s) in parallel
    for (int i0 = tid; i0 < T * S; i0 += 4 * CTC_NT) {                   // four gathers in flight per thread
        float pv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = min(i0 + CTC_NT * j, T * S - 1), t = i / S, s_ = i - t * S;
            pv[j] = p[(long)t * V + xl[s_]];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = i0 + CTC_NT * j;
            if (i < T * S) {
                const int t = i / S, s_ = i - t * S;
                const float e = logf(pv[j] + CTC_EPS) - ld[t];
                ew[(long)t * Smax + s_] = e;                          // backward reads it from the workspace
                if (LDS_E) e_s[t * Smax + s_] = e;
            }
        }
    }
    __syncthreads();
    // (3) alpha recursion: four waves are plenty for <= 145 states, and a barrier among four is cheaper than among sixteen -- the waves
    // that only served the parallel phases leave (a finished wave does not take part in later barriers)
    // (gfx9 ISA: s_barrier counts the waves of the workgroup that have not terminated; this is hardware behaviour, not HIP's programming
    // model -- the guard below keeps the file from being built for a target where it has not been checked, tests/test_gpu_ops.py::test_ctc
    // runs S > 256 and T = 1 through it)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "ctc.hip: early wave exit in front of workgroup barriers is verified on gfx950 only"
#endif
    if (tid >= 256) return;
    for (int s_ = tid; s_ < S; s_ += 256) {
        const float a = (s_ < 2 && T > 0) ? (LDS_E ? e_s[s_] : ew[s_]) : -INFINITY;
        al[0][s_] = a; aw[s_] = a;
    }
    __syncthreads();
    for (int t = 1; t < T; ++t) {
        const int cur = t & 1, prv = cur ^ 1;
        for (int s_ = tid; s_ < S; s_ += 256) {
            const int k = xl[s_];
            float a = al[prv][s_];
            if (s_ >= 1) a = lse2f(a, al[prv][s_ - 1]);
            if (s_ >= 2 && k != blank && k != xl[s_ - 2]) a = lse2f(a, al[prv][s_ - 2]);
            a += LDS_E ? e_s[t * Smax + s_] : ew[(long)t * Smax + s_];
            al[cur][s_] = a; aw[(long)t * Smax + s_] = a;
        }
        lds_barrier();                                                // the frames talk through al[][] only; the alpha stores stay in flight
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

// dp[t][v] = g * (1/den_t - occ_t(v) / (p_t(v) + eps)); occ_t(v) = sum over states s with l'_s = v of exp(alpha+beta-ll).
// Dense part g/den_t first (parallel), then per frame the few label symbols are corrected with atomics:
// occ_t(l'_s)/(p+eps) = exp(alpha_t(s) + beta_t(s) - ll - e_t(s) - log den_t)   [beta excludes the emission at t].
template <bool LDS_E>
__global__ __launch_bounds__(CTC_NT) void ctc_bwd_kernel(const float* __restrict__ probs, const float* __restrict__ labels,
                                                      const int* __restrict__ in_len, const int* __restrict__ lab_len,
                                                      const float* __restrict__ loss, const float* __restrict__ dloss,
                                                      const float* __restrict__ alpha_ws,
                                                      const float* __restrict__ logden_ws, const float* __restrict__ e_ws,
                                                      float* __restrict__ dprobs, int L, int V, int maxU, int Smax) {
    extern __shared__ float dyn[];               // LDS_E: e [L][Smax] | alpha - e - log den [L][Smax]
    __shared__ float be[2][CTC_SMAX];            // beta (excludes the emission at t)
    const int b = blockIdx.x, tid = threadIdx.x;
    int T = in_len[b]; if (T > L) T = L;
    const int U = min(lab_len[b], maxU), S = 2 * U + 1, blank = V - 1;
    const float* lab = labels + (long)b * maxU;
    // the extended label sequence l' (blank, l1, blank, l2, ...) once into LDS: the recursion below reads it every frame, and a
    // global (even L1-resident) load per frame sits in its critical path
    __shared__ int xl[CTC_SMAX];
    for (int s_ = threadIdx.x; s_ < 2 * min(lab_len[b], maxU) + 1; s_ += CTC_NT) xl[s_] = ext_label(lab, s_, V - 1);
    const float* aw = alpha_ws + (long)b * L * Smax;
    const float* ew = e_ws + (long)b * L * Smax;
    const float* ld = logden_ws + (long)b * L;
    float* dp = dprobs + (long)b * L * V;
    float* e_s = dyn; float* c_s = dyn + (LDS_E ? L * Smax : 0);
    const float ll = -loss[b];
    const float g = dloss[b];
    const bool feasible = (ll > -INFINITY) && (ll < INFINITY);
    // dense part; frames beyond the input length (and everything when infeasible) get zero gradient
    const int Tl = feasible ? T : 0;
    const int lane = tid & 63, wave = tid >> 6;
    const bool vec = (V & 3) == 0 && (reinterpret_cast<uintptr_t>(dp) & 15) == 0;
    for (int t = wave; t < L; t += CTC_NT / 64) {                     // one wave per frame row, 16-byte stores
        const float val = t < Tl ? g * expf(-ld[t]) : 0.f;
        float* row = dp + (long)t * V;
        if (vec) {
            const float4 v4 = make_float4(val, val, val, val);
            for (int v = lane * 4; v < V; v += 256) *reinterpret_cast<float4*>(row + v) = v4;
        } else {
            for (int v = lane; v < V; v += 64) row[v] = val;
        }
    }
    if (!feasible) return;
    if (LDS_E) {
        for (int i0 = tid; i0 < T * S; i0 += 4 * CTC_NT) {               // eight loads in flight per thread
            float ev[4], av[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = min(i0 + CTC_NT * j, T * S - 1), t = i / S, s_ = i - t * S;
                ev[j] = ew[(long)t * Smax + s_]; av[j] = aw[(long)t * Smax + s_];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = i0 + CTC_NT * j;
                if (i < T * S) {
                    const int t = i / S, s_ = i - t * S;
                    e_s[t * Smax + s_] = ev[j]; c_s[t * Smax + s_] = av[j] - ev[j] - ld[t] - ll;
                }
            }
        }
    }
    __syncthreads();                                                  // dense rows written before the atomics below
    if (tid >= 256) return;                                           // the recursion runs on four waves (see the forward kernel)
    for (int t = T - 1; t >= 0; --t) {
        const int cur = t & 1, nxt = cur ^ 1;
        for (int s_ = tid; s_ < S; s_ += 256) {
            float bb;
            if (t == T - 1) {
                bb = (s_ >= S - 2) ? 0.f : -INFINITY;
            } else {
                const int k0 = xl[s_];
                const float e0 = LDS_E ? e_s[(t + 1) * Smax + s_] : ew[(long)(t + 1) * Smax + s_];
                bb = be[nxt][s_] + e0;
                if (s_ + 1 < S) bb = lse2f(bb, be[nxt][s_ + 1] + (LDS_E ? e_s[(t + 1) * Smax + s_ + 1] : ew[(long)(t + 1) * Smax + s_ + 1]));
                if (s_ + 2 < S) {
                    const int k2 = xl[s_ + 2];
                    if (k2 != blank && k2 != k0) bb = lse2f(bb, be[nxt][s_ + 2] + (LDS_E ? e_s[(t + 1) * Smax + s_ + 2] : ew[(long)(t + 1) * Smax + s_ + 2]));
                }
            }
            be[cur][s_] = bb;
            const float c = LDS_E ? c_s[t * Smax + s_] : (aw[(long)t * Smax + s_] - ew[(long)t * Smax + s_] - ld[t] - ll);
            const float w = __expf(c + bb);
            if (w > 0.f) atomicAdd(&dp[(long)t * V + xl[s_]], -g * w);
        }
        lds_barrier();                                                // be[][] only; the atomics stay in flight
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
    return sizeof(float) * (2 * (size_t)B * L * Smax + (size_t)B * L);       // alpha | emissions e | log denominators
}

extern "C" int sar_ctc_batch_cost_fwd_f32(const float* probs, const float* labels, const int* in_len, const int* lab_len,
                                          float* loss, void* workspace, size_t workspace_bytes, int B, int L, int V,
                                          int max_label_len, hipStream_t stream) {
    const int Smax = 2 * max_label_len + 1;
    if (!probs || !labels || !in_len || !lab_len || !loss || !workspace || Smax > CTC_SMAX ||
        workspace_bytes < sar_ctc_workspace_bytes(B, L, max_label_len))
        return SAR_EINVAL;
    float* alpha = static_cast<float*>(workspace);
    float* ews = alpha + (size_t)B * L * Smax;
    float* logden = ews + (size_t)B * L * Smax;
    const size_t lds = sizeof(float) * (size_t)L * Smax;
    // LDS-resident recursion up to CTC_LDS_CAP; where the device refuses that much dynamic LDS (a partition with less than gfx950's 160 KB,
    // or static arrays that have grown) the shape falls back to the workspace kernel instead of failing
    static PerDeviceFlag attr, refused;
    if (lds > 48 * 1024 && lds <= CTC_LDS_CAP && !attr && !refused) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(ctc_fwd_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)CTC_LDS_CAP) != hipSuccess) {
            (void)hipGetLastError(); refused = true;
        } else attr = true;
    }
    if (lds <= 48 * 1024 || (lds <= CTC_LDS_CAP && attr))
        hipLaunchKernelGGL(ctc_fwd_kernel<true>, dim3(B), dim3(CTC_NT), lds, stream, probs, labels, in_len, lab_len, loss, alpha, logden, ews, L, V, max_label_len, Smax);
    else
        hipLaunchKernelGGL(ctc_fwd_kernel<false>, dim3(B), dim3(CTC_NT), 0, stream, probs, labels, in_len, lab_len, loss, alpha, logden, ews, L, V, max_label_len, Smax);
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
    const float* ews = alpha + (size_t)B * L * Smax;
    const float* logden = ews + (size_t)B * L * Smax;
    const size_t lds = 2 * sizeof(float) * (size_t)L * Smax;
    // LDS-resident recursion up to CTC_LDS_CAP; where the device refuses that much dynamic LDS (a partition with less than gfx950's 160 KB,
    // or static arrays that have grown) the shape falls back to the workspace kernel instead of failing
    static PerDeviceFlag attr, refused;
    if (lds > 48 * 1024 && lds <= CTC_LDS_CAP && !attr && !refused) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(ctc_bwd_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)CTC_LDS_CAP) != hipSuccess) {
            (void)hipGetLastError(); refused = true;
        } else attr = true;
    }
    if (lds <= 48 * 1024 || (lds <= CTC_LDS_CAP && attr))
        hipLaunchKernelGGL(ctc_bwd_kernel<true>, dim3(B), dim3(CTC_NT), lds, stream, probs, labels, in_len, lab_len, loss, dloss, alpha, logden, ews, dprobs, L, V, max_label_len, Smax);
    else
        hipLaunchKernelGGL(ctc_bwd_kernel<false>, dim3(B), dim3(CTC_NT), 0, stream, probs, labels, in_len, lab_len, loss, dloss, alpha, logden, ews, dprobs, L, V, max_label_len, Smax);
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
