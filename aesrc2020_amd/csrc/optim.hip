// Fused L2-regularisation + Keras-Adam over the flat parameter buffer (model.py:197 Adam(lr, decay=2e-4);
// l2(1e-4) regularisers at model.py:35-48,87-104 and resnet.py:36,56,87,117; unit_norm at model.py:163).
// HBM-bound sweep: 16 B/param read + 12 B/param written.  Parameters are padded to 256-element chunks so a
// per-chunk table carries the L2 coefficient; step-dependent scalars live in a device-side state block so
// the launch can be replayed from a HIP graph.
#include "common.h"

#ifndef SAR_ADAM_WT
#define SAR_ADAM_WT 0
#endif

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

// FUSED (sar_adam_l2_update_f32, round 3): the step-size bookkeeping of adam_prepare_kernel rides in this launch -- every block derives
// lr_t from the iteration count it finds, and the LAST block to finish (a ticket: every block draws one at its end, so whoever draws the
// last knows that all blocks have read the count) advances it.  A one-thread kernel costs 4.6 us of the step's timeline here: kernel
// arguments, its load and the write-back of its store are three dependent trips behind the fabric, whatever it computes.
template <bool FUSED>
__global__ __launch_bounds__(256) void adam_l2_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                      float* __restrict__ m, float* __restrict__ v,
                                                      const float* __restrict__ chunk_l2, SarAdamState* st, unsigned* ticket,
                                                      float* __restrict__ l2_loss, long nchunks) {
    __shared__ float red[16];
    const float b1 = st->beta1, b2 = st->beta2, eps = st->eps, gs = st->grad_scale;
    float lr_t = st->lr_t;
    if constexpr (FUSED) {
        const float it = (float)st->iterations, lr = st->lr / (1.f + st->decay * it), t = it + 1.f;
        lr_t = lr * sqrtf(1.f - powf(b2, t)) / (1.f - powf(b1, t));
    }
    float lacc = 0.f;
    // Loss-scaling guard (ticket[1] != 0: the modes with fp16 operands -- the fp16 trunk, and the split-precision mode's fp16 weight-gradient
    // operands and GRU gradients).  Those carry gradients times a static power-of-two scale; a value past 65504 / scale is inf there and NaN in
    // the parameter gradients it feeds.  sar_grads_check_f32, launched in front of this kernel, has set ticket[3] when ANY gradient element
    // is not finite: the WHOLE update is then skipped -- weights, m, v and the iteration count stay as they are (round 3 dropped only the bad
    // elements and applied the rest: earlier layers got a zero-gradient update, later ones a real one) -- and ticket[2] counts the
    // skipped step.  The host reads that counter with its logs, warns, halves the scale and re-captures the step (model.SARModel).  The
    // modes without fp16 operands run with the guard off: a non-finite gradient there is a bug to be seen, not an overflow to be survived.
    const bool skip = FUSED && ticket[1] != 0u && ticket[3] != 0u;
    // one wave per 256-element chunk, float4 per lane (16-byte accesses: the scalar version ran at 3.4 TB/s), two chunks per
    // trip so that 14 loads are in flight per lane
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    auto upd = [&](float w, float gi, float& mi, float& vi, float l2) {
        const float gr = fmaf(2.f * l2, w, gi * gs);
        lacc = fmaf(l2 * w, w, lacc);
        mi = b1 * mi + (1.f - b1) * gr;
        vi = b2 * vi + (1.f - b2) * gr * gr;
        return w - lr_t * mi / (sqrtf(vi) + eps);
    };
    auto one = [&](long c, float4 w, float4 gg, float4 mm, float4 vv, float l2) {
        w.x = upd(w.x, gg.x, mm.x, vv.x, l2); w.y = upd(w.y, gg.y, mm.y, vv.y, l2);
        w.z = upd(w.z, gg.z, mm.z, vv.z, l2); w.w = upd(w.w, gg.w, mm.w, vv.w, l2);
        if (skip) return;                                           // (block-uniform: the l2 sum above is of the unchanged weights)
        const long i = c * 256 + lane * 4;
#if SAR_ADAM_WT                                                      // write-through (sc1) stores of p, m, v
        // SAR_ADAM_WT=1 (tools/sc1_adam_repro.py only, never in libsarnet_hip.so): round 5's form, a bare store in an asm statement -- wrong results, the
        // "write-through anomaly" of DESIGN 5.000; 2: the same with the wait states a >64-bit VMEM store needs in front of a VALU write of its data
        // registers -- hipcc's hazard recognizer inserts them for its own stores and cannot see into an asm statement
#if SAR_ADAM_WT >= 2
#define WT_TAIL "\n\ts_nop 1"
#else
#define WT_TAIL ""
#endif
        asm volatile("global_store_dwordx4 %0, %1, off sc1" WT_TAIL :: "v"(m + i), "v"(f32x4{mm.x, mm.y, mm.z, mm.w}) : "memory");
        asm volatile("global_store_dwordx4 %0, %1, off sc1" WT_TAIL :: "v"(v + i), "v"(f32x4{vv.x, vv.y, vv.z, vv.w}) : "memory");
        asm volatile("global_store_dwordx4 %0, %1, off sc1" WT_TAIL :: "v"(p + i), "v"(f32x4{w.x, w.y, w.z, w.w}) : "memory");
#undef WT_TAIL
#else
        *reinterpret_cast<float4*>(m + i) = mm; *reinterpret_cast<float4*>(v + i) = vv; *reinterpret_cast<float4*>(p + i) = w;
#endif
    };
    const long stride = (long)gridDim.x * 4;
    long c = (long)blockIdx.x * 4 + wv;
    for (; c + stride < nchunks; c += 2 * stride) {
        const long i0 = c * 256 + lane * 4, i1 = (c + stride) * 256 + lane * 4;
        const float4 w0 = *reinterpret_cast<const float4*>(p + i0), g0 = *reinterpret_cast<const float4*>(g + i0);
        const float4 m0 = *reinterpret_cast<const float4*>(m + i0), v0 = *reinterpret_cast<const float4*>(v + i0);
        const float4 w1 = *reinterpret_cast<const float4*>(p + i1), g1 = *reinterpret_cast<const float4*>(g + i1);
        const float4 m1 = *reinterpret_cast<const float4*>(m + i1), v1 = *reinterpret_cast<const float4*>(v + i1);
        const float l20 = chunk_l2[c], l21 = chunk_l2[c + stride];
        one(c, w0, g0, m0, v0, l20); one(c + stride, w1, g1, m1, v1, l21);
    }
    for (; c < nchunks; c += stride) {
        const long i0 = c * 256 + lane * 4;
        one(c, *reinterpret_cast<const float4*>(p + i0), *reinterpret_cast<const float4*>(g + i0), *reinterpret_cast<const float4*>(m + i0),
            *reinterpret_cast<const float4*>(v + i0), chunk_l2[c]);
    }
    if (l2_loss) {
        lacc = block_sum(lacc, red);
        if (threadIdx.x == 0) atomicAdd(l2_loss, lacc);
    }
    if constexpr (FUSED) {
        __syncthreads();                                            // every thread of this block has read the state and the skip flag (ADVICE r3:
                                                                    // without l2_loss nothing else orders those reads in front of the ticket)
        if (threadIdx.x == 0 && atomicAdd(ticket, 1u) == gridDim.x - 1) {   // every block has read the iteration count
            *ticket = 0u;
            if (skip) { ticket[2] += 1u; ticket[3] = 0u; ticket[4] += 1u; }   // a skipped step: nothing advances
            else { st->lr_t = lr_t; st->iterations += 1; ticket[4] = 0u; }   // [4]: skipped steps IN A ROW (an applied step resets it)
        }
    }
}

// ticket[3] = 1 when any of the n gradient elements is not finite (see adam_l2_kernel: the loss-scaling guard).  One read of the flat
// gradient buffer (34 MB at F = 32: ~10 us), only in the modes with fp16 operands.
__global__ __launch_bounds__(256) void grads_check_kernel(const float4* __restrict__ g, long n4, unsigned* __restrict__ ticket) {
    bool bad = false;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 a = g[i];
        bad |= !(fabsf(a.x) <= 3.0e38f) | !(fabsf(a.y) <= 3.0e38f) | !(fabsf(a.z) <= 3.0e38f) | !(fabsf(a.w) <= 3.0e38f);
    }
    if (bad) ticket[3] = 1u;
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
    const long nb4 = (nchunks + 3) / 4;                                // a block = 4 waves = 4 chunks per trip
    const int nb = (int)(nb4 < 2048 ? nb4 : 2048);
    hipLaunchKernelGGL(adam_l2_kernel<false>, dim3(nb), dim3(256), 0, stream, params, grads, m, v, chunk_l2,
                       static_cast<SarAdamState*>(const_cast<void*>(state)), static_cast<unsigned*>(nullptr), l2_loss, nchunks);
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}

// Loss-scaling guard, first half: sets ticket[3] when any element of grads is not finite; sar_adam_l2_update_f32 (ticket[1] != 0) then skips the step.
extern "C" int sar_grads_check_f32(const float* grads, void* ticket, long n, hipStream_t stream) {
    if (!grads || !ticket || n < 4 || (n % 4)) return SAR_EINVAL;
    const long n4 = n / 4;
    const long nb = (n4 + 256 * 8 - 1) / (256 * 8);
    hipLaunchKernelGGL(grads_check_kernel, dim3((unsigned)(nb < 2048 ? nb : 2048)), dim3(256), 0, stream, reinterpret_cast<const float4*>(grads), n4,
                       static_cast<unsigned*>(ticket));
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}

// sar_adam_prepare + sar_adam_l2_step_f32 in ONE launch.  `ticket`: EIGHT 32-bit words the caller owns, zero-initialised: [0] the block ticket
// (zero again when the launch has finished), [1] the caller's switch "loss-scaling guard on" (modes with fp16 operands), [2] how many steps
// were skipped so far because a gradient was not finite, [3] the flag sar_grads_check_f32 sets (cleared here), [4] how many steps IN A ROW
// were skipped (reset by every applied step: what "the training has diverged" is decided on -- a host that looks once per epoch cannot tell
// consecutive from scattered skips by the total), [5..7] spare.
extern "C" int sar_adam_l2_update_f32(float* params, const float* grads, float* m, float* v, const float* chunk_l2,
                                      void* state, void* ticket, float* l2_loss, long n, hipStream_t stream) {
    if (!params || !grads || !m || !v || !chunk_l2 || !state || !ticket || (n % 256)) return SAR_EINVAL;
    const long nchunks = n / 256;
    const long nb4 = (nchunks + 3) / 4;
    const int nb = (int)(nb4 < 2048 ? (nb4 > 0 ? nb4 : 1) : 2048);
    hipLaunchKernelGGL(adam_l2_kernel<true>, dim3(nb), dim3(256), 0, stream, params, grads, m, v, chunk_l2,
                       static_cast<SarAdamState*>(state), static_cast<unsigned*>(ticket), l2_loss, nchunks);
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
