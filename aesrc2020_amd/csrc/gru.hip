// Bidirectional reset-after GRU (Keras CuDNNGRU semantics, model.py:44-50) on gfx950.
// The input projection xa = x*W + b_i for all time steps is one MFMA GEMM outside this file (gemm.hip);
// here: the sequential recurrence, one persistent workgroup per (direction, group of GR batch rows),
// lane owns 4 hidden units, wave owns a k-slice (v3, see below). h lives in LDS; the recurrent kernel U (H,3H) streams from L2 every step
// (coalesced across j).  Gate order z, r, h; bias_h = bias[3H:6H].
//   z = sig(xa_z + hU_z + b_hz); r = sig(xa_r + hU_r + b_hr); q = hU_h + b_hh
//   hh = tanh(xa_h + r*q);  h' = z*h + (1-z)*hh
// Saved for BPTT: gates[d][b][t][{z,r,hh,q}][j] and hprev[d][b][t][j].
#include "common.h"

namespace {

constexpr int GR = 2;    // batch rows per workgroup
constexpr int NW = 8;    // waves per workgroup = k-slices of the reduction (512 threads: 256 VGPRs per thread)
// v3 mapping (H % 4 == 0, H <= 256): lane owns 4 consecutive hidden units (float4 loads of U rows: 1 KB per wave
// instruction instead of 256 B), wave w owns the k-slice [w*H/16, (w+1)*H/16).  The recurrence is bound by how fast one
// CU can stream the 3*H*H recurrent kernel from L2 every step, so bytes per load instruction and loads in flight matter,
// not FLOPs.  Partial sums go through LDS: part[NW][GR*12][64] -> summed by the gate thread of each unit -> hs[GR][H].
// v4: one third of that stream is taken out of the loop -- every thread keeps the candidate-gate rows of its k-slice
// (forward: U[k][2H + 4*lane ..], backward: the first KRES rows of its UT slice) in registers for all L steps: KRES float4 =
// 128 of the 256 VGPRs a 512-thread workgroup allows, 256 KB of the 786 KB per CU (with 1024 threads and a 128-register
// budget the compiler spilled).  The operands of the gate math that do not depend on the recurrence (xa; gates / hprev /
// dout in BPTT) are fetched ahead of the streaming phase instead of after it.
constexpr int KRES = 32;  // resident rows per thread (= H/NW at H = 256)
constexpr int XRES = 12;  // further resident rows (forward: of the reset gate; backward: rows KRES.. of the slice): what is left of 256 VGPRs

__device__ __forceinline__ void fma4(float4& a, float h, const float4& u) {
    a.x = fmaf(h, u.x, a.x); a.y = fmaf(h, u.y, a.y); a.z = fmaf(h, u.z, a.z); a.w = fmaf(h, u.w, a.w);
}

// dynamic LDS: hs[GR][H] | part[NW][GR*12][64]
template <bool RES>                                                // RES: H == KRES * NW, the k-slices are exactly KRES rows
__global__ __launch_bounds__(512) void gru_fwd_kernel(const float* __restrict__ xa_f, const float* __restrict__ xa_b,
                               const float* __restrict__ U_f, const float* __restrict__ U_b,
                               const float* __restrict__ bh_f, const float* __restrict__ bh_b,
                               float* __restrict__ out, float* __restrict__ gates, float* __restrict__ hprev, int B,
                               int L, int H) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* hs = sm; float* part = hs + GR * H;
    const int d = blockIdx.y, tid = threadIdx.x, lane = tid & 63, b0 = blockIdx.x * GR;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);         // wave-uniform: row addresses of the U stream stay in SGPRs
    const float* xa = d ? xa_b : xa_f; const float* U = d ? U_b : U_f; const float* bh = d ? bh_b : bh_f;
    const int H3 = 3 * H, H4 = H / 4;
    const int kper = (H + NW - 1) / NW, k0 = w * kper, k1 = min(H, k0 + kper);
    const bool lane_ok = lane < H4;
    for (int i = tid; i < GR * H; i += blockDim.x) hs[i] = 0.f;
    float4 uq[RES ? KRES : 1], ux[RES ? XRES : 1];                    // resident candidate-gate rows (+ XRES reset-gate rows)
    if constexpr (RES) {
#pragma unroll
        for (int kk = 0; kk < KRES; ++kk) uq[kk] = *reinterpret_cast<const float4*>(U + (long)(k0 + kk) * H3 + 2 * H + lane * 4);
#pragma unroll
        for (int kk = 0; kk < XRES; ++kk) ux[kk] = *reinterpret_cast<const float4*>(U + (long)(k0 + kk) * H3 + H + lane * 4);
    }
    __syncthreads();
    float* g_d = gates + (long)d * B * L * 4 * H;
    float* hp_d = hprev + (long)d * B * L * H;
    // gate-math ownership: thread tid < GR*H handles (row gr, unit gj) -- one output per thread (GR*H <= blockDim)
    const int gr = tid / H, gj = tid - gr * H, gb = b0 + gr;
    const bool gate_thread = tid < GR * H && gb < B;
    // recurrent bias of this thread's unit: loaded once, not in every step's critical path
    float bz = 0.f, br = 0.f, bq = 0.f;
    if (gate_thread) { bz = bh[gj]; br = bh[H + gj]; bq = bh[2 * H + gj]; }
    for (int s = 0; s < L; ++s) {
        const int t = d ? (L - 1 - s) : s;
        float xz = 0.f, xr = 0.f, xh = 0.f;                           // input projections of this step: in flight under the stream
        if (gate_thread) {
            const long xo = ((long)gb * L + t) * H3;
            xz = xa[xo + gj]; xr = xa[xo + H + gj]; xh = xa[xo + 2 * H + gj];
        }
        float4 az[GR], ar[GR], aq[GR];
#pragma unroll
        for (int r = 0; r < GR; ++r) { az[r] = make_float4(0, 0, 0, 0); ar[r] = az[r]; aq[r] = az[r]; }
        if constexpr (RES) {                                         // H == 256: every lane owns 4 units, every slice is KRES rows
            int opq = 0;
            asm volatile("" : "+s"(opq));                             // opaque 0: keeps the 64 row addresses of the unrolled loop from
            const float* Us = U + opq;                                // being hoisted out of the time loop as 128 live (spilled) VGPRs
#pragma unroll
            for (int kk = 0; kk < KRES; ++kk) {                       // fully unrolled (uq[] must stay in registers) ...
                if (kk % 4 == 0) __builtin_amdgcn_sched_barrier(0);   // ... but loads are batched four rows at a time, not all 64
                const int k = k0 + kk;
                const float* ur = Us + (long)k * H3 + lane * 4;
                const float4 u0 = *reinterpret_cast<const float4*>(ur);
                float4 u1;
                if (kk < XRES) u1 = ux[kk < XRES ? kk : 0]; else u1 = *reinterpret_cast<const float4*>(ur + H);
#pragma unroll
                for (int r = 0; r < GR; ++r) { const float hv = hs[r * H + k]; fma4(az[r], hv, u0); fma4(ar[r], hv, u1); fma4(aq[r], hv, uq[kk]); }
            }
        } else if (lane_ok) {
#pragma unroll 4
            for (int k = k0; k < k1; ++k) {
                const float* ur = U + (long)k * H3 + lane * 4;
                const float4 u0 = *reinterpret_cast<const float4*>(ur), u1 = *reinterpret_cast<const float4*>(ur + H),
                             u2 = *reinterpret_cast<const float4*>(ur + 2 * H);
#pragma unroll
                for (int r = 0; r < GR; ++r) { const float hv = hs[r * H + k]; fma4(az[r], hv, u0); fma4(ar[r], hv, u1); fma4(aq[r], hv, u2); }
            }
        }
#pragma unroll
        for (int r = 0; r < GR; ++r) {
            float* p = part + ((w * GR + r) * 12) * 64 + lane;
            p[0 * 64] = az[r].x; p[1 * 64] = az[r].y; p[2 * 64] = az[r].z; p[3 * 64] = az[r].w;
            p[4 * 64] = ar[r].x; p[5 * 64] = ar[r].y; p[6 * 64] = ar[r].z; p[7 * 64] = ar[r].w;
            p[8 * 64] = aq[r].x; p[9 * 64] = aq[r].y; p[10 * 64] = aq[r].z; p[11 * 64] = aq[r].w;
        }
        __syncthreads();
        // cross-wave reduction: output o = (r, g, j); j = 4*l4 + e lives at part[*][r*12 + g*4 + e][l4]
        // the gate thread of (row, unit) sums the NW partials of its three pre-activations itself: no separate reduction phase,
        // one barrier and one LDS round trip fewer per step.  Unit j = 4*l4 + e lives at part[q][r*12 + g*4 + e][l4].
        if (gate_thread) {
            const int r = gr, jj = gj, o = tid, b = gb;
            const float* p = part + (r * 12 + (jj & 3)) * 64 + (jj >> 2);
            float sz = 0.f, sr = 0.f, sq = 0.f;
#pragma unroll
            for (int q = 0; q < NW; ++q) { sz += p[q * GR * 12 * 64]; sr += p[q * GR * 12 * 64 + 4 * 64]; sq += p[q * GR * 12 * 64 + 8 * 64]; }
            const float hold = hs[o];
            const float q = sq + bq;
            const float z = sigmoidf_(xz + (sz + bz));
            const float rr = sigmoidf_(xr + (sr + br));
            const float hh = tanhf(xh + rr * q);
            const float hn = z * hold + (1.f - z) * hh;
            const long go = (((long)b * L + t) * 4) * H;
            g_d[go + jj] = z; g_d[go + H + jj] = rr; g_d[go + 2 * H + jj] = hh; g_d[go + 3 * H + jj] = q;
            hp_d[((long)b * L + t) * H + jj] = hold;
            out[((long)b * L + t) * 2 * H + d * H + jj] = hn;
            hs[o] = hn;
        }
        __syncthreads();
    }
}

// BPTT.  UT = transpose(U): (3H, H).  dai/dah: (2, B, L, 3H) input-side / recurrent-side pre-activation grads.
// dynamic LDS: ds[GR][3H] | part[NW][GR][H]
template <bool RES>                                                // RES: H == 256, UT slices of 3 * KRES rows
__global__ __launch_bounds__(512) void gru_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ gates,
                               const float* __restrict__ hprev, const float* __restrict__ UT_f,
                               const float* __restrict__ UT_b, float* __restrict__ dai, float* __restrict__ dah, int B,
                               int L, int H) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int H3 = 3 * H, H4 = H / 4;
    float* ds = sm; float* part = ds + GR * H3;
    const int d = blockIdx.y, tid = threadIdx.x, lane = tid & 63, b0 = blockIdx.x * GR;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);         // wave-uniform: row addresses of the U stream stay in SGPRs
    const float* UT = d ? UT_b : UT_f;
    const int cper = (H3 + NW - 1) / NW, c0 = w * cper, c1 = min(H3, c0 + cper);
    const bool lane_ok = lane < H4;
    const float* g_d = gates + (long)d * B * L * 4 * H;
    const float* hp_d = hprev + (long)d * B * L * H;
    float* dai_d = dai + (long)d * B * L * H3; float* dah_d = dah + (long)d * B * L * H3;
    constexpr int BRES = KRES + XRES;
    float4 ures[RES ? BRES : 1];                                      // resident: the first BRES rows of this wave's UT slice
    if constexpr (RES) {
#pragma unroll
        for (int cc = 0; cc < BRES; ++cc) ures[cc] = *reinterpret_cast<const float4*>(UT + (long)(c0 + cc) * H + lane * 4);
    }
    __syncthreads();
    // gate backward ownership: thread tid < GR*H handles (row gr, unit gj); its saved activations for step s-1 are fetched
    // while step s streams UT
    const int gr = tid / H, gj = tid - gr * H, gb = b0 + gr;
    const bool gate_thread = tid < GR * H && gb < B;
    struct Saved { float z, rr, hh, q, hp, dy; };
    auto fetch = [&](int s) {
        Saved v{0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (gate_thread && s >= 0) {
            const int t = d ? (L - 1 - s) : s;
            const long go = (((long)gb * L + t) * 4) * H;
            v.z = g_d[go + gj]; v.rr = g_d[go + H + gj]; v.hh = g_d[go + 2 * H + gj]; v.q = g_d[go + 3 * H + gj];
            v.hp = hp_d[((long)gb * L + t) * H + gj];
            v.dy = dout[((long)gb * L + t) * 2 * H + d * H + gj];
        }
        return v;
    };
    Saved nxt = fetch(L - 1);
    float dh = 0.f;                                                   // dL/dh carried to the previous step, of this thread's (row, unit)
    for (int s = L - 1; s >= 0; --s) {
        const int t = d ? (L - 1 - s) : s;
        const Saved cur = nxt;
        if (tid < GR * H) {                                          // gate backward for (row r, unit jj)
            const int r = gr, jj = gj, b = gb;
            float a0 = 0.f, a1 = 0.f, a2 = 0.f;
            if (b < B) {
                const float z = cur.z;
                const float rr = cur.rr, hh = cur.hh, q = cur.q, hp = cur.hp;
                const float g = dh + cur.dy;
                dh = g * z;                                           // the direct path; the recurrent-kernel path is added below
                const float da_h = g * (1.f - z) * (1.f - hh * hh);
                const float da_z = g * (hp - hh) * z * (1.f - z);
                const float da_r = da_h * q * rr * (1.f - rr);
                const long oo = ((long)b * L + t) * H3;
                dai_d[oo + jj] = da_z; dai_d[oo + H + jj] = da_r; dai_d[oo + 2 * H + jj] = da_h;
                a0 = da_z; a1 = da_r; a2 = da_h * rr;
                dah_d[oo + jj] = a0; dah_d[oo + H + jj] = a1; dah_d[oo + 2 * H + jj] = a2;
            }
            ds[r * H3 + jj] = a0; ds[r * H3 + H + jj] = a1; ds[r * H3 + 2 * H + jj] = a2;
        }
        __syncthreads();
        nxt = fetch(s - 1);
        float4 acc[GR];
#pragma unroll
        for (int r = 0; r < GR; ++r) acc[r] = make_float4(0, 0, 0, 0);
        if (lane_ok) {
#pragma unroll 4
            for (int c = c0 + (RES ? BRES : 0); c < c1; ++c) {        // streamed rows
                const float4 u = *reinterpret_cast<const float4*>(UT + (long)c * H + lane * 4);
#pragma unroll
                for (int r = 0; r < GR; ++r) fma4(acc[r], ds[r * H3 + c], u);
            }
            if constexpr (RES) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int cc = 0; cc < BRES; ++cc) {                   // resident rows
#pragma unroll
                    for (int r = 0; r < GR; ++r) fma4(acc[r], ds[r * H3 + c0 + cc], ures[cc]);
                }
            }
        }
        // each wave's partial sums as float4 per lane (unit j at float index j: conflict-free 16-byte stores, consecutive-lane reads)
        if (lane_ok) {
#pragma unroll
            for (int r = 0; r < GR; ++r) *reinterpret_cast<float4*>(part + (w * GR + r) * H + lane * 4) = acc[r];
        }
        __syncthreads();
        if (tid < GR * H) {                                          // same thread as the gate phase: dh stays in a register, and the
            float a = 0.f;                                            // next gate phase needs no barrier in between
#pragma unroll
            for (int q = 0; q < NW; ++q) a += part[(q * GR + gr) * H + gj];
            dh += a;
        }
    }
}

// seq=False helper (model.py:128): y[b] = [out[b, L-1, :H], out[b, 0, H:]]
__global__ void bigru_last_fwd_kernel(const float* __restrict__ out, float* __restrict__ y, int B, int L, int H) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * 2 * H) return;
    const int b = (int)(i / (2 * H)), c = (int)(i % (2 * H));
    const int t = c < H ? L - 1 : 0;
    y[i] = out[((long)b * L + t) * 2 * H + c];
}
__global__ void bigru_last_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dout, int B, int L, int H) {
    const long n = (long)B * L * 2 * H;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % (2 * H)); const long bt = i / (2 * H);
        const int t = (int)(bt % L), b = (int)(bt / L);
        const int tt = c < H ? L - 1 : 0;
        dout[i] = (t == tt) ? dy[(long)b * 2 * H + c] : 0.f;
    }
}

// ---- any hidden_dim (round 4): ONE TIME STEP of both directions.  The persistent kernels above end at 256 units (a lane owns four of
// 64 x 4; at 512 units the recurrent kernels are 3 MB per direction and no CU holds or streams them in a step's time).  Larger layers run
// the recurrence as L x (one batched GEMM launch for ha = h U of both directions + this gate kernel) -- launch-bound (two launches per
// step), but the reference takes any hidden_dim (model.py:204-224) and so does this.  Same saved tensors and layouts as gru_fwd_kernel.
__global__ __launch_bounds__(256) void gru_step_fwd_kernel(const float* __restrict__ xa_f, const float* __restrict__ xa_b, const float* __restrict__ ha,
                                                           const float* __restrict__ bh_f, const float* __restrict__ bh_b, float* __restrict__ h,
                                                           float* __restrict__ out, float* __restrict__ gates, float* __restrict__ hprev, int B, int L,
                                                           int H, int s) {
    const long n = 2L * B * H;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int j = (int)(i % H), b = (int)(i / H % B), d = (int)(i / ((long)H * B));
        const int t = d ? (L - 1 - s) : s;
        const float* xa = d ? xa_b : xa_f; const float* bh = d ? bh_b : bh_f;
        const long xo = ((long)b * L + t) * 3 * H, ho = ((long)d * B + b) * 3 * H;
        const float z = sigmoidf_(xa[xo + j] + (ha[ho + j] + bh[j]));
        const float rr = sigmoidf_(xa[xo + H + j] + (ha[ho + H + j] + bh[H + j]));
        const float q = ha[ho + 2 * H + j] + bh[2 * H + j];
        const float hh = tanhf(xa[xo + 2 * H + j] + rr * q);
        const float hold = h[i];
        const float hn = z * hold + (1.f - z) * hh;
        float* g = gates + ((((long)d * B + b) * L + t) * 4) * H;
        g[j] = z; g[H + j] = rr; g[2 * H + j] = hh; g[3 * H + j] = q;
        hprev[(((long)d * B + b) * L + t) * H + j] = hold;
        out[((long)b * L + t) * 2 * H + d * H + j] = hn;
        h[i] = hn;
    }
}
// BPTT step s (the caller walks s = L-1 .. 0): dh holds the gradient flowing into h_s from step s+1; on return dh = its direct path into
// h_{s-1} (g * z) and dstep = this step's recurrent-side pre-activation gradients, which the caller multiplies by U^T and adds to dh.
__global__ __launch_bounds__(256) void gru_step_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ gates, const float* __restrict__ hprev,
                                                           float* __restrict__ dh, float* __restrict__ dai, float* __restrict__ dah, float* __restrict__ dstep,
                                                           int B, int L, int H, int s) {
    const long n = 2L * B * H;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int j = (int)(i % H), b = (int)(i / H % B), d = (int)(i / ((long)H * B));
        const int t = d ? (L - 1 - s) : s;
        const float* g4 = gates + ((((long)d * B + b) * L + t) * 4) * H;
        const float z = g4[j], rr = g4[H + j], hh = g4[2 * H + j], q = g4[3 * H + j];
        const float hp = hprev[(((long)d * B + b) * L + t) * H + j];
        const float g = dh[i] + dout[((long)b * L + t) * 2 * H + d * H + j];
        dh[i] = g * z;
        const float dahh = g * (1.f - z) * (1.f - hh * hh);
        const float daz = g * (hp - hh) * z * (1.f - z);
        const float dar = dahh * q * rr * (1.f - rr);
        const float a2 = dahh * rr;
        const long oo = (((long)d * B + b) * L + t) * 3 * H, so = ((long)d * B + b) * 3 * H;
        dai[oo + j] = daz; dai[oo + H + j] = dar; dai[oo + 2 * H + j] = dahh;
        dah[oo + j] = daz; dah[oo + H + j] = dar; dah[oo + 2 * H + j] = a2;
        dstep[so + j] = daz; dstep[so + H + j] = dar; dstep[so + 2 * H + j] = a2;
    }
}

}  // namespace

// One time step s of the recurrence of BOTH directions for any hidden_dim (see gru_step_fwd_kernel): ha (2, B, 3H) = h U computed by the caller
// (sar_gemm_batched_f32), h (2, B, H) the running state (zero in front of step 0), the other tensors as sar_bigru_fwd_f32.
extern "C" int sar_bigru_step_fwd_f32(const float* xa_fwd, const float* xa_bwd, const float* ha, const float* bh_fwd, const float* bh_bwd, float* h,
                                      float* out, float* gates, float* hprev, int B, int L, int H, int s, hipStream_t stream) {
    if (!xa_fwd || !xa_bwd || !ha || !bh_fwd || !bh_bwd || !h || !out || !gates || !hprev || B < 1 || L < 1 || H < 1 || s < 0 || s >= L) return SAR_EINVAL;
    long nb = (2L * B * H + 255) / 256; if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(gru_step_fwd_kernel, dim3((unsigned)nb), dim3(256), 0, stream, xa_fwd, xa_bwd, ha, bh_fwd, bh_bwd, h, out, gates, hprev, B, L, H, s);
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}
// BPTT step s (s = L-1 .. 0): dh (2, B, H) in / out, dstep (2, B, 3H) out; the caller then adds dstep U^T to dh (sar_gemm_batched_f32, accumulate).
extern "C" int sar_bigru_step_bwd_f32(const float* dout, const float* gates, const float* hprev, float* dh, float* dai, float* dah, float* dstep,
                                      int B, int L, int H, int s, hipStream_t stream) {
    if (!dout || !gates || !hprev || !dh || !dai || !dah || !dstep || B < 1 || L < 1 || H < 1 || s < 0 || s >= L) return SAR_EINVAL;
    long nb = (2L * B * H + 255) / 256; if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(gru_step_bwd_kernel, dim3((unsigned)nb), dim3(256), 0, stream, dout, gates, hprev, dh, dai, dah, dstep, B, L, H, s);
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}

extern "C" int sar_bigru_fwd_f32(const float* xa_fwd, const float* xa_bwd, const float* U_fwd, const float* U_bwd,
                                 const float* bh_fwd, const float* bh_bwd, float* out, float* gates, float* hprev, int B,
                                 int L, int H, hipStream_t stream) {
    if (!xa_fwd || !xa_bwd || !U_fwd || !U_bwd || !bh_fwd || !bh_bwd || !out || !gates || !hprev || H < 1 || H > 1024)
        return SAR_EINVAL;
    if (H > 256 || H % 4) return SAR_EINVAL;
    const size_t lds = sizeof(float) * (GR * H + NW * GR * 12 * 64);
    static PerDeviceFlag attr_f;
    if (!attr_f) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(gru_fwd_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(gru_fwd_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return SAR_ELAUNCH;
        attr_f = true;
    }
    if (H == KRES * NW)
        hipLaunchKernelGGL(gru_fwd_kernel<true>, dim3(cdiv(B, GR), 2), dim3(64 * NW), lds, stream,
                           xa_fwd, xa_bwd, U_fwd, U_bwd, bh_fwd, bh_bwd, out, gates, hprev, B, L, H);
    else
        hipLaunchKernelGGL(gru_fwd_kernel<false>, dim3(cdiv(B, GR), 2), dim3(64 * NW), lds, stream,
                           xa_fwd, xa_bwd, U_fwd, U_bwd, bh_fwd, bh_bwd, out, gates, hprev, B, L, H);
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}

extern "C" int sar_bigru_bwd_f32(const float* dout, const float* gates, const float* hprev, const float* UT_fwd,
                                 const float* UT_bwd, float* dai, float* dah, int B, int L, int H, hipStream_t stream) {
    if (!dout || !gates || !hprev || !UT_fwd || !UT_bwd || !dai || !dah || H < 1 || H > 1024) return SAR_EINVAL;
    if (H > 256 || H % 4) return SAR_EINVAL;
    const size_t lds = sizeof(float) * (GR * 3 * H + NW * GR * H);
    if (H == KRES * NW)
        hipLaunchKernelGGL(gru_bwd_kernel<true>, dim3(cdiv(B, GR), 2), dim3(64 * NW), lds, stream,
                           dout, gates, hprev, UT_fwd, UT_bwd, dai, dah, B, L, H);
    else
        hipLaunchKernelGGL(gru_bwd_kernel<false>, dim3(cdiv(B, GR), 2), dim3(64 * NW), lds, stream,
                           dout, gates, hprev, UT_fwd, UT_bwd, dai, dah, B, L, H);
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}

extern "C" int sar_bigru_last_fwd_f32(const float* out, float* y, int B, int L, int H, hipStream_t stream) {
    if (!out || !y) return SAR_EINVAL;
    hipLaunchKernelGGL(bigru_last_fwd_kernel, dim3(cdiv((long)B * 2 * H, 256)), dim3(256), 0, stream, out, y, B, L, H);
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}
extern "C" int sar_bigru_last_bwd_f32(const float* dy, float* dout, int B, int L, int H, hipStream_t stream) {
    if (!dy || !dout) return SAR_EINVAL;
    long nb = ((long)B * L * 2 * H + 255) / 256; if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(bigru_last_bwd_kernel, dim3((int)nb), dim3(256), 0, stream, dy, dout, B, L, H);
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}
