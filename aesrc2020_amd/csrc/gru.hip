// Bidirectional reset-after GRU (Keras CuDNNGRU semantics, model.py:44-50) on gfx950.
// The input projection xa = x*W + b_i for all time steps is one MFMA GEMM outside this file (gemm.hip);
// here: the sequential recurrence, one persistent workgroup per (direction, group of GR batch rows),
// thread (j, kq) owns hidden unit j and a quarter of the reduction. h lives in LDS; the recurrent kernel U (H,3H) streams from L2 every step
// (coalesced across j).  Gate order z, r, h; bias_h = bias[3H:6H].
//   z = sig(xa_z + hU_z + b_hz); r = sig(xa_r + hU_r + b_hr); q = hU_h + b_hh
//   hh = tanh(xa_h + r*q);  h' = z*h + (1-z)*hh
// Saved for BPTT: gates[d][b][t][{z,r,hh,q}][j] and hprev[d][b][t][j].
#include "common.h"

namespace {

constexpr int GR = 4;   // batch rows per workgroup
constexpr int KS = 4;   // k-splits: the block is H x KS threads, thread (j, kq) covers a quarter of the reduction
                        // (4x the loads in flight per CU: the recurrence is L2-latency-bound, not FLOP-bound)

// dynamic LDS: hs[GR][H] | part[KS][GR][3][H]
__global__ void gru_fwd_kernel(const float* __restrict__ xa_f, const float* __restrict__ xa_b,
                               const float* __restrict__ U_f, const float* __restrict__ U_b,
                               const float* __restrict__ bh_f, const float* __restrict__ bh_b,
                               float* __restrict__ out, float* __restrict__ gates, float* __restrict__ hprev, int B,
                               int L, int H) {
    extern __shared__ float sm[];
    float* hs = sm; float* part = sm + GR * H;
    const int d = blockIdx.y, j = threadIdx.x % H, kq = threadIdx.x / H, b0 = blockIdx.x * GR;
    const float* xa = d ? xa_b : xa_f; const float* U = d ? U_b : U_f; const float* bh = d ? bh_b : bh_f;
    const int H3 = 3 * H;
    const int kper = (H + KS - 1) / KS, k0 = kq * kper, k1 = min(H, k0 + kper);
    const float bz = bh[j], br = bh[H + j], bq = bh[2 * H + j];
    for (int i = threadIdx.x; i < GR * H; i += blockDim.x) hs[i] = 0.f;
    __syncthreads();
    float* g_d = gates + (long)d * B * L * 4 * H;
    float* hp_d = hprev + (long)d * B * L * H;
    for (int s = 0; s < L; ++s) {
        const int t = d ? (L - 1 - s) : s;
        float az[GR], ar[GR], aq[GR];
#pragma unroll
        for (int r = 0; r < GR; ++r) { az[r] = 0.f; ar[r] = 0.f; aq[r] = 0.f; }
#pragma unroll 8
        for (int k = k0; k < k1; ++k) {
            const float u0 = U[(long)k * H3 + j], u1 = U[(long)k * H3 + H + j], u2 = U[(long)k * H3 + 2 * H + j];
#pragma unroll
            for (int r = 0; r < GR; ++r) {
                const float hv = hs[r * H + k];
                az[r] = fmaf(hv, u0, az[r]); ar[r] = fmaf(hv, u1, ar[r]); aq[r] = fmaf(hv, u2, aq[r]);
            }
        }
#pragma unroll
        for (int r = 0; r < GR; ++r) {
            float* p = part + ((kq * GR + r) * 3) * H;
            p[j] = az[r]; p[H + j] = ar[r]; p[2 * H + j] = aq[r];
        }
        __syncthreads();                               // partials complete; everyone is done reading hs
        for (int r = kq; r < GR; r += KS) {            // thread (j, kq) finishes batch row r = kq
            const int b = b0 + r;
            float sz = bz, sr = br, sq = bq;
#pragma unroll
            for (int q = 0; q < KS; ++q) {
                const float* p = part + ((q * GR + r) * 3) * H;
                sz += p[j]; sr += p[H + j]; sq += p[2 * H + j];
            }
            if (b < B) {
                const long xo = ((long)b * L + t) * H3;
                const float hold = hs[r * H + j];
                const float z = sigmoidf_(xa[xo + j] + sz);
                const float rr = sigmoidf_(xa[xo + H + j] + sr);
                const float hh = tanhf(xa[xo + 2 * H + j] + rr * sq);
                const float hn = z * hold + (1.f - z) * hh;
                const long go = (((long)b * L + t) * 4) * H;
                g_d[go + j] = z; g_d[go + H + j] = rr; g_d[go + 2 * H + j] = hh; g_d[go + 3 * H + j] = sq;
                hp_d[((long)b * L + t) * H + j] = hold;
                out[((long)b * L + t) * 2 * H + d * H + j] = hn;
                hs[r * H + j] = hn;                      // only this thread touches hs[r][j] in this phase
            }
        }
        __syncthreads();
    }
}

// BPTT.  UT = transpose(U): (3H, H).  dai/dah: (2, B, L, 3H) input-side / recurrent-side pre-activation grads.
// dynamic LDS: ds[GR][3H] | dhs[GR][H] | zs[GR][H] | part[KS][GR][H]
__global__ void gru_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ gates,
                               const float* __restrict__ hprev, const float* __restrict__ UT_f,
                               const float* __restrict__ UT_b, float* __restrict__ dai, float* __restrict__ dah, int B,
                               int L, int H) {
    extern __shared__ float sm[];
    const int H3 = 3 * H;
    float* ds = sm; float* dhs = ds + GR * H3; float* zs = dhs + GR * H; float* part = zs + GR * H;
    const int d = blockIdx.y, j = threadIdx.x % H, kq = threadIdx.x / H, b0 = blockIdx.x * GR;
    const float* UT = d ? UT_b : UT_f;
    const int cper = (H3 + KS - 1) / KS, c0 = kq * cper, c1 = min(H3, c0 + cper);
    const float* g_d = gates + (long)d * B * L * 4 * H;
    const float* hp_d = hprev + (long)d * B * L * H;
    float* dai_d = dai + (long)d * B * L * H3; float* dah_d = dah + (long)d * B * L * H3;
    for (int i = threadIdx.x; i < GR * H; i += blockDim.x) dhs[i] = 0.f;
    __syncthreads();
    for (int s = L - 1; s >= 0; --s) {
        const int t = d ? (L - 1 - s) : s;
        for (int r = kq; r < GR; r += KS) {            // gate backward for row r by thread (j, kq = r)
            const int b = b0 + r;
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, z = 0.f, g = 0.f;
            if (b < B) {
                const long go = (((long)b * L + t) * 4) * H;
                z = g_d[go + j];
                const float rr = g_d[go + H + j], hh = g_d[go + 2 * H + j], q = g_d[go + 3 * H + j];
                const float hp = hp_d[((long)b * L + t) * H + j];
                g = dhs[r * H + j] + dout[((long)b * L + t) * 2 * H + d * H + j];
                const float da_h = g * (1.f - z) * (1.f - hh * hh);
                const float da_z = g * (hp - hh) * z * (1.f - z);
                const float da_r = da_h * q * rr * (1.f - rr);
                const long o = ((long)b * L + t) * H3;
                dai_d[o + j] = da_z; dai_d[o + H + j] = da_r; dai_d[o + 2 * H + j] = da_h;
                a0 = da_z; a1 = da_r; a2 = da_h * rr;
                dah_d[o + j] = a0; dah_d[o + H + j] = a1; dah_d[o + 2 * H + j] = a2;
            }
            ds[r * H3 + j] = a0; ds[r * H3 + H + j] = a1; ds[r * H3 + 2 * H + j] = a2;
            dhs[r * H + j] = g; zs[r * H + j] = z;
        }
        __syncthreads();
        float acc[GR];
#pragma unroll
        for (int r = 0; r < GR; ++r) acc[r] = 0.f;
#pragma unroll 8
        for (int c = c0; c < c1; ++c) {
            const float u = UT[(long)c * H + j];
#pragma unroll
            for (int r = 0; r < GR; ++r) acc[r] = fmaf(ds[r * H3 + c], u, acc[r]);
        }
#pragma unroll
        for (int r = 0; r < GR; ++r) part[(kq * GR + r) * H + j] = acc[r];
        __syncthreads();
        for (int r = kq; r < GR; r += KS) {
            float a = 0.f;
#pragma unroll
            for (int q = 0; q < KS; ++q) a += part[(q * GR + r) * H + j];
            dhs[r * H + j] = dhs[r * H + j] * zs[r * H + j] + a;
        }
        __syncthreads();
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

}  // namespace

extern "C" int sar_bigru_fwd_f32(const float* xa_fwd, const float* xa_bwd, const float* U_fwd, const float* U_bwd,
                                 const float* bh_fwd, const float* bh_bwd, float* out, float* gates, float* hprev, int B,
                                 int L, int H, hipStream_t stream) {
    if (!xa_fwd || !xa_bwd || !U_fwd || !U_bwd || !bh_fwd || !bh_bwd || !out || !gates || !hprev || H < 1 || H > 1024)
        return SAR_EINVAL;
    if (H * KS > 1024) return SAR_EINVAL;
    hipLaunchKernelGGL(gru_fwd_kernel, dim3(cdiv(B, GR), 2), dim3(H * KS), sizeof(float) * (GR * H + KS * GR * 3 * H), stream,
                       xa_fwd, xa_bwd, U_fwd, U_bwd, bh_fwd, bh_bwd, out, gates, hprev, B, L, H);
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}

extern "C" int sar_bigru_bwd_f32(const float* dout, const float* gates, const float* hprev, const float* UT_fwd,
                                 const float* UT_bwd, float* dai, float* dah, int B, int L, int H, hipStream_t stream) {
    if (!dout || !gates || !hprev || !UT_fwd || !UT_bwd || !dai || !dah || H < 1 || H > 1024) return SAR_EINVAL;
    if (H * KS > 1024) return SAR_EINVAL;
    hipLaunchKernelGGL(gru_bwd_kernel, dim3(cdiv(B, GR), 2), dim3(H * KS), sizeof(float) * (GR * 3 * H + 2 * GR * H + KS * GR * H), stream,
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
