// Small HBM-bound helpers for the SAR-Net step on gfx950: activation backward, row softmax (Dense softmax heads,
// model.py:268,296,150), column sums (bias gradients), axpy, GlobalAveragePooling1D (model.py:125),
// row-wise l2_normalize (model.py:162), 2-D transpose (GRU recurrent kernel for BPTT), fill.
#include "common.h"

namespace {

__global__ void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dz, long n,
                               int act) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float g = dy[i], v = y[i];
        dz[i] = act == ACT_TANH ? g * (1.f - v * v) : (act == ACT_RELU ? (v > 0.f ? g : 0.f) : g);
    }
}

// one block per row
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const float* __restrict__ z, float* __restrict__ p, int N) {
    __shared__ float red[16];
    const float* zr = z + (long)blockIdx.x * N; float* pr = p + (long)blockIdx.x * N;
    float m = -INFINITY;
    for (int i = threadIdx.x; i < N; i += blockDim.x) m = fmaxf(m, zr[i]);
    m = block_max(m, red);
    float s = 0.f;
    for (int i = threadIdx.x; i < N; i += blockDim.x) { const float e = expf(zr[i] - m); pr[i] = e; s += e; }
    s = block_sum(s, red);
    const float inv = 1.f / s;
    for (int i = threadIdx.x; i < N; i += blockDim.x) pr[i] *= inv;
}

// One WAVE per row, the row in registers (N % 4 == 0, N <= 1024: the bpe-1k CTC output layer, model.py:270): one pass over the row instead of
// three, wave reductions instead of two block reductions, v_exp_f32 instead of libm's expf (round 4: 15.9 -> 5.9 us forward, 8.6 -> 6.7 us backward for 2 496 x 1 000).
constexpr int SMW_V = 4;                       // float4 per lane
__global__ __launch_bounds__(256) void softmax_fwd_wave_kernel(const float* __restrict__ z, float* __restrict__ p, int rows, int N) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float4* zr = reinterpret_cast<const float4*>(z + (long)row * N); float4* pr = reinterpret_cast<float4*>(p + (long)row * N);
    const int n4 = N >> 2;
    float4 v[SMW_V];
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < SMW_V; ++j) {
        const int i = lane + 64 * j;
        v[j] = i < n4 ? zr[i] : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        m = fmaxf(m, fmaxf(fmaxf(v[j].x, v[j].y), fmaxf(v[j].z, v[j].w)));
    }
    m = wave_max(m);
    float s = 0.f;
    constexpr float L2E = 1.44269504f;
#pragma unroll
    for (int j = 0; j < SMW_V; ++j) {
        v[j].x = __builtin_amdgcn_exp2f((v[j].x - m) * L2E); v[j].y = __builtin_amdgcn_exp2f((v[j].y - m) * L2E);
        v[j].z = __builtin_amdgcn_exp2f((v[j].z - m) * L2E); v[j].w = __builtin_amdgcn_exp2f((v[j].w - m) * L2E);
        s += (v[j].x + v[j].y) + (v[j].z + v[j].w);              // (lanes past the row: exp2(-inf) = 0)
    }
    s = wave_sum(s);
    const float inv = 1.f / s;
#pragma unroll
    for (int j = 0; j < SMW_V; ++j) {
        const int i = lane + 64 * j;
        if (i < n4) pr[i] = make_float4(v[j].x * inv, v[j].y * inv, v[j].z * inv, v[j].w * inv);
    }
}
__global__ __launch_bounds__(256) void softmax_bwd_wave_kernel(const float* __restrict__ dp, const float* __restrict__ p, float* __restrict__ dz, int rows, int N) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const long o = (long)row * N;
    const float4* dr = reinterpret_cast<const float4*>(dp + o); const float4* pr = reinterpret_cast<const float4*>(p + o);
    float4* zr = reinterpret_cast<float4*>(dz + o);
    const int n4 = N >> 2;
    float4 d[SMW_V], q[SMW_V];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < SMW_V; ++j) {
        const int i = lane + 64 * j;
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        d[j] = i < n4 ? dr[i] : z4; q[j] = i < n4 ? pr[i] : z4;
        s = fmaf(d[j].x, q[j].x, s); s = fmaf(d[j].y, q[j].y, s); s = fmaf(d[j].z, q[j].z, s); s = fmaf(d[j].w, q[j].w, s);
    }
    s = wave_sum(s);
#pragma unroll
    for (int j = 0; j < SMW_V; ++j) {
        const int i = lane + 64 * j;
        if (i < n4) zr[i] = make_float4(q[j].x * (d[j].x - s), q[j].y * (d[j].y - s), q[j].z * (d[j].z - s), q[j].w * (d[j].w - s));
    }
}

// dz = p * (dp - sum(dp*p))
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* __restrict__ dp, const float* __restrict__ p,
                                                          float* __restrict__ dz, int N) {
    __shared__ float red[16];
    const long o = (long)blockIdx.x * N;
    float s = 0.f;
    for (int i = threadIdx.x; i < N; i += blockDim.x) s = fmaf(dp[o + i], p[o + i], s);
    s = block_sum(s, red);
    for (int i = threadIdx.x; i < N; i += blockDim.x) dz[o + i] = p[o + i] * (dp[o + i] - s);
}

// out[c] += sum_r x[r][c].  C % 4 == 0: float4 lanes over channel-vectors x row-groups (coalesced), <= 128 blocks,
// one fp32 atomic per column per block.  Otherwise (tiny C such as the 8 accents): thread-per-column fallback.
__global__ __launch_bounds__(256) void colsum_vec_kernel(const float* __restrict__ x, float* __restrict__ out, long R,
                                                         int C, int tpr) {
    __shared__ float4 s1[256];
    const int nrg = 256 / tpr, rg = threadIdx.x / tpr;
    for (int cv0 = 0; cv0 * 4 < C; cv0 += tpr) {            // C > 1024: several passes over column groups
        const int cv = cv0 + threadIdx.x % tpr;
        float4 a = make_float4(0, 0, 0, 0);
        if (cv * 4 < C)
            for (long r = (long)blockIdx.x * nrg + rg; r < R; r += (long)gridDim.x * nrg) {
                const float4 v = *reinterpret_cast<const float4*>(x + r * C + cv * 4);
                a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
            }
        __syncthreads();
        s1[threadIdx.x] = a;
        __syncthreads();
        if (threadIdx.x < tpr && cv * 4 < C) {
            float4 d = make_float4(0, 0, 0, 0);
            for (int g = 0; g < nrg; ++g) { const float4 u = s1[g * tpr + threadIdx.x]; d.x += u.x; d.y += u.y; d.z += u.z; d.w += u.w; }
            atomicAdd(out + cv * 4 + 0, d.x); atomicAdd(out + cv * 4 + 1, d.y);
            atomicAdd(out + cv * 4 + 2, d.z); atomicAdd(out + cv * 4 + 3, d.w);
        }
    }
}
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, float* __restrict__ out, long R, int C,
                                                     int rows_per_block) {
    const long r0 = (long)blockIdx.x * rows_per_block, r1 = min(R, r0 + rows_per_block);
    for (int c = threadIdx.x; c < C; c += 256) {
        float s = 0.f;
        for (long r = r0; r < r1; ++r) s += x[r * C + c];
        atomicAdd(out + c, s);
    }
}

__global__ void axpy_kernel(const float* __restrict__ x, float* __restrict__ y, float a, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i] = fmaf(a, x[i], y[i]);
}

__global__ void fill_kernel(float* __restrict__ y, float v, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i] = v;
}

// x (B, L, C) -> y (B, C) mean over L
__global__ void avgpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int L, int C) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * C) return;
    const int b = (int)(i / C), c = (int)(i % C);
    float s = 0.f;
    for (int l = 0; l < L; ++l) s += x[((long)b * L + l) * C + c];
    y[i] = s / L;
}
__global__ void avgpool_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int B, int L, int C, int accumulate) {
    const long n = (long)B * L * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C); const int b = (int)(i / ((long)L * C));
        const float g = dy[(long)b * C + c] / L;
        dx[i] = accumulate ? dx[i] + g : g;
    }
}

// rows: v = x * rsqrt(max(sum x^2, 1e-12)); one wave per row
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float* __restrict__ x, float* __restrict__ v,
                                                         float* __restrict__ inv, int R, int C) {
    const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) { const float a = x[(long)r * C + c]; s = fmaf(a, a, s); }
    s = wave_sum(s);
    const float k = 1.f / sqrtf(fmaxf(s, 1e-12f));
    if (lane == 0) inv[r] = (s >= 1e-12f) ? k : -k;      // sign flags the clamped branch for backward
    for (int c = lane; c < C; c += 64) v[(long)r * C + c] = x[(long)r * C + c] * k;
}
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ dv, const float* __restrict__ v,
                                                         const float* __restrict__ inv, float* __restrict__ dx, int R,
                                                         int C, int accumulate) {
    const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const float k = inv[r];
    float s = 0.f;
    if (k > 0.f) for (int c = lane; c < C; c += 64) s = fmaf(dv[(long)r * C + c], v[(long)r * C + c], s);
    s = wave_sum(s);
    const float ak = fabsf(k);
    for (int c = lane; c < C; c += 64) {
        const long i = (long)r * C + c;
        const float g = (dv[i] - v[i] * s) * ak;
        dx[i] = accumulate ? dx[i] + g : g;
    }
}

__global__ void transpose_kernel(const float* __restrict__ x, float* __restrict__ y, int R, int C) {
    __shared__ float t[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    for (int j = threadIdx.y; j < 32; j += blockDim.y) {
        const int r = by + j, c = bx + threadIdx.x;
        if (r < R && c < C) t[j][threadIdx.x] = x[(long)r * C + c];
    }
    __syncthreads();
    for (int j = threadIdx.y; j < 32; j += blockDim.y) {
        const int c = bx + j, r = by + threadIdx.x;
        if (r < R && c < C) y[(long)c * R + r] = t[threadIdx.x][j];
    }
}

__global__ __launch_bounds__(256) void zero_ranges_kernel(const long long* __restrict__ table) {
    uint4* p = reinterpret_cast<uint4*>(table[2 * blockIdx.y]);
    const long long nw = table[2 * blockIdx.y + 1], n4 = nw >> 2;              // 16-byte aligned buffers (torch allocations)
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += 256LL * gridDim.x) p[i] = make_uint4(0u, 0u, 0u, 0u);
    if (blockIdx.x == 0 && threadIdx.x < (nw & 3)) reinterpret_cast<unsigned*>(p)[n4 * 4 + threadIdx.x] = 0u;
}

// rows {destination, source, 32-bit words} (3 x int64); both 16-byte aligned
__global__ __launch_bounds__(256) void copy_ranges_kernel(const long long* __restrict__ table) {
    uint4* d = reinterpret_cast<uint4*>(table[3 * blockIdx.y]);
    const uint4* s = reinterpret_cast<const uint4*>(table[3 * blockIdx.y + 1]);
    const long long nw = table[3 * blockIdx.y + 2], n4 = nw >> 2;
    // four 16-byte loads in flight per thread: with one, the 8 MB input batch was eight dependent round trips per thread (9 us; now bound by the launch)
    const long long st = 256LL * gridDim.x;
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * st < n4; i += 4 * st) {
        const uint4 a = s[i], b = s[i + st], c = s[i + 2 * st], e = s[i + 3 * st];
        d[i] = a; d[i + st] = b; d[i + 2 * st] = c; d[i + 3 * st] = e;
    }
    for (; i < n4; i += st) d[i] = s[i];
    if (blockIdx.x == 0 && threadIdx.x < (nw & 3))
        reinterpret_cast<unsigned*>(d)[n4 * 4 + threadIdx.x] = reinterpret_cast<const unsigned*>(s)[n4 * 4 + threadIdx.x];
}

}  // namespace

static int ew_blocks(long n) { long b = (n + 255) / 256; return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b)); }

extern "C" int sar_act_bwd_f32(const float* dy, const float* y, float* dz, long n, int act, hipStream_t stream) {
    if (!dy || !y || !dz) return SAR_EINVAL;
    hipLaunchKernelGGL(act_bwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, stream, dy, y, dz, n, act);
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}
extern "C" int sar_softmax_fwd_f32(const float* z, float* p, int rows, int N, hipStream_t stream) {
    if (!z || !p || rows <= 0 || N <= 0) return SAR_EINVAL;
    if (N % 4 == 0 && N > 64 && N <= 256 * SMW_V && ((reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(p)) & 15) == 0) {
        hipLaunchKernelGGL(softmax_fwd_wave_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, z, p, rows, N);
        SAR_CHECK_LAUNCH();
        return SAR_OK;
    }
    hipLaunchKernelGGL(softmax_fwd_kernel, dim3(rows), dim3(N <= 64 ? 64 : 256), 0, stream, z, p, N);
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}
extern "C" int sar_softmax_bwd_f32(const float* dp, const float* p, float* dz, int rows, int N, hipStream_t stream) {
    if (!dp || !p || !dz || rows <= 0 || N <= 0) return SAR_EINVAL;
    if (N % 4 == 0 && N > 64 && N <= 256 * SMW_V &&
        ((reinterpret_cast<uintptr_t>(dp) | reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(dz)) & 15) == 0) {
        hipLaunchKernelGGL(softmax_bwd_wave_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, dp, p, dz, rows, N);
        SAR_CHECK_LAUNCH();
        return SAR_OK;
    }
    hipLaunchKernelGGL(softmax_bwd_kernel, dim3(rows), dim3(N <= 64 ? 64 : 256), 0, stream, dp, p, dz, N);
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}
extern "C" int sar_colsum_f32(const float* x, float* out, long R, int C, hipStream_t stream) {
    if (!x || !out || R <= 0 || C <= 0) return SAR_EINVAL;
    if (C % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
        int tpr = 1; while (tpr < C / 4 && tpr < 256) tpr <<= 1;
        const int nrg = 256 / tpr;
        long nb = (R + nrg * 8 - 1) / (nrg * 8);
        if (nb > 128) nb = 128;
        if (nb < 1) nb = 1;
        hipLaunchKernelGGL(colsum_vec_kernel, dim3((int)nb), dim3(256), 0, stream, x, out, R, C, tpr);
    } else {
        int rpb = (int)((R + 511) / 512); if (rpb < 8) rpb = 8;
        hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(R, rpb)), dim3(256), 0, stream, x, out, R, C, rpb);
    }
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}
extern "C" int sar_axpy_f32(const float* x, float* y, float a, long n, hipStream_t stream) {
    if (!x || !y) return SAR_EINVAL;
    hipLaunchKernelGGL(axpy_kernel, dim3(ew_blocks(n)), dim3(256), 0, stream, x, y, a, n);
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}
// Zero several buffers in one launch: table = device array of n rows {pointer, 32-bit words} (2 x int64); grid.y = row.
// The train step needs its gradient buffer, the BatchNorm statistic bins, split-K outputs and a few scalars clean when it starts;
// one launch each was four ~5 us launches.
extern "C" int sar_zero_ranges(const void* table, int n, hipStream_t stream) {
    if (!table || n < 1 || n > 65535) return SAR_EINVAL;
    hipLaunchKernelGGL(zero_ranges_kernel, dim3(256, n), dim3(256), 0, stream, static_cast<const long long*>(table));
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}
// Copy several buffers in one launch (the input tensors of a device-resident batch into the step's static input buffers):
// table = device array of n rows {destination, source, number of 32-bit words} as 3 x int64, 16-byte aligned pointers.
extern "C" int sar_copy_ranges(const void* table, int n, hipStream_t stream) {
    if (!table || n < 1 || n > 65535) return SAR_EINVAL;
    hipLaunchKernelGGL(copy_ranges_kernel, dim3(512, n), dim3(256), 0, stream, static_cast<const long long*>(table));
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}
extern "C" int sar_fill_f32(float* y, float v, long n, hipStream_t stream) {
    if (!y) return SAR_EINVAL;
    hipLaunchKernelGGL(fill_kernel, dim3(ew_blocks(n)), dim3(256), 0, stream, y, v, n);
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}
extern "C" int sar_avgpool1d_fwd_f32(const float* x, float* y, int B, int L, int C, hipStream_t stream) {
    if (!x || !y) return SAR_EINVAL;
    hipLaunchKernelGGL(avgpool_fwd_kernel, dim3(cdiv((long)B * C, 256)), dim3(256), 0, stream, x, y, B, L, C);
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}
extern "C" int sar_avgpool1d_bwd_f32(const float* dy, float* dx, int B, int L, int C, int accumulate, hipStream_t stream) {
    if (!dy || !dx) return SAR_EINVAL;
    hipLaunchKernelGGL(avgpool_bwd_kernel, dim3(ew_blocks((long)B * L * C)), dim3(256), 0, stream, dy, dx, B, L, C, accumulate);
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}
extern "C" int sar_l2norm_rows_fwd_f32(const float* x, float* v, float* inv, int R, int C, hipStream_t stream) {
    if (!x || !v || !inv) return SAR_EINVAL;
    hipLaunchKernelGGL(l2norm_fwd_kernel, dim3(cdiv(R, 4)), dim3(256), 0, stream, x, v, inv, R, C);
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}
extern "C" int sar_l2norm_rows_bwd_f32(const float* dv, const float* v, const float* inv, float* dx, int R, int C,
                                       int accumulate, hipStream_t stream) {
    if (!dv || !v || !inv || !dx) return SAR_EINVAL;
    hipLaunchKernelGGL(l2norm_bwd_kernel, dim3(cdiv(R, 4)), dim3(256), 0, stream, dv, v, inv, dx, R, C, accumulate);
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}
extern "C" int sar_transpose_f32(const float* x, float* y, int R, int C, hipStream_t stream) {
    if (!x || !y) return SAR_EINVAL;
    hipLaunchKernelGGL(transpose_kernel, dim3(cdiv(C, 32), cdiv(R, 32)), dim3(32, 8), 0, stream, x, y, R, C);
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}
