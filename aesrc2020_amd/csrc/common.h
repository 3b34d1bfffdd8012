// Shared device helpers for the SAR-Net gfx950 (CDNA4) kernels.
// Wave = 64 lanes everywhere; no CUDA compatibility paths.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SAR_OK 0
#define SAR_EINVAL (-1)
#define SAR_ELAUNCH (-2)

#define SAR_CHECK_LAUNCH()                                   \
    do {                                                     \
        hipError_t e__ = hipGetLastError();                  \
        if (e__ != hipSuccess) return SAR_ELAUNCH;           \
    } while (0)

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// A "done once" flag that is per DEVICE, not per process: hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-device
// attribute, so a process that touches a second device has to raise the cap there as well.  Reads / assigns like the `static bool` it replaces.
struct PerDeviceFlag {
    unsigned long long seen = 0;
    static unsigned long long bit() { int d = 0; (void)hipGetDevice(&d); return 1ull << (d & 63); }
    operator bool() const { return (seen & bit()) != 0; }
    PerDeviceFlag& operator=(bool v) { if (v) seen |= bit(); else seen &= ~bit(); return *this; }
};

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt(0): inside a recurrence whose steps store to (or
// atomically update) global memory, every step then waits for the write acknowledgements of the step before (~1 us each) although the
// threads only exchange data through LDS.  (An asm statement with a memory clobber: the s_barrier builtin alone does not keep LDS
// accesses on their side of it.)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Wave-wide reductions without LDS: four DPP steps give every lane the result of its row of 16 lanes (quad_perm [1,0,3,2], quad_perm
// [2,3,0,1], row_half_mirror, row_mirror), v_readlane of one lane per row combines the four rows.  (__shfl_xor is ds_bpermute_b32 on this
// target -- six dependent LDS round trips per reduction, and most kernels here are one wave per row waiting on exactly that.)
#define SAR_DPP_F(v, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), 0xf, 0xf, true))
__device__ __forceinline__ float wave_sum(float v) {
    v += SAR_DPP_F(v, 0xB1); v += SAR_DPP_F(v, 0x4E); v += SAR_DPP_F(v, 0x141); v += SAR_DPP_F(v, 0x140);
    const int i = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 48));
    return (r0 + r1) + (r2 + r3);
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
    // inactive / out-of-row sources read as 0 with bound_ctrl, which is wrong for a maximum of negative numbers: these four patterns
    // are permutations within the row, every lane has a source
    v = fmaxf(v, SAR_DPP_F(v, 0xB1)); v = fmaxf(v, SAR_DPP_F(v, 0x4E)); v = fmaxf(v, SAR_DPP_F(v, 0x141)); v = fmaxf(v, SAR_DPP_F(v, 0x140));
    const int i = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 48));
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}

// Block-wide sum; `red` is a __shared__ float[16] scratch. All threads get the result.
__device__ __forceinline__ float block_sum(float v, float* red) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float r = 0.f;
    for (int i = 0; i < nw; ++i) r += red[i];
    return r;
}
__device__ __forceinline__ float block_max(float v, float* red) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_max(v);
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float r = red[0];
    for (int i = 1; i < nw; ++i) r = fmaxf(r, red[i]);
    return r;
}

__device__ __forceinline__ float sigmoidf_(float a) { return 1.f / (1.f + expf(-a)); }

// log(exp(a)+exp(b)) with -inf handling
__device__ __forceinline__ float lse2(float a, float b) {
    const float m = fmaxf(a, b);
    if (m == -INFINITY) return -INFINITY;
    return m + logf(expf(a - m) + expf(b - m));
}

// Zero-fill as a kernel node.  hipMemsetAsync nodes inside a captured HIP graph were observed (ROCm 7.2, gfx950) to
// corrupt split-K accumulation on replay, so the library never uses memset on the launch path.
__global__ static void sar_zero_kernel(uint32_t* __restrict__ p, size_t n_words) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (size_t)gridDim.x * blockDim.x) p[i] = 0u;
}
static inline hipError_t sar_zero_async(void* p, size_t bytes, hipStream_t stream) {
    const size_t n = bytes / 4;                 // all buffers here are multiples of 4 bytes
    size_t nb = (n + 255) / 256;
    if (nb > 4096) nb = 4096;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(sar_zero_kernel, dim3((unsigned)nb), dim3(256), 0, stream, static_cast<uint32_t*>(p), n);
    return hipGetLastError();
}

// activation codes shared by host and device (include/sarnet.h mirrors them)
enum { ACT_NONE = 0, ACT_TANH = 1, ACT_RELU = 2 };
