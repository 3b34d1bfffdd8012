// tools/launch_floor.py's kernels (developer tool, NOT part of libsarnet_hip.so): what a dependent kernel boundary costs on this
// stack as a function of what the launch carries -- nothing, kernel arguments, dynamic LDS, threads per block, a memory round trip,
// a streaming body.  MI355X_MICROARCH.md ("boundary" row of the price list) measures 1.45 us between trivial 256-workgroup kernels
// and 1.7-1.9 us between streaming kernels; DESIGN.md (round 2) quoted ~4.5 us per launch of a replayed graph's timeline.
#include <hip/hip_runtime.h>

namespace {

struct BigArg { unsigned long long v[40]; };   // 320 bytes by value, the size class of conv_bf16.hip's ConvP

__global__ void k_empty() {}
__global__ void k_empty_big(const BigArg a) { if (a.v[0] == 0x123456789abcdefULL) __builtin_trap(); }
__global__ void k_lds() {
    extern __shared__ unsigned char smem[];
    if (threadIdx.x == 0) smem[0] = 1;         // the allocation must be used or the compiler drops it
}
// one dependent round trip per thread: load, add, store (a chain of these is a chain of real RAW dependencies through memory)
__global__ void k_touch(float* p) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    p[i] = p[i] + 1.f;
}
// 16-byte streaming copy, grid-stride
__global__ void k_stream(const float4* __restrict__ src, float4* __restrict__ dst, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) dst[i] = src[i];
}
// a body of fixed duration with no memory traffic: `cycles` shader cycles of s_sleep-free spinning on s_memtime
__global__ void k_spin(int cycles) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while ((long long)(__builtin_amdgcn_s_memtime() - t0) < cycles) {}
}

// a body of `cycles` shader cycles that stamps the device-wide 100 MHz clock at block start and end: buf[(id * gridDim.x + block) * 2 + {0, 1}].
// A chain of these shows where a boundary's time goes: dispatch ramp (last start - first start), drain (last end - first end) and the
// gap between the last end of launch i and the first start of launch i + 1.
__global__ void k_stamp(unsigned long long* buf, int id, int cycles) {
    extern __shared__ unsigned char smem[];
    unsigned long long t0 = 0;
    if (threadIdx.x == 0) { t0 = __builtin_amdgcn_s_memrealtime(); smem[0] = 1; }
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    while ((long long)(__builtin_amdgcn_s_memtime() - c0) < cycles) {}
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long* o = buf + ((size_t)id * gridDim.x + blockIdx.x) * 2;
        o[0] = t0; o[1] = __builtin_amdgcn_s_memrealtime();
    }
}

}  // namespace

extern "C" {

int lf_stamp(void* buf, int id, int cycles, int grid, int threads, int lds_bytes, hipStream_t s) {
    static bool attr = false;
    if (!attr) { if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_stamp), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return -1; attr = true; }
    hipLaunchKernelGGL(k_stamp, dim3(grid), dim3(threads), lds_bytes, s, static_cast<unsigned long long*>(buf), id, cycles);
    return (int)hipGetLastError();
}

int lf_empty(int grid, int threads, hipStream_t s) { hipLaunchKernelGGL(k_empty, dim3(grid), dim3(threads), 0, s); return (int)hipGetLastError(); }
int lf_empty_big(int grid, int threads, hipStream_t s) {
    BigArg a; for (int i = 0; i < 40; ++i) a.v[i] = (unsigned long long)i;
    hipLaunchKernelGGL(k_empty_big, dim3(grid), dim3(threads), 0, s, a);
    return (int)hipGetLastError();
}
int lf_lds(int grid, int threads, int lds_bytes, hipStream_t s) {
    static bool attr = false;
    if (!attr) { if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_lds), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return -1; attr = true; }
    hipLaunchKernelGGL(k_lds, dim3(grid), dim3(threads), lds_bytes, s);
    return (int)hipGetLastError();
}
int lf_touch(float* p, int grid, int threads, hipStream_t s) { hipLaunchKernelGGL(k_touch, dim3(grid), dim3(threads), 0, s, p); return (int)hipGetLastError(); }
int lf_stream(const void* src, void* dst, long n16, int grid, hipStream_t s) {
    hipLaunchKernelGGL(k_stream, dim3(grid), dim3(256), 0, s, static_cast<const float4*>(src), static_cast<float4*>(dst), n16);
    return (int)hipGetLastError();
}
int lf_spin(int grid, int threads, int cycles, hipStream_t s) { hipLaunchKernelGGL(k_spin, dim3(grid), dim3(threads), 0, s, cycles); return (int)hipGetLastError(); }

// n launches of the empty kernel from a C loop (no Python between launches), bracketed by events on the launch stream: the eager
// figure with the host at its fastest
float lf_eager_chain(int n, int grid, int threads, hipStream_t s) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 64; ++i) hipLaunchKernelGGL(k_empty, dim3(grid), dim3(threads), 0, s);
    hipStreamSynchronize(s);
    hipEventRecord(e0, s);
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_empty, dim3(grid), dim3(threads), 0, s);
    hipEventRecord(e1, s);
    hipStreamSynchronize(s);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return ms * 1e3f / (float)n;
}

// the same chain as an explicitly built hipGraph of n kernel nodes (no stream capture, no torch): us per node of one replay
float lf_graph_chain(int n, int grid, int threads, int reps, hipStream_t s) {
    hipGraph_t g; hipGraphExec_t ge;
    if (hipGraphCreate(&g, 0) != hipSuccess) return -1.f;
    hipGraphNode_t prev = nullptr;
    for (int i = 0; i < n; ++i) {
        hipKernelNodeParams kp = {};
        kp.func = reinterpret_cast<void*>(k_empty);
        kp.gridDim = dim3(grid); kp.blockDim = dim3(threads); kp.sharedMemBytes = 0; kp.kernelParams = nullptr; kp.extra = nullptr;
        hipGraphNode_t node;
        if (hipGraphAddKernelNode(&node, g, prev ? &prev : nullptr, prev ? 1 : 0, &kp) != hipSuccess) return -2.f;
        prev = node;
    }
    if (hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) != hipSuccess) return -3.f;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipGraphLaunch(ge, s); hipStreamSynchronize(s);
    hipEventRecord(e0, s);
    for (int r = 0; r < reps; ++r) hipGraphLaunch(ge, s);
    hipEventRecord(e1, s);
    hipStreamSynchronize(s);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
    return ms * 1e3f / (float)(n * reps);
}

}  // extern "C"
