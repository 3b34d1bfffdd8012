// tools/cu_contention.py's kernels (developer tool, NOT part of libsarnet_hip.so): stand-ins for what a collective's kernels take from a train
// step that runs beside them on one GPU ("Next" item 4a).  No N > 1 box was available to any round, so the 33.6 MB all-reduce that the
// data-parallel step overlaps with its stage 1-0 + stem backward has never met a real RCCL ring; what CAN be measured on one GPU is the step's
// sensitivity to losing CUs (an RCCL channel is one workgroup that sits on a CU for the whole collective) and memory bandwidth.
//   hog_exclusive: `blocks` workgroups of 1024 threads that declare the whole LDS -- nothing else fits on their CU -- and spin for `us` microseconds
//   (device-wide 100 MHz clock): the CU is GONE, as under a CU mask.
//   hog_channel: `blocks` workgroups of `threads` threads and `lds` bytes that copy a buffer of `n16` 16-byte pieces round and round for `us`
//   microseconds: a ring all-reduce channel's footprint (512 threads, tens of KB of LDS, a stream of loads and stores) without the links.
#include <hip/hip_runtime.h>

namespace {

__global__ __launch_bounds__(1024) void hog_exclusive_kernel(unsigned us, unsigned* out) {
    extern __shared__ unsigned char smem[];
    if (threadIdx.x == 0) smem[0] = 1;                       // (the allocation must be used)
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)us * 100ull) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0 && out != nullptr) atomicAdd(out, (unsigned)smem[0]);
}

__global__ __launch_bounds__(1024) void hog_channel_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, long n16, unsigned us, unsigned* out) {
    extern __shared__ unsigned char smem[];
    if (threadIdx.x == 0) smem[0] = 1;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const long per = n16 / gridDim.x, base = (long)blockIdx.x * per;
    unsigned acc = 0;
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)us * 100ull) {
        for (long i = threadIdx.x; i < per; i += blockDim.x) {  // one sweep of this block's share: load, (reduce), store
            uint4 v = src[base + i];
            v.x += 1u; acc += v.y;
            dst[base + i] = v;
        }
    }
    if (threadIdx.x == 0 && out != nullptr) atomicAdd(out, acc + (unsigned)smem[0]);
}

}  // namespace

extern "C" {

int hog_exclusive(int blocks, unsigned us, void* out, hipStream_t s) {
    static bool attr = false;
    if (!attr) { if (hipFuncSetAttribute(reinterpret_cast<const void*>(hog_exclusive_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return -1; attr = true; }
    hipLaunchKernelGGL(hog_exclusive_kernel, dim3(blocks), dim3(1024), 160 * 1024, s, us, static_cast<unsigned*>(out));
    return (int)hipGetLastError();
}

int hog_channel(int blocks, int threads, int lds, const void* src, void* dst, long n16, unsigned us, void* out, hipStream_t s) {
    static bool attr = false;
    if (!attr) { if (hipFuncSetAttribute(reinterpret_cast<const void*>(hog_channel_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return -1; attr = true; }
    hipLaunchKernelGGL(hog_channel_kernel, dim3(blocks), dim3(threads), lds, s, static_cast<const uint4*>(src), static_cast<uint4*>(dst), n16, us, static_cast<unsigned*>(out));
    return (int)hipGetLastError();
}

}  // extern "C"
