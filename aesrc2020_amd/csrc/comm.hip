// Gradient all-reduce of the data-parallel step over RCCL (xGMI), behind the C-ABI: what keras.utils.multi_gpu_model's gradient
// aggregation is for the reference (model.py:193-194), one process per GPU here.  RCCL is bound at run time (dlopen: the copy this
// process already has -- PyTorch ships one -- or /opt/rocm's), so the library loads on hosts without it and fails loudly only when a
// communicator is asked for.  No transfers are implemented here: ncclAllReduce does the work; this file is the boundary.
#include "common.h"
#include <dlfcn.h>
#include <string.h>
#include <mutex>

namespace {

struct Uid { char b[128]; };                                   // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
typedef int (*get_uid_t)(Uid*);
typedef int (*init_rank_t)(void**, int, Uid, int);
typedef int (*allreduce_t)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*destroy_t)(void*);
struct Rccl { void* h; get_uid_t get_uid; init_rank_t init_rank; allreduce_t allreduce; destroy_t destroy; };

const Rccl* rccl() {
    static Rccl r{};
    static std::once_flag once;                                  // (the lazy binding was not thread-safe)
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            if ((r.h = dlopen(name, RTLD_NOW | RTLD_GLOBAL)) != nullptr) break;
        }
        if (r.h) {
            r.get_uid = reinterpret_cast<get_uid_t>(dlsym(r.h, "ncclGetUniqueId"));
            r.init_rank = reinterpret_cast<init_rank_t>(dlsym(r.h, "ncclCommInitRank"));
            r.allreduce = reinterpret_cast<allreduce_t>(dlsym(r.h, "ncclAllReduce"));
            r.destroy = reinterpret_cast<destroy_t>(dlsym(r.h, "ncclCommDestroy"));
            if (!r.get_uid || !r.init_rank || !r.allreduce || !r.destroy) r.h = nullptr;
        }
    });
    return r.h ? &r : nullptr;
}
constexpr int NCCL_FLOAT32 = 7, NCCL_BFLOAT16 = 9, NCCL_SUM = 0;   // rccl.h: ncclFloat32, ncclBfloat16, ncclSum

// fp32 <-> bf16 (round to nearest even) of a gradient bucket: 16-byte pieces, tail element-wise
__global__ __launch_bounds__(256) void grad_to_bf16_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst, long n) {
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long)gridDim.x * 1024) {
        if (i + 3 < n) {
            const float4 v = *reinterpret_cast<const float4*>(src + i);
            __bf16 b[4] = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
            *reinterpret_cast<uint2*>(dst + i) = *reinterpret_cast<const uint2*>(b);
        } else {
            for (long j = i; j < n; ++j) { const __bf16 b = (__bf16)src[j]; dst[j] = *reinterpret_cast<const unsigned short*>(&b); }
        }
    }
}
__global__ __launch_bounds__(256) void grad_from_bf16_kernel(const unsigned short* __restrict__ src, float* __restrict__ dst, long n) {
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long)gridDim.x * 1024) {
        if (i + 3 < n) {
            const uint2 q = *reinterpret_cast<const uint2*>(src + i);
            *reinterpret_cast<float4*>(dst + i) = make_float4(__uint_as_float(q.x << 16), __uint_as_float(q.x & 0xffff0000u),
                                                              __uint_as_float(q.y << 16), __uint_as_float(q.y & 0xffff0000u));
        } else {
            for (long j = i; j < n; ++j) dst[j] = __uint_as_float((unsigned)src[j] << 16);
        }
    }
}

}  // namespace

// 128 bytes identifying a new communicator: rank 0 creates it, every rank passes the same bytes to sar_comm_init (exchange them over
// whatever channel the launcher has: torch.distributed's store, a file, MPI).
extern "C" int sar_comm_unique_id(void* out128) {
    const Rccl* r = rccl();
    if (!r) return SAR_ELAUNCH;
    if (!out128) return SAR_EINVAL;
    Uid u;
    if (r->get_uid(&u) != 0) return SAR_ELAUNCH;
    memcpy(out128, u.b, sizeof(u.b));
    return SAR_OK;
}

// Joins rank `rank` of `nranks` to the communicator `id128` on the CURRENT HIP device (collective: every rank calls it).
extern "C" int sar_comm_init(void** comm, const void* id128, int nranks, int rank) {
    const Rccl* r = rccl();
    if (!r) return SAR_ELAUNCH;
    if (!comm || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return SAR_EINVAL;
    Uid u;
    memcpy(u.b, id128, sizeof(u.b));
    return r->init_rank(comm, nranks, u, rank) == 0 ? SAR_OK : SAR_ELAUNCH;
}

// In-place sum over all ranks of buf[0 .. count) (fp32): one bucket of the flat gradient buffer.  Enqueued on `stream`; returns at once.
extern "C" int sar_allreduce_bucket_f32(void* comm, float* buf, long count, hipStream_t stream) {
    const Rccl* r = rccl();
    if (!r) return SAR_ELAUNCH;
    if (!comm || !buf || count < 0) return SAR_EINVAL;
    if (count == 0) return SAR_OK;
    return r->allreduce(buf, buf, (size_t)count, NCCL_FLOAT32, NCCL_SUM, comm, stream) == 0 ? SAR_OK : SAR_ELAUNCH;
}

// The same bucket through HALF the bytes (optional: SAR_DP_GRAD_DTYPE=bf16): the fp32 gradients are rounded to bf16
// into `scratch16` (count 2-byte elements, 8-byte aligned; buf 16-byte aligned), summed over the ranks by RCCL in bf16, and the sum is
// widened back into buf -- the fp32 master gradient Adam reads.  17 MB instead of 34 on the ring at F = 32; each hop of the ring rounds
// its partial sum to 8 mantissa bits, so this is a bandwidth / precision trade the caller opts into.  Three operations enqueued on
// `stream` (cast, ncclAllReduce, cast); returns at once.
extern "C" int sar_allreduce_bucket_bf16(void* comm, float* buf, void* scratch16, long count, hipStream_t stream) {
    const Rccl* r = rccl();
    if (!r) return SAR_ELAUNCH;
    if (!comm || !buf || !scratch16 || count < 0 || (reinterpret_cast<uintptr_t>(buf) & 15) || (reinterpret_cast<uintptr_t>(scratch16) & 7)) return SAR_EINVAL;
    if (count == 0) return SAR_OK;
    const int blocks = (int)((count / 4 + 255) / 256 < 2048 ? (count / 4 + 255) / 256 + 1 : 2048);
    unsigned short* s16 = static_cast<unsigned short*>(scratch16);
    hipLaunchKernelGGL(grad_to_bf16_kernel, dim3(blocks), dim3(256), 0, stream, buf, s16, count);
    SAR_CHECK_LAUNCH();
    if (r->allreduce(s16, s16, (size_t)count, NCCL_BFLOAT16, NCCL_SUM, comm, stream) != 0) return SAR_ELAUNCH;
    hipLaunchKernelGGL(grad_from_bf16_kernel, dim3(blocks), dim3(256), 0, stream, s16, buf, count);
    SAR_CHECK_LAUNCH();
    return SAR_OK;
}

extern "C" int sar_comm_destroy(void* comm) {
    const Rccl* r = rccl();
    if (!r) return SAR_ELAUNCH;
    if (!comm) return SAR_EINVAL;
    return r->destroy(comm) == 0 ? SAR_OK : SAR_ELAUNCH;
}
