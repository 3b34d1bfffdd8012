// Gradient all-reduce of the data-parallel step over RCCL (xGMI), behind the C-ABI: what keras.utils.multi_gpu_model's gradient
// aggregation is for the reference (model.py:193-194), one process per GPU here.  RCCL is bound at run time (dlopen: the copy this
// process already has -- PyTorch ships one -- or /opt/rocm's), so the library loads on hosts without it and fails loudly only when a
// communicator is asked for.  No transfers are implemented here: ncclAllReduce does the work; this file is the boundary.
#include "common.h"
#include <dlfcn.h>
#include <string.h>

namespace {

struct Uid { char b[128]; };                                   // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
typedef int (*get_uid_t)(Uid*);
typedef int (*init_rank_t)(void**, int, Uid, int);
typedef int (*allreduce_t)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*destroy_t)(void*);
struct Rccl { void* h; get_uid_t get_uid; init_rank_t init_rank; allreduce_t allreduce; destroy_t destroy; };

const Rccl* rccl() {
    static Rccl r{};
    static bool tried = false;
    if (!tried) {
        tried = true;
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
    }
    return r.h ? &r : nullptr;
}
constexpr int NCCL_FLOAT32 = 7, NCCL_SUM = 0;                   // rccl.h: ncclFloat32, ncclSum

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

extern "C" int sar_comm_destroy(void* comm) {
    const Rccl* r = rccl();
    if (!r) return SAR_ELAUNCH;
    if (!comm) return SAR_EINVAL;
    return r->destroy(comm) == 0 ? SAR_OK : SAR_ELAUNCH;
}
