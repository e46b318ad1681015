// The one exchange step of the path through the C ABI (SURVEY 8b "sgg_allreduce_*", 8e): gradient all-reduce (sum) over RCCL for a host that
// is not torch.  The Python host of this repo keeps using torch.distributed ('nccl' IS RCCL on ROCm: sgg_amd/dist.py) -- same library,
// same collective; these entry points bind it for anything that can call C.  RCCL is looked up at run time (dlopen): libsgg_hip.so has no
// link-time dependency on it, and a process that already carries an RCCL (torch's) gets that one.
#include <dlfcn.h>

#include <cstdint>
#include <cstdlib>
#include <mutex>

#include "common.h"

namespace {
struct Id128 { char b[128]; };     // ncclUniqueId (passed to ncclCommInitRank by value)
struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, Id128, int) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, void*) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
};
Rccl rccl;
std::mutex rccl_mu;

bool load_rccl() {
    std::lock_guard<std::mutex> lk(rccl_mu);
    if (rccl.lib) return rccl.AllReduce != nullptr;
    const char* env = getenv("SGG_RCCL_LIB");
    const char* names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"};
    for (int pass = 0; pass < 2 && !rccl.lib; ++pass)
        for (const char* n : names) {
            if (!n) continue;
            rccl.lib = dlopen(n, pass == 0 ? (RTLD_NOW | RTLD_NOLOAD) : RTLD_NOW);      // first: the copy this process already carries
            if (rccl.lib) break;
        }
    if (!rccl.lib) return false;
    rccl.GetUniqueId = reinterpret_cast<decltype(rccl.GetUniqueId)>(dlsym(rccl.lib, "ncclGetUniqueId"));
    rccl.CommInitRank = reinterpret_cast<decltype(rccl.CommInitRank)>(dlsym(rccl.lib, "ncclCommInitRank"));
    rccl.AllReduce = reinterpret_cast<decltype(rccl.AllReduce)>(dlsym(rccl.lib, "ncclAllReduce"));
    rccl.CommDestroy = reinterpret_cast<decltype(rccl.CommDestroy)>(dlsym(rccl.lib, "ncclCommDestroy"));
    if (!rccl.GetUniqueId || !rccl.CommInitRank || !rccl.AllReduce || !rccl.CommDestroy) rccl.AllReduce = nullptr;
    return rccl.AllReduce != nullptr;
}
}  // namespace

// id: 128 bytes written by rank 0 and handed to every rank by the host's own means (file, socket, MPI ...): ncclGetUniqueId
extern "C" int sgg_allreduce_unique_id(void* id128) {
    if (!id128) return SGG_ERR_ARG;
    if (!load_rccl()) return SGG_ERR_LAUNCH;
    return rccl.GetUniqueId(id128) == 0 ? SGG_OK : SGG_ERR_LAUNCH;
}

// one communicator per process (= per GPU: the current HIP device): ncclCommInitRank
extern "C" int sgg_allreduce_init(const void* id128, int world, int rank, void** comm) {
    if (!id128 || !comm || world < 1 || rank < 0 || rank >= world) return SGG_ERR_ARG;
    if (!load_rccl()) return SGG_ERR_LAUNCH;
    Id128 id;
    __builtin_memcpy(&id, id128, sizeof(id));
    *comm = nullptr;
    return rccl.CommInitRank(comm, world, id, rank) == 0 ? SGG_OK : SGG_ERR_LAUNCH;
}

// buf[n] (SGG_F32 / SGG_BF16 / SGG_F16, in place) = sum over the ranks, on `stream`: the gradient exchange of main.py:116-120 run data-parallel
extern "C" int sgg_allreduce_sum(void* comm, void* buf, int64_t n, int dtype, void* stream) {
    if (n == 0) return SGG_OK;
    if (!comm || !buf || n < 0) return SGG_ERR_ARG;
    if (!sgg_is_dtype(dtype)) return SGG_ERR_DTYPE;
    if (!load_rccl()) return SGG_ERR_LAUNCH;
    const int nccl_dt = dtype == SGG_F32 ? 7 : dtype == SGG_F16 ? 6 : 9;      // ncclFloat32 / ncclFloat16 / ncclBfloat16
    return rccl.AllReduce(buf, buf, (size_t)n, nccl_dt, /* ncclSum */ 0, comm, stream) == 0 ? SGG_OK : SGG_ERR_LAUNCH;
}

extern "C" int sgg_allreduce_destroy(void* comm) {
    if (!comm) return SGG_OK;
    if (!load_rccl()) return SGG_ERR_LAUNCH;
    return rccl.CommDestroy(comm) == 0 ? SGG_OK : SGG_ERR_LAUNCH;
}
