// RCCL communicator behind the C ABI of include/unite_comm.h (one communicator per process; one process per GPU).
//
// RCCL is bound at RUN time, not linked: a PyTorch process already has a librccl.so mapped (the wheel's own copy, which torch.distributed's
// "nccl" backend uses), and a second copy from /opt/rocm/lib beside it means two RCCL runtimes in one process -- two sets of proxy threads
// and internal streams, two ROCm versions (measured in round 3: the one-rank rehearsal took 27.5 instead of 20.4 ms per step through the
// linked copy).  Resolution order, first hit wins:
//   1. the path given to unite_comm_bind() (unite_amd._lib.load_comm passes <torch>/lib/librccl.so) or UNITE_RCCL_LIB -- but if a library of
//      that soname is ALREADY mapped, dlopen returns the mapped one, which is the point;
//   2. whatever librccl.so / librccl.so.1 the process has mapped (RTLD_NOLOAD);
//   3. the system search path (a host without PyTorch: /opt/rocm/lib through ld.so).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>      // types and prototypes only: nothing of librccl is linked
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "unite_comm.h"

namespace {
ncclComm_t g_comm = nullptr;
int g_world = 0, g_rank = -1;
inline int rc_of(ncclResult_t r) { return r == ncclSuccess ? 0 : 1000 + (int)r; }

struct Rccl {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    char path[1024] = {0};
} g_rccl;

bool take(void* h) {
    if (!h) return false;
    Rccl r;
    r.handle = h;
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(h, "ncclCommInitRank");
    r.AllReduce = (decltype(r.AllReduce))dlsym(h, "ncclAllReduce");
    r.Broadcast = (decltype(r.Broadcast))dlsym(h, "ncclBroadcast");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
    if (!r.GetUniqueId || !r.CommInitRank || !r.AllReduce || !r.Broadcast || !r.CommDestroy) return false;
    Dl_info info;
    if (dladdr((void*)r.AllReduce, &info) && info.dli_fname) strncpy(r.path, info.dli_fname, sizeof(r.path) - 1);
    g_rccl = r;
    return true;
}

// 0 on success.  `path` may be null.
int bind_rccl(const char* path) {
    if (g_rccl.handle) return 0;
    static const char* names[] = {"librccl.so", "librccl.so.1"};
    for (const char* n : names)                                   // a copy the process has already mapped (the torch wheel's)
        if (take(dlopen(n, RTLD_NOLOAD | RTLD_NOW))) return 0;
    const char* env = getenv("UNITE_RCCL_LIB");
    if (path && *path && take(dlopen(path, RTLD_NOW | RTLD_GLOBAL))) return 0;
    if (env && *env && take(dlopen(env, RTLD_NOW | RTLD_GLOBAL))) return 0;
    for (const char* n : names)
        if (take(dlopen(n, RTLD_NOW | RTLD_GLOBAL))) return 0;
    if (take(dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL))) return 0;
    return -3;
}
}  // namespace

extern "C" int unite_comm_bind(const char* rccl_path) { return bind_rccl(rccl_path); }

extern "C" int unite_comm_library(char* out, size_t bytes) {
    if (!out || bytes == 0) return -1;
    if (bind_rccl(nullptr) != 0) return -3;
    strncpy(out, g_rccl.path, bytes - 1);
    out[bytes - 1] = 0;
    return 0;
}

extern "C" int unite_comm_unique_id(void* id_out, size_t bytes) {
    if (!id_out || bytes < sizeof(ncclUniqueId) || sizeof(ncclUniqueId) > UNITE_COMM_ID_BYTES) return -1;
    if (bind_rccl(nullptr) != 0) return -3;
    ncclUniqueId id;
    const ncclResult_t r = g_rccl.GetUniqueId(&id);
    if (r != ncclSuccess) return rc_of(r);
    memset(id_out, 0, bytes);
    memcpy(id_out, &id, sizeof(id));
    return 0;
}

extern "C" int unite_comm_init(int32_t rank, int32_t world, const void* idp, size_t bytes) {
    if (g_comm || !idp || bytes < sizeof(ncclUniqueId) || world < 1 || rank < 0 || rank >= world) return -1;
    if (bind_rccl(nullptr) != 0) return -3;
    ncclUniqueId id;
    memcpy(&id, idp, sizeof(id));
    const ncclResult_t r = g_rccl.CommInitRank(&g_comm, world, id, rank);
    if (r != ncclSuccess) { g_comm = nullptr; return rc_of(r); }
    g_world = world;
    g_rank = rank;
    return 0;
}

extern "C" int unite_comm_allreduce_bucket(void* buf, int64_t count, int32_t dtype, int32_t average, void* stream) {
    if (!g_comm || !buf || count <= 0 || (dtype != 0 && dtype != 1)) return -1;
    return rc_of(g_rccl.AllReduce(buf, buf, (size_t)count, dtype == 0 ? ncclFloat32 : ncclBfloat16, average ? ncclAvg : ncclSum, g_comm,
                                  (hipStream_t)stream));
}

extern "C" int unite_comm_broadcast(void* buf, int64_t bytes, int32_t root, void* stream) {
    if (!g_comm || !buf || bytes <= 0 || root < 0 || root >= g_world) return -1;
    return rc_of(g_rccl.Broadcast(buf, buf, (size_t)bytes, ncclUint8, root, g_comm, (hipStream_t)stream));
}

extern "C" int unite_comm_world(void) { return g_world; }
extern "C" int unite_comm_rank(void) { return g_rank; }

extern "C" int unite_comm_destroy(void) {
    if (!g_comm) return 0;
    const ncclResult_t r = g_rccl.CommDestroy(g_comm);
    g_comm = nullptr;
    g_world = 0;
    g_rank = -1;
    return rc_of(r);
}
