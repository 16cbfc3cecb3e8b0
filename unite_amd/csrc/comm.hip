// RCCL communicator behind the C ABI of include/unite_comm.h (one communicator per process; one process per GPU).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <string.h>
#include "unite_comm.h"

namespace {
ncclComm_t g_comm = nullptr;
int g_world = 0, g_rank = -1;
inline int rc_of(ncclResult_t r) { return r == ncclSuccess ? 0 : 1000 + (int)r; }
}  // namespace

extern "C" int unite_comm_unique_id(void* id_out, size_t bytes) {
    if (!id_out || bytes < sizeof(ncclUniqueId) || sizeof(ncclUniqueId) > UNITE_COMM_ID_BYTES) return -1;
    ncclUniqueId id;
    const ncclResult_t r = ncclGetUniqueId(&id);
    if (r != ncclSuccess) return rc_of(r);
    memset(id_out, 0, bytes);
    memcpy(id_out, &id, sizeof(id));
    return 0;
}

extern "C" int unite_comm_init(int32_t rank, int32_t world, const void* idp, size_t bytes) {
    if (g_comm || !idp || bytes < sizeof(ncclUniqueId) || world < 1 || rank < 0 || rank >= world) return -1;
    ncclUniqueId id;
    memcpy(&id, idp, sizeof(id));
    const ncclResult_t r = ncclCommInitRank(&g_comm, world, id, rank);
    if (r != ncclSuccess) { g_comm = nullptr; return rc_of(r); }
    g_world = world;
    g_rank = rank;
    return 0;
}

extern "C" int unite_comm_allreduce_bucket(void* buf, int64_t count, int32_t dtype, int32_t average, void* stream) {
    if (!g_comm || !buf || count <= 0 || (dtype != 0 && dtype != 1)) return -1;
    return rc_of(ncclAllReduce(buf, buf, (size_t)count, dtype == 0 ? ncclFloat32 : ncclBfloat16, average ? ncclAvg : ncclSum, g_comm,
                               (hipStream_t)stream));
}

extern "C" int unite_comm_broadcast(void* buf, int64_t bytes, int32_t root, void* stream) {
    if (!g_comm || !buf || bytes <= 0 || root < 0 || root >= g_world) return -1;
    return rc_of(ncclBroadcast(buf, buf, (size_t)bytes, ncclUint8, root, g_comm, (hipStream_t)stream));
}

extern "C" int unite_comm_world(void) { return g_world; }
extern "C" int unite_comm_rank(void) { return g_rank; }

extern "C" int unite_comm_destroy(void) {
    if (!g_comm) return 0;
    const ncclResult_t r = ncclCommDestroy(g_comm);
    g_comm = nullptr;
    g_world = 0;
    g_rank = -1;
    return rc_of(r);
}
