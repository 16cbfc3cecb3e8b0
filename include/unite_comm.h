/* unite_comm.h -- the collective side of the data-parallel hot path (unite_comm_{init,allreduce_bucket,broadcast,destroy})
 * as a C ABI over RCCL, in its own library (unite_amd/lib/libunite_comm.so).  RCCL is BOUND AT RUN TIME (dlopen), never linked: the library
 * uses the librccl.so the process has already mapped -- in a PyTorch process the wheel's copy, the one torch.distributed's "nccl" backend
 * runs on -- so that exactly ONE RCCL runtime lives in the process; a host without one mapped gets the copy named by unite_comm_bind() /
 * UNITE_RCCL_LIB, else the system's (ld.so search path, /opt/rocm/lib).
 *
 * What it replaces in the reference: torch.nn.parallel.DistributedDataParallel's bucketed gradient all-reduce and its initial parameter
 * broadcast (run_stage1.py:809, run_stage2.py DDP wrap, run_stage3.py:942; process-group setup src/utils.py:510-551).  One communicator per
 * process (one process per GPU).  Every call is asynchronous on the given hipStream_t; buffers are borrowed device memory; return 0, a
 * negative UNITE_E* code (unite_hip.h) or 1000 + ncclResult_t.
 *
 * unite_amd.ddp.GradReducer uses torch.distributed (backend "nccl" = the same RCCL) by default and these entry points with
 * UNITE_COMM_NATIVE=1; a host in another language binds them directly. */
#ifndef UNITE_COMM_H
#define UNITE_COMM_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define UNITE_COMM_ID_BYTES 128

/* optional, before any other call: bind RCCL now, preferring a copy already mapped into the process, then `rccl_path` (may be NULL), then
 * UNITE_RCCL_LIB, then the system's.  Every other entry point binds by itself on first use.  0, or -3 if no usable librccl was found. */
int unite_comm_bind(const char* rccl_path);
/* the file the bound RCCL comes from (dladdr of ncclAllReduce), NUL-terminated into out[bytes]; binds if necessary */
int unite_comm_library(char* out, size_t bytes);
/* rank 0 creates the rendezvous id (ncclGetUniqueId) and hands it to the other ranks by any host channel (the launcher's store, a file) */
int unite_comm_unique_id(void* id_out, size_t bytes);
/* collective over all ranks: joins the communicator for the CURRENT HIP device */
int unite_comm_init(int32_t rank, int32_t world, const void* id, size_t bytes);
/* in-place all-reduce of one contiguous gradient bucket: sum, or mean (sum / world) with average != 0.  dtype 0 = f32, 1 = bf16 */
int unite_comm_allreduce_bucket(void* buf, int64_t count, int32_t dtype, int32_t average, void* stream);
/* in-place broadcast of `bytes` bytes from rank `root` (initial parameters: every replica starts from rank 0's weights) */
int unite_comm_broadcast(void* buf, int64_t bytes, int32_t root, void* stream);
int unite_comm_world(void);     /* 0 before unite_comm_init */
int unite_comm_rank(void);
int unite_comm_destroy(void);

#ifdef __cplusplus
}
#endif
#endif
