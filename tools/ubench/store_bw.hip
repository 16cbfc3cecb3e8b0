// micro-benchmark: per-CU global store throughput vs number of active workgroups (one per CU)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
__global__ __launch_bounds__(512) void store_kernel(u32x4* out, size_t per_wg_vec, int iters) {
    u32x4 v = {threadIdx.x, blockIdx.x, 3u, 4u};
    u32x4* base = out + (size_t)blockIdx.x * per_wg_vec;
    for (int it = 0; it < iters; ++it)
        for (size_t i = threadIdx.x; i < per_wg_vec; i += 512) base[i] = v;
}
int main() {
    const size_t per_wg = 128 * 1024;          // bytes per workgroup per iteration (one 256x256 bf16 tile)
    for (int nwg : {1, 16, 64, 128, 256, 512, 1024}) {
        for (int fresh : {0, 1}) {
            const int iters = fresh ? 1 : 64;
            const size_t rounds = fresh ? 64 : 1;   // fresh: every launch writes new memory (no L2 rewrite)
            u32x4* buf;
            size_t total = per_wg * nwg * rounds;
            hipMalloc(&buf, total);
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            store_kernel<<<nwg, 512>>>(buf, per_wg / 16, 1);
            hipDeviceSynchronize();
            hipEventRecord(a);
            for (size_t r = 0; r < rounds; ++r)
                store_kernel<<<nwg, 512>>>(buf + r * (per_wg * nwg / 16), per_wg / 16, iters);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            double bytes = (double)per_wg * nwg * 64;
            printf("nwg %4d %s: %8.1f us  total %7.1f GB/s  per-WG %6.2f GB/s\n", nwg, fresh ? "fresh  " : "rewrite", ms * 1e3, bytes / ms / 1e6, bytes / ms / 1e6 / nwg);
            hipFree(buf);
        }
    }
    return 0;
}
