// Shared device helpers of the attention kernels (LDS image, fragment loaders, staging).  See attention.hip.
#pragma once
#include "common.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f;

__device__ __forceinline__ int row_addr(int row, int c16) {   // byte offset of 16-B chunk c16 of `row`
    return row * 128 + (((((c16 >> 1) ^ ((row >> 1) & 3)) << 1) | (c16 & 1)) << 4);
}
__device__ __forceinline__ bf16x8 row_frag(const char* tile, int row0, int ks, int lane) {
    return *(const bf16x8*)(tile + row_addr(row0 + (lane & 15), ks * 4 + (lane >> 4)));
}
// transposed fragment: element j = X[rows32 + (j<4 ? 4G+j : 16+4G+j-4)][16 dt + (l&15)]
__device__ __forceinline__ bf16x8 tr_frag(const char* tile, int rows32, int dt, int lane) {
    const int G = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const int r_lo = rows32 + 4 * G + q, r_hi = r_lo + 16;
    // the two 8-byte halves are joined as dwords: element-wise construction of the 8 x 16-bit vector costs ~16 SDWA / shift ops
    const u32x2 lo = __builtin_bit_cast(u32x2, lds_read_tr16(tile + r_lo * 128 + ((dt ^ ((r_lo >> 1) & 3)) << 5) + 8 * p));
    const u32x2 hi = __builtin_bit_cast(u32x2, lds_read_tr16(tile + r_hi * 128 + ((dt ^ ((r_hi >> 1) & 3)) << 5) + 8 * p));
    const u32x4 v = {lo[0], lo[1], hi[0], hi[1]};
    return __builtin_bit_cast(bf16x8, v);
}
// plain fmaxf nest: the attention objects are built with -fno-honor-nans (Makefile), which drops the canonicalising v_max x,x
// in front of every maxnum and lets the compiler form v_max3_f32.  (Inline asm is not an option for values that come straight
// out of an MFMA: the hazard recogniser does not see into it.)
__device__ __forceinline__ float max3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
// the same fragment through the asm read (common.h): for loops that keep LDS-DMA tiles in flight -- with the builtin hipcc drains
// the DMA queue (vmcnt(0)) in front of the read.  The caller ends its group of reads with lds_tr_fence<true>() before the MFMAs.
__device__ __forceinline__ bf16x8 tr_frag_raw(const char* tile, int rows32, int dt, int lane) {
    const int G = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const int r_lo = rows32 + 4 * G + q, r_hi = r_lo + 16;
    return tr_join(lds_read_tr16_raw(tile + r_lo * 128 + ((dt ^ ((r_lo >> 1) & 3)) << 5) + 8 * p),
                   lds_read_tr16_raw(tile + r_hi * 128 + ((dt ^ ((r_hi >> 1) & 3)) << 5) + 8 * p));
}
__device__ __forceinline__ bf16x8 pack_pair(f32x4 a, f32x4 b) {
    u32x4 w = {pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3]), pack_bf16x2(b[0], b[1]), pack_bf16x2(b[2], b[3])};
    return __builtin_bit_cast(bf16x8, w);
}
// stage rows [0, rows_padded) of one head slice (element offset base, row stride ld) into the swizzled image
__device__ __forceinline__ void stage_rows(__amdgpu_buffer_rsrc_t rs, char* tile, int rows_padded, int n_valid, uint32_t base, int ld,
                                           int wave, int nwaves, int lane) {
    for (int it = wave; it < rows_padded / 8; it += nwaves) {
        const int r = it * 8 + (lane >> 3), p = lane & 7;
        const int lc16 = ((((p >> 1) ^ ((r >> 1) & 3)) << 1) | (p & 1));
        const uint32_t voff = (r < n_valid) ? (base + (uint32_t)r * ld + lc16 * 8) * 2u : OOB_OFFSET;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (LDS_AS void*)(tile + it * 1024), 16, voff, 0, 0, 0);
    }
}
__device__ __forceinline__ float group_max(float v) { v = fmaxf(v, __shfl_xor(v, 16, 64)); return fmaxf(v, __shfl_xor(v, 32, 64)); }
__device__ __forceinline__ float group_sum(float v) { v += __shfl_xor(v, 16, 64); return v + __shfl_xor(v, 32, 64); }
__device__ __forceinline__ void store_bf16x4(uint16_t* p, f32x4 v, float s) {
    *(u32x2*)p = (u32x2){pack_bf16x2(v[0] * s, v[1] * s), pack_bf16x2(v[2] * s, v[3] * s)};
}


}  // namespace
