// Shared device helpers for the gfx950 kernels (wave64, MFMA 16x16x32 bf16, LDS-DMA staging).
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "unite_hip.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define LDS_AS __attribute__((address_space(3)))
#define OOB_OFFSET 0x80000000u     // beyond any buffer we accept (< 2 GiB): buffer loads return 0

#define UNITE_LAUNCH_CHECK()                     \
    do {                                         \
        hipError_t e__ = hipGetLastError();      \
        if (e__ != hipSuccess) return (int)e__;  \
    } while (0)

__device__ __forceinline__ float bf16_to_f32(uint16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {
    // hipcc lowers the cast to v_cvt_pk_bf16_f32 (round-to-nearest-even, NaN preserved)
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(uint16_t, b);
}
// both halves in ONE v_cvt_pk_bf16_f32 (the scalar casts + shift + or form costs four VALU instructions per pair)
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}

// IEEE half pairs (the frozen teacher's f16 residual stream): round-to-nearest-even packs, exact unpacks
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_f16x2(float lo, float hi) {
    // saturating: a stream value beyond the half range becomes +-65504, not inf (an inf would turn its whole row into NaN in the next LayerNorm)
    lo = __builtin_fminf(__builtin_fmaxf(lo, -65504.0f), 65504.0f);      // v_med3_f32
    hi = __builtin_fminf(__builtin_fmaxf(hi, -65504.0f), 65504.0f);
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2_t));
}
__device__ __forceinline__ float unpack_f16_lo(uint32_t w) { return (float)__builtin_bit_cast(f16x2_t, w)[0]; }
__device__ __forceinline__ float unpack_f16_hi(uint32_t w) { return (float)__builtin_bit_cast(f16x2_t, w)[1]; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// erf to ~1.2e-7 absolute (Abramowitz & Stegun 7.1.26 refined: W. J. Cody-style rational is overkill for values that are
// rounded to bf16 next): erf(|x|) = 1 - t (a1 + t (a2 + t (a3 + t (a4 + t a5)))) exp(-x^2), t = 1 / (1 + p |x|).
// Two transcendental issues (v_rcp_f32, v_exp_f32) + 8 plain VALU instead of ocml erff's ~30-instruction branchy polynomial:
// the GEMM epilogues that apply it are VALU-bound (64 lanes / clk / CU against 65536 outputs per 256 x 256 tile).
__device__ __forceinline__ float fast_exp_neg_sq(float x) { return __builtin_amdgcn_exp2f(-1.4426950408889634f * x * x); }
__device__ __forceinline__ float erf_fast(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float y = 1.0f - poly * fast_exp_neg_sq(ax);
    return copysignf(y, x);
}
// Standard normal CDF without a transcendental: Phi(x) = 1/2 + xc Q(xc^2), xc = clamp(x, +-3 sqrt 2), Q of degree 8 (a weighted
// minimax fit of erf(y) / y on y^2 <= 9, rescaled to x = y sqrt 2).  |error| <= 1.2e-5 against the exact CDF in fp32 Horner form
// (the clamp alone costs 1.1e-5: 1 - Phi(4.24)), 0 < Phi < 1 everywhere; x Phi(x) is within 5e-5 of the exact GELU.  The outputs
// it feeds are rounded to bf16 (relative 4e-3) next.  Why: v_rcp_f32 / v_exp_f32 issue at a quarter of the FMA rate and the GEMM
// epilogues that apply GELU / GELU' are VALU-bound -- 68 -> 28 and 109 -> 52 VALU cycles per element (measured: the student's
// fc1 forward 90 -> 78 us, fc2 input gradient 87 -> 83 us; the rest of their distance to the 57-us plain product is the 63 MB of
// saved / re-read pre-activations).  -DUNITE_GELU_POLY=0 builds the erf_fast forms (1.2e-7) instead.
#ifndef UNITE_GELU_POLY
#define UNITE_GELU_POLY 1
#endif
__device__ __forceinline__ float norm_cdf_poly(float x) {
    const float xc = __builtin_fminf(__builtin_fmaxf(x, -4.2426405f), 4.2426405f);      // v_med3_f32
    const float u = xc * xc;
    float q = 5.626682453e-11f;
    q = __builtin_fmaf(q, u, -5.371804335e-09f);
    q = __builtin_fmaf(q, u, 2.268276091e-07f);
    q = __builtin_fmaf(q, u, -5.646181762e-06f);
    q = __builtin_fmaf(q, u, 9.359031537e-05f);
    q = __builtin_fmaf(q, u, -1.109398669e-03f);
    q = __builtin_fmaf(q, u, 9.818114340e-03f);
    q = __builtin_fmaf(q, u, -6.634691358e-02f);
    q = __builtin_fmaf(q, u, 3.989031017e-01f);
    return __builtin_fmaf(xc, q, 0.5f);
}
__device__ __forceinline__ float norm_cdf(float x) {
    return UNITE_GELU_POLY ? norm_cdf_poly(x) : 0.5f * (1.0f + erf_fast(x * 0.70710678118654752f));
}
__device__ __forceinline__ float gelu_erf(float x) { return x * norm_cdf(x); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
    const float pdf = 0.3989422804014327f * __builtin_amdgcn_exp2f(-0.72134752044448170f * x * x);
    return __builtin_fmaf(x, pdf, norm_cdf(x));
}
// gelu_erf(x) and gelu_erf'(x) from ONE evaluation of the normal CDF: the forward fc1 epilogue saves the derivative (UNITE_ACT_GELU_DSAVE), so
// that the backward epilogue is a decode + multiply (UNITE_ACT_MULAUX) instead of ~25 VALU instructions per element.
// The derivative lies in [-0.129, 1.129]; it is saved as a 16-bit fixed-point number q = round(65535 (d + 0.25) / 2) (v_cvt_pknorm_u16_f32):
// absolute error <= 1.6e-5, where bf16 would carry 2e-3 relative and recomputing from the bf16-rounded z carries 5e-4 (tools/gelu_dsave_error.py).
__device__ __forceinline__ void gelu_erf_both(float x, float& y, float& dy) {
    const float c = norm_cdf(x);
    const float pdf = 0.3989422804014327f * __builtin_amdgcn_exp2f(-0.72134752044448170f * x * x);
    y = x * c;
    dy = __builtin_fmaf(x, pdf, c);
}
__device__ __forceinline__ uint32_t pack_dgelu_x2(float d0, float d1) {
    const auto q = __builtin_amdgcn_cvt_pknorm_u16(__builtin_fmaf(d0, 0.5f, 0.125f), __builtin_fmaf(d1, 0.5f, 0.125f));
    return __builtin_bit_cast(uint32_t, q);
}
__device__ __forceinline__ float unpack_dgelu_lo(uint32_t q) { return __builtin_fmaf((float)(q & 0xFFFFu), 2.0f / 65535.0f, -0.25f); }
__device__ __forceinline__ float unpack_dgelu_hi(uint32_t q) { return __builtin_fmaf((float)(q >> 16), 2.0f / 65535.0f, -0.25f); }
// x * sigmoid(1.702 x) with v_exp_f32 + v_rcp_f32 (1 ulp) instead of the IEEE division sequence (~10 instructions)
__device__ __forceinline__ float quick_gelu(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.4554669595930156f * x)); }

// 16x16x32 bf16 MFMA: A[row l&15][k = 8(l>>4)+j], B[k = 8(l>>4)+j][col l&15], C[row 4(l>>4)+r][col l&15]
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// ds_read_b64_tr_b16 through inline asm.  With the builtin, hipcc cannot tell that the read does not alias the LDS-DMA writes in
// flight and puts `s_waitcnt vmcnt(0)` in front of the first transposing read of EVERY phase: the counted vmcnt(8) pipeline of the
// k-strided operand layouts (dgrad, wgrad) was drained four times per K-tile.  The asm read is invisible to the compiler's
// wait-count tracking, so every phase that uses it ends its reads with lds_tr_fence() (lgkmcnt(0) + sched_barrier, guide rule 18).
__device__ __forceinline__ u32x2 lds_read_tr16_raw(const char* p) {
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"((uint32_t)(uintptr_t)(LDS_AS const char*)p));
    return v;
}
__device__ __forceinline__ bf16x8 tr_join(u32x2 lo, u32x2 hi) {
    const u32x4 v = {lo[0], lo[1], hi[0], hi[1]};
    return __builtin_bit_cast(bf16x8, v);
}
// 16-byte LDS read the compiler does not track either (same purpose: a register set filled for the NEXT group of MFMAs while the
// current group runs; with the tracked read hipcc puts lgkmcnt(0) in front of the current group).  addr = LDS byte address,
// OFF = immediate offset (< 64 KiB).  Fence with lds_tr_fence<true>() before the first use.
template <int OFF>
__device__ __forceinline__ bf16x8 lds_read_b128_raw(uint32_t addr) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return __builtin_bit_cast(bf16x8, v);
}
// the transposing read in the same untracked form, LDS byte address + immediate offset
template <int OFF>
__device__ __forceinline__ u32x2 lds_read_tr16_raw_off(uint32_t addr) {
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
__device__ __forceinline__ uint32_t lds_address(const void* p) { return (uint32_t)(uintptr_t)(LDS_AS const char*)p; }
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
template <bool ANY_TR>
__device__ __forceinline__ void lds_tr_fence() {
    if (ANY_TR) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ---- inter-workgroup hand-off inside one launch (cdna_hip_programming.md Guideline 16, counter form) -------------------------------
// Every workgroup of a group of `expected` calls this AFTER its global stores; it returns true in every thread of the workgroup that
// arrived LAST, and that workgroup may then read what the others stored with plain loads.  Producer side: every wave drains its stores
// (vmcnt(0)), workgroup barrier, one lane: agent-scope release, drained again by hand (hipcc may drop the fence's own wait), relaxed
// agent-scope ticket.  Consumer side (the last arriver only): agent-scope acquire (invalidates this CU's L1), drained, barrier.
// Nobody waits for anybody: no spin, no residency assumption.  The counter returns to zero (the last arriver resets it), so a
// workspace header that was zero before the first launch stays usable for ever; `lds_word` is any 4 idle bytes of the ONE LDS array.
// RELEASE = false: the caller stored EVERY handed-off byte write-through (sc1 / agent-scope relaxed atomic stores), so no L2 write-back is
// needed (a release with tens of KB freshly dirtied per workgroup costs several us: MI355X_MICROARCH.md, publish-large).
template <bool RELEASE = true>
__device__ __forceinline__ bool arrive_last(uint32_t* counter, uint32_t expected, volatile uint32_t* lds_word) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        if (RELEASE) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const uint32_t t = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool last = t + 1u == expected;
        if (last) {
            __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        *lds_word = last ? 1u : 0u;
    }
    __syncthreads();
    return *lds_word != 0u;
}

// ds_read_b64_tr_b16: within each 16-lane group, lane 4q+p supplies the address of row q, columns 4p..4p+3 of a
// 4x16 block of 16-bit elements; lane i receives column i of the four rows (row q in element q).
__device__ __forceinline__ s16x4 lds_read_tr16(const void* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)p);
}
