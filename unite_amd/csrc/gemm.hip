// bf16 MFMA GEMM with fused epilogues for gfx950 (MI355X).
//
//   out[M,N] = epilogue( op(A)[M,K] * op(B)[K,N] )
//
// One 256-thread workgroup (4 waves as 2x2) computes a 128x128 tile; each wave owns 64x64 =
// 4x4 v_mfma_f32_16x16x32_bf16 accumulators.  Operand tiles (128 x 64 k) are staged HBM -> LDS by
// LDS-DMA (buffer_load_dwordx4 ... lds, 1 KiB per wave-instruction) into a double buffer; out-of-range
// rows / k are handled by the buffer descriptor's range check (they read as zero), so every shape
// with 16-byte rows is legal.  Both operand layouts are supported without a transposed copy in HBM:
//   * k-contiguous operand ([rows][k], 128-B LDS rows, XOR-swizzled 16-B chunks, ds_read_b128)
//   * k-strided operand   ([k][cols], 256-B LDS rows, swizzled, ds_read_b64_tr_b16 transposing reads)
// so forward (x W^T), dgrad (dy W) and wgrad (dy^T x) all run on this one kernel.
// The LDS image is lane-linear (LDS-DMA cannot scatter), so the swizzle is applied to the per-lane
// SOURCE address and again on the read (cdna_hip_programming.md rule 21).
// The accumulators leave through LDS so that the epilogue (bias, GELU / QuickGELU / GELU', stochastic-depth
// row scale, residual add, bf16 + f32 outputs) reads and writes 16 B (bf16) / 32 B (f32) per lane.
#include "common.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include <utility>
#include <vector>

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;          // 16 KiB per operand tile (both layouts)
constexpr int STAGE_BYTES = 2 * TILE_BYTES;      // A + B
constexpr int CS_LD = 132;                       // f32 row stride of the epilogue image (conflict-free writes)
constexpr int LDS_BYTES = 2 * STAGE_BYTES;       // 64 KiB  (epilogue image: 64 x 132 x 4 = 33 KiB, reuses it)

constexpr int MAX_GROUP = 4;
struct Params {
    unite_gemm_args a;            // the problem (group 0 of a grouped launch)
    uint32_t a_bytes, b_bytes;
    // grouped launch (unite_gemm_bf16_grouped, deep 128^2 kernel only): problems 1..ngroups-1 and the running tile counts;
    // workgroup `lin` (after the XCD remap) works on problem gi with tile_end[gi-1] <= lin < tile_end[gi]
    unite_gemm_args more[MAX_GROUP - 1];
    uint32_t more_a_bytes[MAX_GROUP - 1], more_b_bytes[MAX_GROUP - 1];
    int32_t tile_end[MAX_GROUP];
    int32_t ngroups;
    int32_t splitk, k_chunk;      // K is cut into `splitk` slices of k_chunk (multiple of BK); slice s writes slab s
    float* slab;                  // f32 [splitk][M][N] partial products (split-K only, deep kernels only): the slice of a tile that finishes
                                  // LAST sums the slabs in slice order (+ the old output if accumulate) and writes the tile
    uint32_t* counters;           // one arrival counter per output tile (zero outside a launch)
    float* rowsum_slab;           // f32 [splitk][M] per-slice row sums of op(A) (rowsum_a_out with split-K)
    int32_t debug_skip;           // timing experiments only (UNITE_GEMM_DEBUG_SKIP=1: no epilogue; 2: no global stores).  Keep such switches OUT of the staging loop: two more of them there (round 4's
                                  // epilogue anatomy) cost the plain 256^2 kernel 12 registers (234 -> 246), i.e. the room beside it on a SIMD
    float* colsum_partial;        // per-tile-row column sums of the stored output (deep kernels only), or NULL
    int32_t nt_store;             // stream the output past the L2 (non-temporal stores) so that it does not evict the operand panels
    int32_t separate_reduce;      // split-K: the slices only write their slabs (and row-sum slabs); splitk_finish_kernel sums them afterwards
    int32_t group_rows;           // deep kernels: tiles run column-major inside groups of this many tile rows (<= 1: plain row-major order), so
                                  // that the workgroups resident on an XCD share few A panels AND few B panels of its 4-MiB L2
};

__device__ __forceinline__ int swz256(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

// Stage one 128 x 64 operand tile.  TR == false: memory is [rows][k] (ld elements per row);
// TR == true: memory is [k][rows].  `r0` = first row (m or n) of the tile, `k0` = first k.
template <bool TR>
__device__ __forceinline__ void stage_tile(__amdgpu_buffer_rsrc_t rs, char* lds_tile, int r0, int k0, int R, int K,
                                           int ld, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int it = wave * 4 + i;             // 16 wave-instructions of 1 KiB cover the tile
        uint32_t voff;
        if (!TR) {
            const int r = it * 8 + (lane >> 3);  // tile row; 8 rows x 128 B per instruction
            const int lc = (lane & 7) ^ (r & 7); // logical 16-B chunk this lane's LDS slot must hold
            const int gr = r0 + r, gk = k0 + lc * 8;
            voff = (gr < R && gk < K) ? (uint32_t)(gr * ld + gk) * 2u : OOB_OFFSET;
        } else {
            const int kr = it * 4 + (lane >> 4); // tile k-row; 4 rows x 256 B per instruction
            const int lc = (lane & 15) ^ swz256(kr);
            const int gk = k0 + kr, gr = r0 + lc * 8;
            voff = (gk < K && gr < R) ? (uint32_t)(gk * ld + gr) * 2u : OOB_OFFSET;
        }
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (LDS_AS void*)(lds_tile + it * 1024), 16, voff, 0, 0, 0);
    }
}

// Fragment of a 16-row block for k-step `ks` (32 k): element j = X[row0 + (l&15)][32 ks + 8 (l>>4) + j].
template <bool TR>
__device__ __forceinline__ bf16x8 load_frag(const char* lds_tile, int row0, int ks, int lane) {
    if (!TR) {
        const int row = row0 + (lane & 15);
        const int lc = ks * 4 + (lane >> 4);
        return *(const bf16x8*)(lds_tile + row * 128 + ((lc ^ (row & 7)) << 4));
    } else {
        const int G = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
        const int ch = (row0 >> 3) + (p >> 1);
        const int k_lo = ks * 32 + 8 * G + q, k_hi = k_lo + 4;
        return tr_join(lds_read_tr16_raw(lds_tile + 256 * k_lo + ((ch ^ swz256(k_lo)) << 4) + 8 * (p & 1)),
                       lds_read_tr16_raw(lds_tile + 256 * k_hi + ((ch ^ swz256(k_hi)) << 4) + 8 * (p & 1)));
    }
}

// Feature word of an epilogue: which of its run-time tests are true for a launch.  The epilogue is a template on (KNOWN, VALUE): a test whose bit
// is in KNOWN is decided at compile time from VALUE, the others stay run-time tests (KNOWN = 0: the generic form).  The 256^2 kernel computes
// the word once per workgroup and calls a fully-known instantiation for the combinations the training step uses: left as run-time tests inside
// the chunk loops they were a dozen scalar branches per 8-column chunk (round 4: ~750 per workgroup in the bf16 form, profiles/r04_clock_notes.txt).
enum : uint32_t { EF_ACT = 7u, EF_SPLIT = 8u, EF_AUXOUT = 16u, EF_SCALE = 32u, EF_RES = 64u, EF_RESBF = 128u, EF_SKIP2 = 256u, EF_OUTF32 = 512u,
                  EF_ACCUM = 1024u, EF_NT2 = 2048u, EF_NT = 4096u, EF_COPY = 8192u, EF_CSUM = 16384u,
                  EF_F16 = 65536u,           // the 16-bit residual and a 16-bit output are IEEE half (residual_bf16 == 2), not bf16
                  EF_ALL = 32767u | 65536u };
__device__ __forceinline__ uint32_t epilogue_features(const Params& p, const unite_gemm_args& g, bool csum) {
    return (uint32_t)g.act | (p.splitk > 1 ? EF_SPLIT : 0u) | (g.aux_out ? EF_AUXOUT : 0u) | (g.row_scale ? EF_SCALE : 0u) | (g.residual ? EF_RES : 0u) |
           (g.residual_bf16 ? EF_RESBF : 0u) | (g.residual_bf16 == 2 ? EF_F16 : 0u) | (p.debug_skip == 2 ? EF_SKIP2 : 0u) | (g.out_f32 ? EF_OUTF32 : 0u) | (g.accumulate ? EF_ACCUM : 0u) |
           (p.nt_store == 2 ? EF_NT2 : 0u) | (p.nt_store ? EF_NT : 0u) | (g.out_bf16_copy ? EF_COPY : 0u) | (csum ? EF_CSUM : 0u);
}
#define UNITE_EF(bit, runtime) ((KNOWN & (bit)) ? ((VALUE & (bit)) != 0u) : (runtime))
#define UNITE_EF_ACT(a) ((KNOWN & EF_ACT) ? ((VALUE & EF_ACT) == (uint32_t)(a)) : (g.act == (a)))

// Epilogue for 8 consecutive output columns of one row (f32 accumulators v0|v1): see unite_hip.h for the order of operations.
// The bias chunk (b0|b1) is loaded by the caller ONCE per tile, ahead of the stores: vmcnt retires in issue order, so a
// load issued between the stores of two passes could only be waited for together with every store before it.
template <uint32_t KNOWN = 0u, uint32_t VALUE = 0u>
__device__ __forceinline__ void epilogue_chunk(const Params& p, const unite_gemm_args& g, int slice, int gm, int gn, f32x4 v0, f32x4 v1,
                                               f32x4 b0, f32x4 b1, float* csum = nullptr) {
    if (UNITE_EF(EF_SPLIT, p.splitk > 1)) {      // raw partial product -> slab, write-through (sc1): the tile's last slice to finish reads it in this launch
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.slab + (size_t)slice * g.M * g.N), 0, 0x7FFFFFF0, 0x00020000);
        const uint32_t off = (uint32_t)(((size_t)gm * g.N + gn) * 4);
        __builtin_amdgcn_raw_buffer_store_b128((u32x4){__float_as_uint(v0[0]), __float_as_uint(v0[1]), __float_as_uint(v0[2]), __float_as_uint(v0[3])}, rs, off, 0, 16);
        __builtin_amdgcn_raw_buffer_store_b128((u32x4){__float_as_uint(v1[0]), __float_as_uint(v1[1]), __float_as_uint(v1[2]), __float_as_uint(v1[3])}, rs, off + 16, 0, 16);
        return;
    }
    float v[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = v0[e]; v[4 + e] = v1[e]; }
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] += b0[e]; v[4 + e] += b1[e]; }
    if (UNITE_EF_ACT(UNITE_ACT_GELU)) {
        if (UNITE_EF(EF_AUXOUT, g.aux_out != nullptr)) {
            u32x4 z = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
            *(u32x4*)((uint16_t*)g.aux_out + (size_t)gm * g.ld_aux_out + gn) = z;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = gelu_erf(v[e]);
    } else if (UNITE_EF_ACT(UNITE_ACT_QUICKGELU)) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = quick_gelu(v[e]);
    } else if (UNITE_EF_ACT(UNITE_ACT_DGELU)) {
        const u32x4 z = *(const u32x4*)((const uint16_t*)g.aux_in + (size_t)gm * g.ld_aux_in + gn);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[2 * e] *= gelu_erf_grad(__uint_as_float(z[e] << 16));
            v[2 * e + 1] *= gelu_erf_grad(__uint_as_float(z[e] & 0xFFFF0000u));
        }
    } else if (UNITE_EF_ACT(UNITE_ACT_GELU_DSAVE)) {      // the derivative is saved (16-bit fixed point) instead of the pre-activation: one CDF evaluation serves both
        float d[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) gelu_erf_both(v[e], v[e], d[e]);
        *(u32x4*)((uint16_t*)g.aux_out + (size_t)gm * g.ld_aux_out + gn) =
            (u32x4){pack_dgelu_x2(d[0], d[1]), pack_dgelu_x2(d[2], d[3]), pack_dgelu_x2(d[4], d[5]), pack_dgelu_x2(d[6], d[7])};
    } else if (UNITE_EF_ACT(UNITE_ACT_MULAUX)) {
        const u32x4 z = *(const u32x4*)((const uint16_t*)g.aux_in + (size_t)gm * g.ld_aux_in + gn);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[2 * e] *= unpack_dgelu_lo(z[e]);
            v[2 * e + 1] *= unpack_dgelu_hi(z[e]);
        }
    }
    if (UNITE_EF(EF_SCALE, g.row_scale != nullptr)) {
        const float sc = g.row_scale[gm / g.rows_per_scale];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= sc;
    }
    if (UNITE_EF(EF_RES, g.residual != nullptr)) {
        if (UNITE_EF(EF_RESBF, g.residual_bf16 != 0)) {
            const u32x4 r = *(const u32x4*)((const uint16_t*)g.residual + (size_t)gm * g.ldr + gn);
            if (UNITE_EF(EF_F16, g.residual_bf16 == 2)) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[2 * e] += unpack_f16_lo(r[e]);
                    v[2 * e + 1] += unpack_f16_hi(r[e]);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[2 * e] += __uint_as_float(r[e] << 16);
                    v[2 * e + 1] += __uint_as_float(r[e] & 0xFFFF0000u);
                }
            }
        } else {
            const float* rp = (const float*)g.residual + (size_t)gm * g.ldr + gn;
            const f32x4 r0 = *(const f32x4*)rp, r1 = *(const f32x4*)(rp + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] += r0[e]; v[4 + e] += r1[e]; }
        }
    }
    if (UNITE_EF(EF_SKIP2, p.debug_skip == 2)) {      // timing experiment: everything but the stores
        asm volatile("" ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]));
        return;
    }
    if (UNITE_EF(EF_OUTF32, g.out_f32 != 0)) {
        float* op = (float*)g.out + (size_t)gm * g.ldc + gn;
        if (UNITE_EF(EF_ACCUM, g.accumulate != 0)) {
            const f32x4 o0 = *(const f32x4*)op, o1 = *(const f32x4*)(op + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] += o0[e]; v[4 + e] += o1[e]; }
        }
        if (UNITE_EF(EF_NT2, p.nt_store == 2)) {       // write-through, line dropped from the XCD's L2 (sc1): the output must not evict A/B panels
            const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)g.out, 0, 0x7FFFFFF0, 0x00020000);
            const uint32_t off = (uint32_t)(((size_t)gm * g.ldc + gn) * 4);
            __builtin_amdgcn_raw_buffer_store_b128((u32x4){__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])}, ro, off, 0, 16);
            __builtin_amdgcn_raw_buffer_store_b128((u32x4){__float_as_uint(v[4]), __float_as_uint(v[5]), __float_as_uint(v[6]), __float_as_uint(v[7])}, ro, off + 16, 0, 16);
        } else if (UNITE_EF(EF_NT, p.nt_store != 0)) {
            __builtin_nontemporal_store((f32x4){v[0], v[1], v[2], v[3]}, (f32x4*)op);
            __builtin_nontemporal_store((f32x4){v[4], v[5], v[6], v[7]}, (f32x4*)(op + 4));
        } else {
            *(f32x4*)op = (f32x4){v[0], v[1], v[2], v[3]};
            *(f32x4*)(op + 4) = (f32x4){v[4], v[5], v[6], v[7]};
        }
    } else {
        u32x4 o;
        if (UNITE_EF(EF_F16, g.residual_bf16 == 2)) o = (u32x4){pack_f16x2(v[0], v[1]), pack_f16x2(v[2], v[3]), pack_f16x2(v[4], v[5]), pack_f16x2(v[6], v[7])};
        else o = (u32x4){pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
        if (UNITE_EF(EF_NT2, p.nt_store == 2)) {
            const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)g.out, 0, 0x7FFFFFF0, 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b128(o, ro, (uint32_t)(((size_t)gm * g.ldc + gn) * 2), 0, 16);
        } else if (UNITE_EF(EF_NT, p.nt_store != 0)) __builtin_nontemporal_store(o, (u32x4*)((uint16_t*)g.out + (size_t)gm * g.ldc + gn));
        else *(u32x4*)((uint16_t*)g.out + (size_t)gm * g.ldc + gn) = o;
    }
    if (UNITE_EF(EF_COPY, g.out_bf16_copy != nullptr)) {
        u32x4 o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
        *(u32x4*)((uint16_t*)g.out_bf16_copy + (size_t)gm * g.ld_copy + gn) = o;
    }
    if (UNITE_EF(EF_CSUM, csum != nullptr)) {      // column sums of what was stored (a later colsum over a bf16 output would read the rounded values)
        if (UNITE_EF(EF_OUTF32, g.out_f32 != 0)) {
#pragma unroll
            for (int e = 0; e < 8; ++e) csum[e] += v[e];
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const uint32_t w = pack_bf16x2(v[2 * e], v[2 * e + 1]);
                csum[2 * e] += __uint_as_float(w << 16);
                csum[2 * e + 1] += __uint_as_float(w & 0xFFFF0000u);
            }
        }
    }
}

__device__ __forceinline__ void load_bias8(const unite_gemm_args& g, int gn, f32x4& b0, f32x4& b1) {
    b0 = b1 = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (g.bias && gn < g.N) {
        b0 = *(const f32x4*)(g.bias + gn);
        b1 = *(const f32x4*)(g.bias + gn + 4);
    }
}

template <bool TA, bool TB>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(const Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unite_gemm_args& g = p.a;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    // XCD-aware tile order: workgroups b, b+8, ... share an XCD (and its L2) -> give each XCD a contiguous
    // range of tiles, N fastest, so the A row panel and the whole B operand stay L2-resident.
    const int nbn = (g.N + BN - 1) / BN, nbm = (g.M + BM - 1) / BM, nb = nbm * nbn, nbt = nb * p.splitk;
    const int bid = blockIdx.x, xcd = bid & 7, qq = nbt >> 3, rr = nbt & 7;
    const int lin = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
    const int tile = lin % nb, slice = lin / nb;      // neighbours on an XCD: different tiles of the SAME K slice
    const int m0 = (tile / nbn) * BM, n0 = (tile % nbn) * BN;
    const int k_begin = slice * p.k_chunk, k_end = min(g.K, k_begin + p.k_chunk);

    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)g.A, 0, (int)p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)g.B, 0, (int)p.b_bytes, 0x00020000);

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 bias0, bias1;
    load_bias8(g, n0 + (tid & 15) * 8, bias0, bias1);

    const int nk = (k_end - k_begin + BK - 1) / BK;
    stage_tile<TA>(rsA, smem, m0, k_begin, g.M, k_end, g.lda, wave, lane);
    stage_tile<TB>(rsB, smem + TILE_BYTES, n0, k_begin, g.N, k_end, g.ldb, wave, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int t = 0; t < nk; ++t) {
        char* cur = smem + (t & 1) * STAGE_BYTES;
        char* nxt = smem + ((t + 1) & 1) * STAGE_BYTES;
        if (t + 1 < nk) {
            stage_tile<TA>(rsA, nxt, m0, k_begin + (t + 1) * BK, g.M, k_end, g.lda, wave, lane);
            stage_tile<TB>(rsB, nxt + TILE_BYTES, n0, k_begin + (t + 1) * BK, g.N, k_end, g.ldb, wave, lane);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[4], bf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = load_frag<TA>(cur, wm * 64 + i * 16, ks, lane);
#pragma unroll
            for (int j = 0; j < 4; ++j) bf[j] = load_frag<TB>(cur + TILE_BYTES, wn * 64 + j * 16, ks, lane);
            lds_tr_fence<TA || TB>();
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(af[i], bf[j], acc[i][j]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // LDS-DMA of tile t+1 has landed
        __syncthreads();                                   // ... for every wave; reads of tile t retired
    }

    // ---- epilogue: accumulators -> LDS (64 rows at a time) -> coalesced stores
    float* cs = (float*)smem;
    const int G = lane >> 4, c16 = lane & 15;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if (wm == half) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        cs[(i * 16 + 4 * G + r) * CS_LD + wn * 64 + j * 16 + c16] = acc[i][j][r];
        }
        __syncthreads();
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int lr = pass * 16 + (tid >> 4);
            const int col = (tid & 15) * 8;
            const int gm = m0 + half * 64 + lr, gn = n0 + col;
            if (gm < g.M && gn < g.N)
                epilogue_chunk(p, g, slice, gm, gn, *(const f32x4*)(cs + lr * CS_LD + col), *(const f32x4*)(cs + lr * CS_LD + col + 4), bias0, bias1);
        }
        __syncthreads();
    }
}

// =====================================================================================================================
// Deep-pipelined kernel.  Tile = (2 HALF) x (2 HALF) x 64 with HALF = 128 (8 waves, 1 workgroup / CU, 128 KiB LDS) or
// HALF = 64 (4 waves, 2 workgroups / CU, 64 KiB LDS each).  The LDS holds two K-tiles of four half-tiles (A rows 0..HALF-1
// / HALF..2HALF-1, B cols likewise).  Each K-tile is computed in four phases, one quadrant of the wave's output each;
// wave (wm, wn) owns rows {h HALF + wm HALF/2 ..} and columns {nh HALF + wn 32 ..} of BOTH halves, so quadrant (h, nh)
// reads only half-tiles A_h and B_nh.  With the quadrant order (0,0) (0,1) (1,1) (1,0) the half-tiles of K-tile t are first
// needed at phases 1 (A0, B0), 2 (B1), 3 (A1) and are dead one phase later, which lets the LDS-DMA stream run far ahead
// through only two K-tiles of LDS:
//   at global phase g the wave issues half-tile load q = g + 6   (sequence A0,B0,B1,A1 per K-tile, 2 DMAs per wave),
//   phases 1-3 start with  s_waitcnt vmcnt(8); s_barrier  (4 half-tiles stay in flight across the barrier),
// and the queue is never drained inside the loop.  The B0 fragments stay in registers for phase 4.  Loads past the last
// K-tile still issue (their k is out of range: they only zero-fill a dead slot) so that the vmcnt arithmetic is uniform.
// =====================================================================================================================
__device__ __forceinline__ int swz128t(int k) { return ((k >> 1) & 1) | (((k >> 3) & 1) << 1); }   // 32-B chunk swizzle of [k][64] images

// one 1-KiB LDS-DMA piece `it` of a half-tile image (HALF rows/cols x 64 k)
__device__ __forceinline__ int swzperm(int n) { return ((n >> 1) & 1) | (((n >> 4) & 3) << 1); }   // for rows read as 16g + 4j + r

template <bool TR, int HALF, bool PERM = false>
__device__ __forceinline__ void stage_piece(__amdgpu_buffer_rsrc_t rs, char* slot, int it, int r0, int k0, int R, int K, int ld, int lane) {
    uint32_t voff;
    if (!TR) {                                   // image [HALF][64 k], 128-B rows, 16-B chunk c at c ^ (row & 7)
        const int r = it * 8 + (lane >> 3);
        const int lc = (lane & 7) ^ (PERM ? swzperm(r) : (r & 7));
        const int gr = r0 + r, gk = k0 + lc * 8;
        voff = (gr < R && gk < K) ? (uint32_t)(gr * ld + gk) * 2u : OOB_OFFSET;
    } else if (HALF == 128) {                    // image [64 k][128], 256-B rows, swz256
        const int kr = it * 4 + (lane >> 4);
        const int lc = (lane & 15) ^ swz256(kr);
        const int gk = k0 + kr, gr = r0 + lc * 8;
        voff = (gk < K && gr < R) ? (uint32_t)(gk * ld + gr) * 2u : OOB_OFFSET;
    } else {                                     // image [64 k][64], 128-B rows, 32-B chunk c at c ^ swz128t(k)
        const int kr = it * 8 + (lane >> 3), pc = lane & 7;
        const int lc = (((pc >> 1) ^ swz128t(kr)) << 1) | (pc & 1);
        const int gk = k0 + kr, gr = r0 + lc * 8;
        voff = (gk < K && gr < R) ? (uint32_t)(gk * ld + gr) * 2u : OOB_OFFSET;
    }
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (LDS_AS void*)(slot + it * 1024), 16, voff, 0, 0, 0);
}

template <bool TR, int HALF>
__device__ __forceinline__ bf16x8 load_frag_h(const char* slot, int row0, int ks, int lane) {
    if (!TR || HALF == 128) return load_frag<TR>(slot, row0, ks, lane);
    const int G = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const int c32 = row0 >> 4;
    const int k_lo = ks * 32 + 8 * G + q, k_hi = k_lo + 4;
    return tr_join(lds_read_tr16_raw(slot + 128 * k_lo + ((c32 ^ swz128t(k_lo)) << 5) + 8 * pp),
                   lds_read_tr16_raw(slot + 128 * k_hi + ((c32 ^ swz128t(k_hi)) << 5) + 8 * pp));
}

// B fragment of the wide kernel with the n <-> MFMA-row permutation  n = bcol + 16 (rho >> 2) + 4 j + (rho & 3):
// with the operands swapped (C^T in the accumulators) lane group G then holds 16 CONSECUTIVE columns over the wave's four
// n-tiles, i.e. 64 B (f32) / 32 B (bf16) per lane and 256 B / 128 B contiguous per output row: the epilogue stores
// straight from the registers (no LDS round trip, which would compete with the co-resident workgroup's main loop).
template <bool TR>
__device__ __forceinline__ bf16x8 load_bfrag_perm(const char* slot, int bcol, int j, int ks, int lane) {
    if (!TR) {
        const int rho = lane & 15;
        const int n = bcol + 16 * (rho >> 2) + 4 * j + (rho & 3);
        const int lc = ks * 4 + (lane >> 4);
        return *(const bf16x8*)(slot + n * 128 + ((lc ^ swzperm(n)) << 4));
    } else {
        const int G = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
        const int ch = (bcol >> 3) + 2 * pp + (j >> 1);
        const int k_lo = ks * 32 + 8 * G + q, k_hi = k_lo + 4;
        return tr_join(lds_read_tr16_raw(slot + 256 * k_lo + ((ch ^ swz256(k_lo)) << 4) + 8 * (j & 1)),
                       lds_read_tr16_raw(slot + 256 * k_hi + ((ch ^ swz256(k_hi)) << 4) + 8 * (j & 1)));
    }
}

// Row sums of op(A) beside the product (unite_gemm_args.rowsum_a_out: the bias gradient of a weight-gradient GEMM): the A fragments a
// wave already holds are multiplied once more, with a B fragment that is 1.0 in column i and 0 elsewhere for m-tile i, so ONE
// accumulator tile collects the wave's MT row-sum vectors in its columns 0 .. MT-1 (lane (G, c) with c < MT: rows 16 c + 4 G + r).
template <int MT>
__device__ __forceinline__ void rowsum_mfma(f32x4& racc, const bf16x8 (&af)[MT][2], int ks_mask, int lane) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
        if (ks_mask & (1 << ks)) {
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const uint32_t w = ((lane & 15) == i) ? 0x3F803F80u : 0u;      // bf16 1.0 | 1.0
                const u32x4 e = {w, w, w, w};
                racc = mfma16(af[i][ks], __builtin_bit_cast(bf16x8, e), racc);
            }
        }
}

// ---- software-pipelined main loop (SCHED 1): untracked fragment reads ------------------------------------------------------------
// One reader per operand and wave: the lane's byte address inside a half-tile image for m-/n-tile 0, k-step 0.  Tile I, k-step KS and the slot
// are immediate offsets of the ds instruction; the XOR-swizzled k-strided images need one v_xor per tile instead (the tile index only flips
// chunk bits the lane part leaves clear).  The reads are inline asm the compiler's wait-count tracking does not see: the loop that issues them
// places every `s_waitcnt lgkmcnt(N)` by hand, N = the number of ds instructions issued AFTER the one a group of MFMAs needs (LDS returns in order).
template <bool TR, int HALF>
struct FragReader {
    static constexpr int READS = TR ? 2 : 1;      // ds instructions per fragment
    uint32_t a0, a1;
    __device__ __forceinline__ void init(int rowbase, int lane) {      // rowbase: the wave's first row (A) / column (B) inside a half, a multiple of 32
        const int G = lane >> 4;
        if (!TR) {                                   // [HALF][64 k] image, chunk c of row r at c ^ (r & 7): k-step 1 flips chunk bit 2
            a0 = (uint32_t)((rowbase + (lane & 15)) * 128 + ((G ^ (lane & 7)) << 4));
            a1 = a0 ^ 64u;
        } else if (HALF == 128) {                    // [64 k][128] image, 16-B chunk c of k-row k at c ^ swz256(k); k + 4 flips chunk bit 0
            const int q = (lane >> 2) & 3, pp = lane & 3, k = 8 * G + q;
            a0 = (uint32_t)(256 * k + ((((rowbase >> 3) + (pp >> 1)) ^ swz256(k)) << 4) + 8 * (pp & 1));
            a1 = a0 ^ 16u;
        } else {                                     // [64 k][64] image, 32-B chunk c of k-row k at c ^ swz128t(k) (the same for k and k + 4)
            const int q = (lane >> 2) & 3, pp = lane & 3, k = 8 * G + q;
            a0 = (uint32_t)(128 * k + (((rowbase >> 4) ^ swz128t(k)) << 5) + 8 * pp);
            a1 = 0;
        }
    }
    // fragment of tile I (16 rows / columns), k-step KS, from the slot at byte offset OFF of the set at LDS address `set`
    template <int I, int KS, int OFF>
    __device__ __forceinline__ bf16x8 read(uint32_t set) const {
        if constexpr (!TR) {
            return lds_read_b128_raw<OFF + I * 2048>((KS ? a1 : a0) + set);
        } else if constexpr (HALF == 128) {
            const uint32_t x = (a0 ^ (uint32_t)(I << 5)) + set, y = (a1 ^ (uint32_t)(I << 5)) + set;
            return tr_join(lds_read_tr16_raw_off<OFF + KS * 8192>(x), lds_read_tr16_raw_off<OFF + KS * 8192 + 1024>(y));
        } else {
            const uint32_t x = (a0 ^ (uint32_t)(I << 5)) + set;
            return tr_join(lds_read_tr16_raw_off<OFF + KS * 4096>(x), lds_read_tr16_raw_off<OFF + KS * 4096 + 512>(x));
        }
    }
};
#define UNITE_LGKM(n) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(n) : "memory")
// diagnostics: shader-clock stamps of a workgroup's phases (thread 0, UNITE_GEMM_DEBUG_SKIP=7, eight 64-bit words per workgroup in the caller's
// workspace; tools/gemm_phase_stamps.py).  Outside every loop.
#define UNITE_STAMP(k) do { if (p.debug_skip == 7 && threadIdx.x == 0 && g.workspace) ((unsigned long long*)g.workspace)[(size_t)blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#define UNITE_FENCE() __builtin_amdgcn_sched_barrier(0)

// WG: the weight-gradient features -- 1: row sums of op(A) (rowsum_a_out); 2: + the in-launch split-K reduction.  A template parameter, not run-time
// flags: the extra accumulator, the one-hot fragment and the values the reduction keeps alive cost the 256^2 kernel 16 registers per lane (232 -> 248) -- with 232 a CU that holds a GEMM workgroup still has 48
// registers per SIMD lane free, enough for a wave of the streaming kernels (LayerNorm, reductions) of another stream to run beside it; with
// 248 it has not, and the overlapped step was 0.3-0.4 ms slower although every kernel timed alone was unchanged (round 3).
// EPI 1 (SCHED 1, WG 0 only): the products are computed TRANSPOSED (the B fragment in the MFMA's A slot), so a lane holds four consecutive columns
// of one output row; bias / activation / row scale are applied in the accumulator registers, the results are packed to bf16 there and staged ONCE
// through a bf16 image of the whole tile (8-byte writes, 128 KiB) instead of twice through an f32 image of half of it (4-byte writes, 2 x 128 KiB):
// a quarter of the LDS bytes, a quarter of the write instructions, two barriers instead of four.  For bf16 outputs without residual / saved
// pre-activation / column sums / split-K (the launcher checks); bit-identical to EPI 0 (same sums, same rounding points).
template <int HALF, bool TA, bool TB, int WG = 0, int SCHED = 0, int EPI = 0>
__global__ __launch_bounds__(HALF * 4, 2) void gemm_deep_kernel(const Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TILE = 2 * HALF, SLOT = HALF * 128, WN = HALF / 32, MT = HALF / 32;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    // XCD-aware bijective remap of the launch order, then (grouped launches) the problem this workgroup belongs to
    const int nbt = (int)gridDim.x;
    const int bid = blockIdx.x, xcd = bid & 7, qq = nbt >> 3, rr = nbt & 7;
    const int lin = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
    int gi = 0;
    if (p.ngroups > 1) {
#pragma unroll
        for (int i = 0; i < MAX_GROUP - 1; ++i) gi += (i + 1 < p.ngroups && lin >= p.tile_end[i]) ? 1 : 0;
    }
    // field-by-field scalar selects with CONSTANT indices: a runtime-indexed address into the by-value kernel argument would
    // make hipcc copy all of Params to scratch memory (162 scratch instructions, 2 x slower kernels)
    unite_gemm_args g = p.a;
    uint32_t a_bytes = p.a_bytes, b_bytes = p.b_bytes;
    int lin0 = 0;
    if (p.ngroups > 1) {
#define UNITE_SEL(f) g.f = gi == 1 ? p.more[0].f : gi == 2 ? p.more[1].f : gi == 3 ? p.more[2].f : g.f
        UNITE_SEL(M); UNITE_SEL(N); UNITE_SEL(K); UNITE_SEL(A); UNITE_SEL(lda); UNITE_SEL(B); UNITE_SEL(ldb); UNITE_SEL(bias);
        UNITE_SEL(act); UNITE_SEL(aux_in); UNITE_SEL(ld_aux_in); UNITE_SEL(aux_out); UNITE_SEL(ld_aux_out); UNITE_SEL(row_scale);
        UNITE_SEL(rows_per_scale); UNITE_SEL(residual); UNITE_SEL(ldr); UNITE_SEL(out); UNITE_SEL(ldc); UNITE_SEL(out_f32);
        UNITE_SEL(accumulate); UNITE_SEL(out_bf16_copy); UNITE_SEL(ld_copy);
#undef UNITE_SEL
        a_bytes = gi == 1 ? p.more_a_bytes[0] : gi == 2 ? p.more_a_bytes[1] : gi == 3 ? p.more_a_bytes[2] : a_bytes;
        b_bytes = gi == 1 ? p.more_b_bytes[0] : gi == 2 ? p.more_b_bytes[1] : gi == 3 ? p.more_b_bytes[2] : b_bytes;
        lin0 = gi == 1 ? p.tile_end[0] : gi == 2 ? p.tile_end[1] : gi == 3 ? p.tile_end[2] : 0;
    }
    UNITE_STAMP(0);
    const int gM = g.M, gN = g.N, gK = g.K, lda = g.lda, ldb = g.ldb;
    const int nbn = (gN + TILE - 1) / TILE, nbm = (gM + TILE - 1) / TILE, nb = nbm * nbn;
    const int tile = (lin - lin0) % nb, slice = (lin - lin0) / nb;
    int tm = tile / nbn, tn = tile % nbn;
    if (p.group_rows > 1) {          // column-major inside a group of tile rows (the last group may be shorter)
        const int per = p.group_rows * nbn, gq = tile / per, within = tile - gq * per;
        const int rows = min(p.group_rows, nbm - gq * p.group_rows);
        tn = within / rows;
        tm = gq * p.group_rows + (within - tn * rows);
    }
    const int m0 = tm * TILE, n0 = tn * TILE;
    const int k_begin = slice * p.k_chunk, k_end = min(gK, k_begin + p.k_chunk);
    const int nk = (k_end - k_begin + BK - 1) / BK;

    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)g.A, 0, (int)a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)g.B, 0, (int)b_bytes, 0x00020000);

    // half-tile load number q: K-tile u = q >> 2, kind = q & 3 (0: A0, 1: B0, 2: B1, 3: A1); slots of set (u & 1): A0 A1 B0 B1
    auto issue = [&](int q) {
        const int u = q >> 2, kind = q & 3;
        char* set = smem + (u & 1) * (4 * SLOT);
        const int k0 = k_begin + u * BK;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int it = wave * 2 + i;
            if (kind == 0) stage_piece<TA, HALF>(rsA, set, it, m0, k0, gM, k_end, lda, lane);
            else if (kind == 1) stage_piece<TB, HALF>(rsB, set + 2 * SLOT, it, n0, k0, gN, k_end, ldb, lane);
            else if (kind == 2) stage_piece<TB, HALF>(rsB, set + 3 * SLOT, it, n0 + HALF, k0, gN, k_end, ldb, lane);
            else stage_piece<TA, HALF>(rsA, set + SLOT, it, m0 + HALF, k0, gM, k_end, lda, lane);
        }
    };

    f32x4 acc[2][MT][2][2];       // [A half][m-tile][B half][n-tile]
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[h][i][nh][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int q = 0; q < (SCHED == 1 ? 7 : 6); ++q) issue(q);

    const int arow = wm * (HALF / 2), bcol = wn * 32;
    // row sums of op(A): the workgroups of column tile 0 only; the (A half, k-step) pairs of a K-tile are dealt over the waves of a row
    // group (HALF 128: four waves, one pair each; HALF 64: two waves, one half each), +4 MFMAs per K-tile and wave
    const bool rs_on = WG >= 1 && p.ngroups == 1 && g.rowsum_a_out != nullptr && n0 == 0;
    const int rs_h = WN == 4 ? (wn >> 1) : wn, rs_ks = WN == 4 ? (1 << (wn & 1)) : 3;
    f32x4 racc = (f32x4){0.f, 0.f, 0.f, 0.f};
    if constexpr (SCHED == 0) {
    for (int t = 0; t < nk; ++t) {
        char* set = smem + (t & 1) * (4 * SLOT);
        const char* A0 = set;
        const char* A1 = set + SLOT;
        const char* B0 = set + 2 * SLOT;
        const char* B1 = set + 3 * SLOT;
        const int gph = 4 * t;
        bf16x8 af[MT][2], b0f[2][2], b1f[2][2];

        // ---- phase 1: quadrant (0,0); A0 and B0 of this K-tile have landed once at most 8 DMAs are outstanding
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) af[i][ks] = load_frag_h<TA, HALF>(A0, arow + i * 16, ks, lane);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) b0f[j][ks] = load_frag_h<TB, HALF>(B0, bcol + j * 16, ks, lane);
        issue(gph + 6);
        lds_tr_fence<TA || TB>();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[0][i][0][j] = mfma16(af[i][ks], b0f[j][ks], acc[0][i][0][j]);
        if constexpr (WG >= 1) { if (rs_on && rs_h == 0) rowsum_mfma<MT>(racc, af, rs_ks, lane); }
        __builtin_amdgcn_s_setprio(0);

        // ---- phase 2: quadrant (0,1); needs B1
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) b1f[j][ks] = load_frag_h<TB, HALF>(B1, bcol + j * 16, ks, lane);
        issue(gph + 7);
        lds_tr_fence<TA || TB>();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[0][i][1][j] = mfma16(af[i][ks], b1f[j][ks], acc[0][i][1][j]);
        __builtin_amdgcn_s_setprio(0);

        // ---- phase 3: quadrant (1,1); needs A1
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) af[i][ks] = load_frag_h<TA, HALF>(A1, arow + i * 16, ks, lane);
        issue(gph + 8);
        lds_tr_fence<TA || TB>();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[1][i][1][j] = mfma16(af[i][ks], b1f[j][ks], acc[1][i][1][j]);
        if constexpr (WG >= 1) { if (rs_on && rs_h == 1) rowsum_mfma<MT>(racc, af, rs_ks, lane); }
        __builtin_amdgcn_s_setprio(0);

        // ---- phase 4: quadrant (1,0); operands already in registers
        issue(gph + 9);
        lds_tr_fence<TA || TB>();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[1][i][0][j] = mfma16(af[i][ks], b0f[j][ks], acc[1][i][0][j]);
        __builtin_amdgcn_s_setprio(0);
    }
    } else {
    // ---- SCHED 1: the same quadrant walk, software-pipelined.  A phase no longer opens with its fragment reads: every fragment is read one
    // phase EARLIER, between the MFMAs of the phase before, into registers that have just died --
    //   P1  X(A0) Y(B0) -> acc(0,0)    reads Z <- B1                       DMA A1 (t+1)
    //   P2  X(A0) Z(B1) -> acc(0,1)    reads X <- A1, tile by tile behind the last MFMA pair that used it     DMA A0 (t+2)
    //   P3  X(A1) Z(B1) -> acc(1,1)    --                                  DMA B0 (t+2)
    //   P4  X(A1) Y(B0) -> acc(1,0)    reads X <- A0', Y <- B0' of the NEXT K-tile the same way             DMA B1 (t+2)
    // so a wave never waits for LDS latency at the head of a phase, and the reads and the two LDS-DMA issues of a phase sit between MFMA pairs
    // instead of in front of the cluster.  Every half-tile is therefore needed one phase earlier than in SCHED 0: the DMA stream runs 7 half-tiles
    // ahead (same eight slots: a slot is refilled two or more barriers after the phase that read it), `vmcnt(8)` + barrier in front of P1, P2 and
    // P4 retires exactly the half-tile that phase reads (P3 reads nothing and has no barrier).  The reads are untracked asm (FragReader); the
    // lgkmcnt numbers below are counts of ds instructions issued after the fragment a pair needs.
    constexpr int RA = FragReader<TA, HALF>::READS, RB = FragReader<TB, HALF>::READS;
    constexpr int SET = 4 * SLOT, OA0 = 0, OA1 = SLOT, OB0 = 2 * SLOT, OB1 = 3 * SLOT;
    constexpr int NP = 2 * MT;                          // MFMA pairs per phase (k-step major)
    constexpr int D0 = MT == 4 ? 2 : 0, D1 = MT == 4 ? 5 : 2;      // the pairs behind which the phase's two LDS-DMA pieces are issued
    FragReader<TA, HALF> ra;
    FragReader<TB, HALF> rb;
    ra.init(arow, lane);
    rb.init(bcol, lane);
    const uint32_t lds0 = lds_address(smem);
    bf16x8 X[MT][2], Y[2][2], Z[2][2];
    // EPI 1: C^T tiles -- the B fragment goes into the MFMA's A slot, so lane (G, c) ends up with C[row c][columns 4 G .. 4 G + 3] of each 16 x 16 tile
    auto mm = [](bf16x8 a, bf16x8 b, f32x4 c) { return EPI == 1 ? mfma16(b, a, c) : mfma16(a, b, c); };
    // one LDS-DMA piece (i = 0, 1) of half-tile `kind` of K-tile u
    auto piece = [&](auto kind_c, int u, int i) {
        constexpr int kind = decltype(kind_c)::value;
        char* set = smem + (u & 1) * SET;
        const int k0 = k_begin + u * BK, it = wave * 2 + i;
        if constexpr (kind == 0) stage_piece<TA, HALF>(rsA, set + OA0, it, m0, k0, gM, k_end, lda, lane);
        else if constexpr (kind == 1) stage_piece<TB, HALF>(rsB, set + OB0, it, n0, k0, gN, k_end, ldb, lane);
        else if constexpr (kind == 2) stage_piece<TB, HALF>(rsB, set + OB1, it, n0 + HALF, k0, gN, k_end, ldb, lane);
        else stage_piece<TA, HALF>(rsA, set + OA1, it, m0 + HALF, k0, gM, k_end, lda, lane);
    };
    // X <- A0, Y <- B0 of K-tile 0, in the order P4 issues them (k-step major: X[.][ks] then Y[.][ks])
    asm volatile("s_waitcnt vmcnt(10)" ::: "memory");      // 14 pieces issued: A0, B0 of K-tile 0 have landed
    __builtin_amdgcn_s_barrier();
    static_for<0, 2>([&](auto ksc) {
        constexpr int ks = decltype(ksc)::value;
        static_for<0, MT>([&](auto ic) { X[decltype(ic)::value][ks] = ra.template read<decltype(ic)::value, ks, OA0>(lds0); });
        static_for<0, 2>([&](auto jc) { Y[decltype(jc)::value][ks] = rb.template read<decltype(jc)::value, ks, OB0>(lds0); });
    });
    UNITE_FENCE();
    for (int t = 0; t < nk; ++t) {
        const uint32_t cur = lds0 + (uint32_t)((t & 1) * SET), nxt = lds0 + (uint32_t)(((t + 1) & 1) * SET);
        // ---- P1: (0,0)
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // B1 of this K-tile has landed
        __builtin_amdgcn_s_barrier();
        UNITE_LGKM(MT * RA + 2 * RB);                      // X[.][0], Y[.][0]: only the k-step-1 fragments may still be on their way
        UNITE_FENCE();
        static_for<0, NP>([&](auto pc) {
            constexpr int pi = decltype(pc)::value, ks = pi / MT, i = pi % MT;
            if constexpr (pi == MT) { UNITE_LGKM(4 * RB); UNITE_FENCE(); }      // X[.][1], Y[.][1]: younger are the four Z fragments only
            acc[0][i][0][0] = mm(X[i][ks], Y[0][ks], acc[0][i][0][0]);
            acc[0][i][0][1] = mm(X[i][ks], Y[1][ks], acc[0][i][0][1]);
            UNITE_FENCE();
            if constexpr (pi == 0) {
                Z[0][0] = rb.template read<0, 0, OB1>(cur);
                Z[1][0] = rb.template read<1, 0, OB1>(cur);
                Z[0][1] = rb.template read<0, 1, OB1>(cur);
                Z[1][1] = rb.template read<1, 1, OB1>(cur);
            }
            if constexpr (pi == D0) piece(std::integral_constant<int, 3>{}, t + 1, 0);
            if constexpr (pi == D1) piece(std::integral_constant<int, 3>{}, t + 1, 1);
            UNITE_FENCE();
        });
        if constexpr (WG >= 1) { if (rs_on && rs_h == 0) rowsum_mfma<MT>(racc, X, rs_ks, lane); UNITE_FENCE(); }
        // ---- P2: (0,1); X dies pair by pair and is refilled with A1
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // A1 of this K-tile has landed
        __builtin_amdgcn_s_barrier();
        UNITE_LGKM(2 * RB);                                // Z[.][0]
        UNITE_FENCE();
        static_for<0, NP>([&](auto pc) {
            constexpr int pi = decltype(pc)::value, ks = pi / MT, i = pi % MT;
            if constexpr (pi == MT) { UNITE_LGKM(MT * RA); UNITE_FENCE(); }     // Z[.][1]: younger are the MT fragments read behind pairs 0 .. MT-1
            acc[0][i][1][0] = mm(X[i][ks], Z[0][ks], acc[0][i][1][0]);
            acc[0][i][1][1] = mm(X[i][ks], Z[1][ks], acc[0][i][1][1]);
            UNITE_FENCE();
            X[i][ks] = ra.template read<i, ks, OA1>(cur);
            if constexpr (pi == D0) piece(std::integral_constant<int, 0>{}, t + 2, 0);
            if constexpr (pi == D1) piece(std::integral_constant<int, 0>{}, t + 2, 1);
            UNITE_FENCE();
        });
        // ---- P3: (1,1); no reads, no barrier
        UNITE_LGKM(MT * RA);                               // X[.][0]
        UNITE_FENCE();
        static_for<0, NP>([&](auto pc) {
            constexpr int pi = decltype(pc)::value, ks = pi / MT, i = pi % MT;
            if constexpr (pi == MT) { UNITE_LGKM(0); UNITE_FENCE(); }
            acc[1][i][1][0] = mm(X[i][ks], Z[0][ks], acc[1][i][1][0]);
            acc[1][i][1][1] = mm(X[i][ks], Z[1][ks], acc[1][i][1][1]);
            UNITE_FENCE();
            if constexpr (pi == D0) piece(std::integral_constant<int, 1>{}, t + 2, 0);
            if constexpr (pi == D1) piece(std::integral_constant<int, 1>{}, t + 2, 1);
            if constexpr (pi == D0 || pi == D1) UNITE_FENCE();
        });
        if constexpr (WG >= 1) { if (rs_on && rs_h == 1) rowsum_mfma<MT>(racc, X, rs_ks, lane); UNITE_FENCE(); }
        // ---- P4: (1,0); X and Y die pair by pair and are refilled with A0, B0 of the next K-tile
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // A0, B0 of K-tile t + 1 have landed
        __builtin_amdgcn_s_barrier();
        static_for<0, NP>([&](auto pc) {
            constexpr int pi = decltype(pc)::value, ks = pi / MT, i = pi % MT;
            acc[1][i][0][0] = mm(X[i][ks], Y[0][ks], acc[1][i][0][0]);
            acc[1][i][0][1] = mm(X[i][ks], Y[1][ks], acc[1][i][0][1]);
            UNITE_FENCE();
            X[i][ks] = ra.template read<i, ks, OA0>(nxt);
            if constexpr (i == MT - 1) {
                Y[0][ks] = rb.template read<0, ks, OB0>(nxt);
                Y[1][ks] = rb.template read<1, ks, OB0>(nxt);
            }
            if constexpr (pi == D0) piece(std::integral_constant<int, 2>{}, t + 2, 0);
            if constexpr (pi == D1) piece(std::integral_constant<int, 2>{}, t + 2, 1);
            UNITE_FENCE();
        });
    }
    UNITE_LGKM(0);      // the last P4 read fragments nobody uses: they must have arrived before their registers are reused
    UNITE_FENCE();
    }
    UNITE_STAMP(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the trailing zero-fill DMAs must not land in the epilogue image
    __syncthreads();
    UNITE_STAMP(2);
    if (p.debug_skip == 1) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                    for (int j = 0; j < 2; ++j) asm volatile("" ::"v"(acc[h][i][nh][j]));
        return;
    }

    if constexpr (EPI == 1) {
        // ---- epilogue from transposed accumulators: lane (G, c) of tile (h, i, nh, j) holds row  h HALF + arow + 16 i + c,  columns  nh HALF + bcol +
        // 16 j + 4 G .. + 3.  Bias, activation and row scale in the registers, bf16 pairs packed, ONE 8-byte LDS write per tile into a bf16 image of the
        // whole tile (512-byte rows; the 16-byte chunk index is XORed with row & 15: the 16 lanes of a write group are 16 different rows of one column);
        // then every thread moves sixteen 16-byte chunks, a wave two whole rows per instruction.
        char* img = smem;
        const int G = lane >> 4, c16 = lane & 15;
        f32x4 bb[2][2];
#pragma unroll
        for (int nh = 0; nh < 2; ++nh)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int gn = n0 + nh * HALF + bcol + j * 16 + 4 * G;
                bb[nh][j] = (g.bias && gn < gN) ? *(const f32x4*)(g.bias + gn) : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        // activation and row scale are decided ONCE, outside the loops (compile-time inside them): left as run-time tests on g.act / g.row_scale per
        // element they became ~750 scalar branches per workgroup -- 12.7 k of a c_fc tile's 56.7 k cycles (round 4, shader-clock stamps)
        auto pack_all = [&](auto act_c, auto scale_c) {
            constexpr int ACT = decltype(act_c)::value;
            constexpr bool SCALE = decltype(scale_c)::value;
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const int row = h * HALF + arow + i * 16 + c16;
                    float sc = 1.f;
                    if constexpr (SCALE) sc = g.row_scale[min(m0 + row, gM - 1) / g.rows_per_scale];
#pragma unroll
                    for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            f32x4 v = acc[h][i][nh][j];
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                float x = v[e] + bb[nh][j][e];
                                if constexpr (ACT == UNITE_ACT_GELU) x = gelu_erf(x);
                                else if constexpr (ACT == UNITE_ACT_QUICKGELU) x = quick_gelu(x);
                                if constexpr (SCALE) x *= sc;
                                v[e] = x;
                            }
                            const int unit = (nh * HALF + bcol + j * 16 + 4 * G) >> 2;            // 8-byte unit (four bf16) inside the 512-byte row
                            const u32x2 w = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
                            *(u32x2*)(img + row * (TILE * 2) + ((((unit >> 1) ^ (row & 15)) << 4) | ((unit & 1) << 3))) = w;
                        }
                }
        };
        const bool scaled = g.row_scale != nullptr;
        if (g.act == UNITE_ACT_GELU) { if (scaled) pack_all(std::integral_constant<int, UNITE_ACT_GELU>{}, std::true_type{}); else pack_all(std::integral_constant<int, UNITE_ACT_GELU>{}, std::false_type{}); }
        else if (g.act == UNITE_ACT_QUICKGELU) { if (scaled) pack_all(std::integral_constant<int, UNITE_ACT_QUICKGELU>{}, std::true_type{}); else pack_all(std::integral_constant<int, UNITE_ACT_QUICKGELU>{}, std::false_type{}); }
        else { if (scaled) pack_all(std::integral_constant<int, UNITE_ACT_NONE>{}, std::true_type{}); else pack_all(std::integral_constant<int, UNITE_ACT_NONE>{}, std::false_type{}); }
        UNITE_STAMP(3);
        __syncthreads();
        UNITE_STAMP(4);
        if (p.debug_skip == 4) return;
        constexpr int CH = TILE / 8;                         // 16-byte chunks per row
        const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)g.out, 0, 0x7FFFFFF0, 0x00020000);
        // the store kind is decided once, outside the loop (see pack_all)
        auto store_all = [&](auto kind_c) {
            constexpr int KIND = decltype(kind_c)::value;      // 0 plain, 1 non-temporal, 2 write-through (sc1), 3 none (timing)
#pragma unroll 4
            for (int e = 0; e < TILE * CH / (4 * HALF); ++e) {
                const int idx = tid + e * 4 * HALF, row = idx / CH, ch = idx % CH;
                const u32x4 o = *(const u32x4*)(img + row * (TILE * 2) + ((ch ^ (row & 15)) << 4));
                const int gm = m0 + row, gn = n0 + ch * 8;
                if (gm < gM && gn < gN) {
                    if constexpr (KIND == 3) asm volatile("" ::"v"(o));
                    else if constexpr (KIND == 2) __builtin_amdgcn_raw_buffer_store_b128(o, ro, (uint32_t)(((size_t)gm * g.ldc + gn) * 2), 0, 16);
                    else if constexpr (KIND == 1) __builtin_nontemporal_store(o, (u32x4*)((uint16_t*)g.out + (size_t)gm * g.ldc + gn));
                    else *(u32x4*)((uint16_t*)g.out + (size_t)gm * g.ldc + gn) = o;
                }
            }
        };
        if (p.debug_skip == 2) store_all(std::integral_constant<int, 3>{});
        else if (p.nt_store == 2) store_all(std::integral_constant<int, 2>{});
        else if (p.nt_store) store_all(std::integral_constant<int, 1>{});
        else store_all(std::integral_constant<int, 0>{});
        UNITE_STAMP(5);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        UNITE_STAMP(6);
        return;
    }

    // ---- epilogue: HALF rows x TILE columns at a time through an unpadded f32 image in the (now idle) ring (row stride = 0 mod 64
    // banks: 16-column groups are XOR-swizzled by the row's lane-group index so the four groups of a ds_write hit 64 banks): two passes,
    // two barriers each, TILE/32 independent 8-column chunks per thread -> wide (16 B / 32 B per lane) coalesced stores.
    // the bias chunk: loaded HERE, behind the main loop and ahead of the first staging pass (it lands under the staging writes and their
    // barrier) -- loaded in front of the main loop it held 8 registers through it, which put the plain 256^2 kernel at 234 registers: two waves
    // of it then leave 32 of a SIMD's 512, with 226 they leave 48 (a LayerNorm-forward wave of the other stream needs 32)
    f32x4 bias0, bias1;
    load_bias8(g, n0 + (tid % (TILE / 8)) * 8, bias0, bias1);
    float* cs = (float*)smem;
    const int G = lane >> 4, c16 = lane & 15;
    constexpr int CPR = TILE / 8;                     // 8-column chunks per row
    const int ccol = (tid % CPR) * 8;                 // this thread's column chunk is the same in every pass
    // the chunk loop of one pass with the epilogue's tests decided at compile time (`known` = EF_ALL) or at run time (0)
    auto chunks = [&](auto known_c, auto value_c, int h, float* csum) {
        constexpr uint32_t KN = decltype(known_c)::value, VL = decltype(value_c)::value;
#pragma unroll 2
        for (int e = 0; e < TILE / 32; ++e) {
            const int lr = (tid + e * 4 * HALF) / CPR;
            const int gm = m0 + h * HALF + lr, gn = n0 + ccol;
            const float* cp = cs + lr * TILE + (ccol ^ (((lr >> 2) & 3) << 4));
            if (gm < g.M && gn < g.N) epilogue_chunk<KN, VL>(p, g, slice, gm, gn, *(const f32x4*)cp, *(const f32x4*)(cp + 4), bias0, bias1, csum);
        }
    };
    // the same loop, FOUR chunks in flight, for the f32 + f32-residual forms (teacher out_proj / c_proj, student proj / fc2): their residual
    // chunks come from memory, and two at a time every pair waits for its own round trip -- 19 k of c_proj's 172 k cycles per pass (stamps)
    auto chunksU = [&](auto value_c, auto unroll_c, int h) {
        constexpr uint32_t VL = decltype(value_c)::value;
        constexpr int U = decltype(unroll_c)::value;
#pragma unroll U
        for (int e = 0; e < TILE / 32; ++e) {
            const int lr = (tid + e * 4 * HALF) / CPR;
            const int gm = m0 + h * HALF + lr, gn = n0 + ccol;
            const float* cp = cs + lr * TILE + (ccol ^ (((lr >> 2) & 3) << 4));
            if (gm < g.M && gn < g.N) epilogue_chunk<EF_ALL, VL>(p, g, slice, gm, gn, *(const f32x4*)cp, *(const f32x4*)(cp + 4), bias0, bias1, nullptr);
        }
    };
    auto chunks4 = [&](auto value_c, int h) { chunksU(value_c, std::integral_constant<int, 4>{}, h); };
    float csum_v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float* const csum = p.colsum_partial ? csum_v : nullptr;
    const uint32_t fw = epilogue_features(p, g, csum != nullptr);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        cs[(arow + i * 16 + 4 * G + r) * TILE + ((nh * HALF + bcol + j * 16 + c16) ^ (G << 4))] = acc[h][i][nh][j][r];
        if (h == 0) UNITE_STAMP(3);
        __syncthreads();
        if (h == 0) UNITE_STAMP(4);
        using KA = std::integral_constant<uint32_t, EF_ALL>;
#define UNITE_HOT(v) if (fw == (uint32_t)(v)) chunks(KA{}, std::integral_constant<uint32_t, (uint32_t)(v)>{}, h, nullptr); else
        if (fw == (EF_OUTF32 | EF_RES)) chunks4(std::integral_constant<uint32_t, EF_OUTF32 | EF_RES>{}, h);              // teacher out_proj / c_proj
        else if (fw == (EF_OUTF32 | EF_RES | EF_SCALE)) chunks4(std::integral_constant<uint32_t, EF_OUTF32 | EF_RES | EF_SCALE>{}, h);   // student proj / fc2
        else if (fw == (EF_RES | EF_RESBF | EF_F16 | EF_NT2 | EF_NT)) chunks4(std::integral_constant<uint32_t, EF_RES | EF_RESBF | EF_F16 | EF_NT2 | EF_NT>{}, h);   // teacher out_proj / c_proj, f16 stream
        else if (fw == (EF_RES | EF_RESBF | EF_F16)) chunks4(std::integral_constant<uint32_t, EF_RES | EF_RESBF | EF_F16>{}, h);
        else
        UNITE_HOT(UNITE_ACT_DGELU)                                           // fc2 input gradient x GELU'(z), bf16
        UNITE_HOT(UNITE_ACT_DGELU | EF_NT2 | EF_NT)
        UNITE_HOT(UNITE_ACT_GELU | EF_AUXOUT)                                // fc1 forward: z saved, a = GELU(z), bf16
        UNITE_HOT(UNITE_ACT_GELU | EF_AUXOUT | EF_NT2 | EF_NT)
        UNITE_HOT(UNITE_ACT_MULAUX)                                          // fc2 input gradient x the saved GELU'(z), bf16
        UNITE_HOT(UNITE_ACT_MULAUX | EF_NT2 | EF_NT)
        UNITE_HOT(UNITE_ACT_GELU_DSAVE | EF_AUXOUT)                          // fc1 forward: GELU'(z) saved, a = GELU(z), bf16
        UNITE_HOT(UNITE_ACT_GELU_DSAVE | EF_AUXOUT | EF_NT2 | EF_NT)
        UNITE_HOT(EF_OUTF32)                                                 // plain f32 products (weight gradients without split-K, ...)
        UNITE_HOT(EF_OUTF32 | EF_ACCUM)
        UNITE_HOT(EF_SPLIT | EF_OUTF32)                                      // split-K slices: raw slabs
        UNITE_HOT(EF_SPLIT | EF_OUTF32 | EF_ACCUM)
        UNITE_HOT(0u)                                                        // plain bf16
        UNITE_HOT(EF_NT2 | EF_NT)
        chunks(std::integral_constant<uint32_t, 0u>{}, std::integral_constant<uint32_t, 0u>{}, h, csum);
#undef UNITE_HOT
        if (h == 0) UNITE_STAMP(5);
        __syncthreads();
    }
    UNITE_STAMP(6);
    if (csum) {
        // this thread's sums cover its rows of column chunk ccol; the 4 * HALF / CPR threads of a column chunk meet in LDS and
        // leave ONE partial row per tile: colsum_partial[tile row][n] (summed over tile rows by colsum_rows_kernel, fixed order)
        constexpr int NG = 4 * HALF / CPR;
#pragma unroll
        for (int e = 0; e < 8; ++e) cs[(tid / CPR) * TILE + ccol + e] = csum_v[e];
        __syncthreads();
        if (tid < TILE && n0 + tid < g.N) {
            float a = 0.f;
#pragma unroll
            for (int r = 0; r < NG; ++r) a += cs[r * TILE + tid];
            p.colsum_partial[(size_t)(m0 / TILE) * g.N + n0 + tid] = a;
        }
    }
    // ---- row sums of op(A): the waves' partial vectors meet in LDS (k-step slots summed in a fixed order)
    const int rz_lo = g.rowsum_zero_lo, rz_hi = g.rowsum_zero_hi;
    if (WG >= 1 && rs_on) {
        float* rl = (float*)smem;                     // [2][TILE]: the epilogue image is dead behind its last barrier
        const int slot = WN == 4 ? (wn & 1) : 0;
        if (c16 < MT) {
#pragma unroll
            for (int r = 0; r < 4; ++r) rl[slot * TILE + rs_h * HALF + arow + c16 * 16 + 4 * G + r] = racc[r];
        }
        __syncthreads();
        if (tid < TILE && m0 + tid < gM) {
            const int gm = m0 + tid;
            float v = WN == 4 ? rl[tid] + rl[TILE + tid] : rl[tid];
            if (p.splitk > 1) __hip_atomic_store(p.rowsum_slab + (size_t)slice * gM + gm, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // sc1
            else {
                if (gm >= rz_lo && gm < rz_hi) v = 0.f;
                g.rowsum_a_out[gm] = g.rowsum_accumulate ? g.rowsum_a_out[gm] + v : v;
            }
        }
    }
    // ---- split-K: the slice of this tile that arrives last adds up the slabs (slice order: bitwise reproducible) and writes the tile
    if constexpr (WG < 2) return;      // (a split launch of these forms is always followed by splitk_finish_kernel)
    if (p.splitk > 1) {
        if (p.separate_reduce) return;
        if (!arrive_last<false>(p.counters + tile, (uint32_t)p.splitk, (volatile uint32_t*)smem)) return;      // slabs are written through
        const size_t mn = (size_t)gM * gN;
        const int S = p.splitk;
        // One workgroup reads S slabs of its tile: latency-bound unless many loads are in flight.  Batches of NB row chunks x up to SMAX
        // slices are requested together (16 loads of 16 bytes per thread: 64-128 KB per workgroup in flight), then added in slice order.
        constexpr int NCH = 2 * (TILE / 32);              // 8-column chunks per thread over the whole tile
        auto reduce = [&](auto smax_c, auto nb_c) {
            constexpr int SMAX = decltype(smax_c)::value, NB = decltype(nb_c)::value;
#pragma unroll 1
            for (int c0 = 0; c0 < NCH; c0 += NB) {
                f32x4 a0[NB], a1[NB];
                bool ok[NB];
                size_t base[NB], obase[NB];
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    const int ch = c0 + b, h = ch / (TILE / 32), e = ch % (TILE / 32);
                    const int lr = (tid + e * 4 * HALF) / CPR;
                    const int gm = m0 + h * HALF + lr, gn = n0 + ccol;
                    ok[b] = ch < NCH && gm < gM && gn < gN;
                    base[b] = (size_t)gm * gN + gn;
                    obase[b] = (size_t)gm * g.ldc + gn;
                    a0[b] = a1[b] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll 1
                for (int s0 = 0; s0 < S; s0 += SMAX) {
                    f32x4 v[NB][SMAX][2];
#pragma unroll
                    for (int b = 0; b < NB; ++b)
#pragma unroll
                        for (int sl = 0; sl < SMAX; ++sl)
                            if (ok[b] && s0 + sl < S) {
                                const float* sp = p.slab + (size_t)(s0 + sl) * mn + base[b];
                                v[b][sl][0] = *(const f32x4*)sp;
                                v[b][sl][1] = *(const f32x4*)(sp + 4);
                            }
#pragma unroll
                    for (int b = 0; b < NB; ++b)
#pragma unroll
                        for (int sl = 0; sl < SMAX; ++sl)
                            if (ok[b] && s0 + sl < S) {
                                if (s0 + sl == 0) { a0[b] = v[b][sl][0]; a1[b] = v[b][sl][1]; }      // the first slab starts the sum: ((s0 + s1) + s2) + ...
                                else {
#pragma unroll
                                    for (int q = 0; q < 4; ++q) { a0[b][q] += v[b][sl][0][q]; a1[b][q] += v[b][sl][1][q]; }
                                }
                            }
                }
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    if (!ok[b]) continue;
                    float* op = (float*)g.out + obase[b];
                    if (g.accumulate) {
                        const f32x4 o0 = *(const f32x4*)op, o1 = *(const f32x4*)(op + 4);
#pragma unroll
                        for (int q = 0; q < 4; ++q) { a0[b][q] += o0[q]; a1[b][q] += o1[q]; }
                    }
                    *(f32x4*)op = a0[b];
                    *(f32x4*)(op + 4) = a1[b];
                }
            }
        };
        if (S <= 4) reduce(std::integral_constant<int, 4>{}, std::integral_constant<int, 2>{});
        else reduce(std::integral_constant<int, 8>{}, std::integral_constant<int, 1>{});
        if (rs_on && tid < TILE && m0 + tid < gM) {
            const int gm = m0 + tid;
            float v = p.rowsum_slab[gm];
            for (int sl = 1; sl < S; ++sl) v += p.rowsum_slab[(size_t)sl * gM + gm];
            if (gm >= rz_lo && gm < rz_hi) v = 0.f;
            g.rowsum_a_out[gm] = g.rowsum_accumulate ? g.rowsum_a_out[gm] + v : v;
        }
    }
}

// =====================================================================================================================
// "Wide" kernel: 128 x 256 x 64 tile, 4 waves (2 x 2), TWO workgroups per CU (64 KiB of LDS each), so one workgroup's
// store-heavy epilogue overlaps the other's MFMA main loop.  Same quadrant schedule as the deep kernel (16 MFMAs per phase
// and wave: A half = 64 rows -> 2 m-tiles, B half = 128 columns -> 4 n-tiles per wave), but with a SINGLE K-tile of LDS
// (A0 A1 8 KiB each, B0 B1 16 KiB each): a half-tile slot is refilled for the next K-tile right after the phase that read
// it, i.e. every load is in flight for three phases:
//   P1: wait vmcnt(6)  reads A0,B0            MFMA (0,0)
//   P2: wait vmcnt(2)  reads B1   issues A0',B0'   MFMA (0,1)
//   P3: wait vmcnt(6)  reads A1   issues B1'       MFMA (1,1)
//   P4:                           issues A1'       MFMA (1,0)      (each phase starts with a raw s_barrier)
// =====================================================================================================================
constexpr int WIDE_LDS = 64 * 256 * 4;       // 64 KiB: the ring uses 48 KiB, the epilogue image (64 rows x 256 f32) all of it

template <bool TA, bool TB, bool DIRECT>
__global__ __launch_bounds__(256, 2) void gemm_wide_kernel(const Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TM = 128, TN = 256;
    const unite_gemm_args& g = p.a;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    const int nbn = (g.N + TN - 1) / TN, nbm = (g.M + TM - 1) / TM, nb = nbm * nbn, nbt = nb * p.splitk;
    const int bid = blockIdx.x, xcd = bid & 7, qq = nbt >> 3, rr = nbt & 7;
    const int lin = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
    const int tile = lin % nb, slice = lin / nb;
    const int m0 = (tile / nbn) * TM, n0 = (tile % nbn) * TN;
    const int k_begin = slice * p.k_chunk, k_end = min(g.K, k_begin + p.k_chunk);
    const int nk = (k_end - k_begin + BK - 1) / BK;

    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)g.A, 0, (int)p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)g.B, 0, (int)p.b_bytes, 0x00020000);
    char* const A0 = smem;
    char* const A1 = smem + 8192;
    char* const B0 = smem + 16384;
    char* const B1 = smem + 32768;

    auto issueA = [&](char* slot, int h, int u) {      // 64 rows x 64 k = 8 pieces, 2 per wave
        const int k0 = k_begin + u * BK;
#pragma unroll
        for (int i = 0; i < 2; ++i) stage_piece<TA, 64>(rsA, slot, wave * 2 + i, m0 + h * 64, k0, g.M, k_end, g.lda, lane);
    };
    auto issueB = [&](char* slot, int nh, int u) {     // 128 cols x 64 k = 16 pieces, 4 per wave
        const int k0 = k_begin + u * BK;
#pragma unroll
        for (int i = 0; i < 4; ++i) stage_piece<TB, 128, DIRECT>(rsB, slot, wave * 4 + i, n0 + nh * 128, k0, g.N, k_end, g.ldb, lane);
    };

    f32x4 acc[2][2][2][4];       // [A half][m-tile][B half][n-tile]
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[h][i][nh][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    issueA(A0, 0, 0);
    issueB(B0, 0, 0);
    issueB(B1, 1, 0);
    issueA(A1, 1, 0);
    f32x4 bias0, bias1;
    if (!DIRECT) load_bias8(g, n0 + (tid & 31) * 8, bias0, bias1);

    const int arow = wm * 32, bcol = wn * 64;
    for (int t = 0; t < nk; ++t) {
        bf16x8 af[2][2], b0f[4][2], b1f[4][2];
        // ---- P1
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) af[i][ks] = load_frag_h<TA, 64>(A0, arow + i * 16, ks, lane);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                b0f[j][ks] = DIRECT ? load_bfrag_perm<TB>(B0, bcol, j, ks, lane) : load_frag_h<TB, 128>(B0, bcol + j * 16, ks, lane);
        lds_tr_fence<TA || TB>();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[0][i][0][j] = DIRECT ? mfma16(b0f[j][ks], af[i][ks], acc[0][i][0][j]) : mfma16(af[i][ks], b0f[j][ks], acc[0][i][0][j]);
        __builtin_amdgcn_s_setprio(0);
        // ---- P2
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                b1f[j][ks] = DIRECT ? load_bfrag_perm<TB>(B1, bcol, j, ks, lane) : load_frag_h<TB, 128>(B1, bcol + j * 16, ks, lane);
        issueA(A0, 0, t + 1);
        issueB(B0, 0, t + 1);
        lds_tr_fence<TA || TB>();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[0][i][1][j] = DIRECT ? mfma16(b1f[j][ks], af[i][ks], acc[0][i][1][j]) : mfma16(af[i][ks], b1f[j][ks], acc[0][i][1][j]);
        __builtin_amdgcn_s_setprio(0);
        // ---- P3
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) af[i][ks] = load_frag_h<TA, 64>(A1, arow + i * 16, ks, lane);
        issueB(B1, 1, t + 1);
        lds_tr_fence<TA || TB>();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[1][i][1][j] = DIRECT ? mfma16(b1f[j][ks], af[i][ks], acc[1][i][1][j]) : mfma16(af[i][ks], b1f[j][ks], acc[1][i][1][j]);
        __builtin_amdgcn_s_setprio(0);
        // ---- P4 (the barrier orders every wave's phase-3 reads of A1 before its refill)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        issueA(A1, 1, t + 1);
        lds_tr_fence<TA || TB>();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[1][i][0][j] = DIRECT ? mfma16(b0f[j][ks], af[i][ks], acc[1][i][0][j]) : mfma16(af[i][ks], b0f[j][ks], acc[1][i][0][j]);
        __builtin_amdgcn_s_setprio(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (p.debug_skip == 1) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                    for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(acc[h][i][nh][j]));
        return;
    }

    const int G = lane >> 4, c16 = lane & 15;
    if (DIRECT) {
        // ---- epilogue from the registers: lane (G, c16) owns row c16 of each m-tile and 16 consecutive columns per B half
        f32x4 bb[2][4];
#pragma unroll
        for (int nh = 0; nh < 2; ++nh) {
            load_bias8(g, n0 + nh * 128 + bcol + 16 * G, bb[nh][0], bb[nh][1]);
            load_bias8(g, n0 + nh * 128 + bcol + 16 * G + 8, bb[nh][2], bb[nh][3]);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int gm = m0 + h * 64 + arow + i * 16 + c16;
#pragma unroll
                for (int nh = 0; nh < 2; ++nh) {
                    const int gn = n0 + nh * 128 + bcol + 16 * G;
                    if (gm < g.M && gn < g.N) epilogue_chunk(p, g, slice, gm, gn, acc[h][i][nh][0], acc[h][i][nh][1], bb[nh][0], bb[nh][1]);
                    if (gm < g.M && gn + 8 < g.N) epilogue_chunk(p, g, slice, gm, gn + 8, acc[h][i][nh][2], acc[h][i][nh][3], bb[nh][2], bb[nh][3]);
                }
            }
        return;
    }
    // ---- epilogue: 64 rows x 256 columns per pass through an f32 image, 8 chunks of 8 columns per thread
    float* cs = (float*)smem;
    const int ccol = (tid & 31) * 8;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        cs[(arow + i * 16 + 4 * G + r) * TN + ((nh * 128 + bcol + j * 16 + c16) ^ (G << 4))] = acc[h][i][nh][j][r];
        __syncthreads();
#pragma unroll 2
        for (int e = 0; e < 8; ++e) {
            const int lr = (tid >> 5) + e * 8;
            const int gm = m0 + h * 64 + lr, gn = n0 + ccol;
            const float* cp = cs + lr * TN + (ccol ^ (((lr >> 2) & 3) << 4));
            if (gm < g.M && gn < g.N) epilogue_chunk(p, g, slice, gm, gn, *(const f32x4*)cp, *(const f32x4*)(cp + 4), bias0, bias1);
        }
        __syncthreads();
    }
}

// Split-K, second launch (separate_reduce): out[m,n] (+)= sum_s slab[s][m,n] and rowsum_out[m] (+)= sum_s rowsum_slab[s][m], slice order.
// A wide HBM-bound pass over S * M * N floats: used where one tile's slabs are too much for its last workgroup to add up on its own (256^2
// tiles: S x 256 KiB read serially while the CU is held) -- the in-launch form is the same sum in the same order, so both are bit-identical.
__global__ __launch_bounds__(256) void splitk_finish_kernel(const float* __restrict__ slab, const float* __restrict__ rowsum_slab, int splitk, int M,
                                                            int N, float* __restrict__ out, int ldc, int accumulate,
                                                            float* __restrict__ rowsum_out, int rs_accumulate, int rz_lo, int rz_hi) {
    const size_t n4 = (size_t)N / 4, total = (size_t)M * n4, mn = (size_t)M * N;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t m = i / n4, c = (i % n4) * 4;
        f32x4 a = *(const f32x4*)(slab + m * N + c);
        for (int s = 1; s < splitk; ++s) {
            const f32x4 b = *(const f32x4*)(slab + (size_t)s * mn + m * N + c);
            a = (f32x4){a[0] + b[0], a[1] + b[1], a[2] + b[2], a[3] + b[3]};
        }
        float* op = out + m * ldc + c;
        if (accumulate) {
            const f32x4 o = *(const f32x4*)op;
            a = (f32x4){a[0] + o[0], a[1] + o[1], a[2] + o[2], a[3] + o[3]};
        }
        *(f32x4*)op = a;
    }
    if (rowsum_out) {
        for (int gm = blockIdx.x * 256 + threadIdx.x; gm < M; gm += gridDim.x * 256) {
            float v = rowsum_slab[gm];
            for (int sl = 1; sl < splitk; ++sl) v += rowsum_slab[(size_t)sl * M + gm];
            if (gm >= rz_lo && gm < rz_hi) v = 0.f;
            rowsum_out[gm] = rs_accumulate ? rowsum_out[gm] + v : v;
        }
    }
}

// colsum_out[n] (+)= sum over tile rows of the per-tile column sums the deep kernels' epilogue left (fixed order: bitwise reproducible)
__global__ __launch_bounds__(256) void colsum_rows_kernel(const float* __restrict__ partial, int nparts, int N, float* __restrict__ out,
                                                          int accumulate) {
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), r = threadIdx.x >> 6;
    float a = 0.f;
    if (c < N)
        for (int i = r; i < nparts; i += 4) a += partial[(size_t)i * N + c];
    red[r][threadIdx.x & 63] = a;
    __syncthreads();
    if (r == 0 && c < N) {
        const int l = threadIdx.x;
        a = (red[0][l] + red[1][l]) + (red[2][l] + red[3][l]);
        out[c] = accumulate ? out[c] + a : a;
    }
}

// ---- launch of the deep tile kernels: HALF x weight-gradient form x layouts x schedule -> one instantiation
template <int HALF, bool TA, bool TB, int WG, int SCHED, int EPI = 0>
int launch_deep_one(int nb, hipStream_t s, const Params& p) {
    constexpr int lds = 8 * HALF * 128;
    auto* k = gemm_deep_kernel<HALF, TA, TB, WG, SCHED, EPI>;
    if (lds > 64 * 1024) {      // 128 KiB of dynamic LDS needs the opt-in, once per instantiation
        static bool ok = false;
        if (!ok) {
            hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            if (e != hipSuccess) return (int)e;
            ok = true;
        }
    }
    hipLaunchKernelGGL(k, dim3(nb), dim3(4 * HALF), lds, s, p);
    return UNITE_OK;
}
template <int HALF, int WG, int SCHED>
int launch_deep_layout(bool ta, bool tb, int nb, hipStream_t s, const Params& p) {
    if (!ta && !tb) return launch_deep_one<HALF, false, false, WG, SCHED>(nb, s, p);
    if (!ta && tb) return launch_deep_one<HALF, false, true, WG, SCHED>(nb, s, p);
    if (ta && !tb) return launch_deep_one<HALF, true, false, WG, SCHED>(nb, s, p);
    return launch_deep_one<HALF, true, true, WG, SCHED>(nb, s, p);
}
// the 256^2 kernel with the transposed-accumulator bf16 epilogue (EPI 1): schedule 1, plain form
inline int launch_deep_epi1(bool ta, bool tb, int nb, hipStream_t s, const Params& p) {
    if (!ta && !tb) return launch_deep_one<128, false, false, 0, 1, 1>(nb, s, p);
    if (!ta && tb) return launch_deep_one<128, false, true, 0, 1, 1>(nb, s, p);
    if (ta && !tb) return launch_deep_one<128, true, false, 0, 1, 1>(nb, s, p);
    return launch_deep_one<128, true, true, 0, 1, 1>(nb, s, p);
}
template <int HALF>
int launch_deep(int wgf, bool ta, bool tb, int sched, int nb, hipStream_t s, const Params& p) {
    if (wgf == 2) return launch_deep_one<HALF, true, true, 2, 0>(nb, s, p);      // in-launch reduction: weight-gradient layout only (UNITE_SPLITK_SEPARATE=0)
    if (wgf == 1) return sched ? launch_deep_layout<HALF, 1, 1>(ta, tb, nb, s, p) : launch_deep_layout<HALF, 1, 0>(ta, tb, nb, s, p);
    return sched ? launch_deep_layout<HALF, 0, 1>(ta, tb, nb, s, p) : launch_deep_layout<HALF, 0, 0>(ta, tb, nb, s, p);
}
// UNITE_GEMM_EPI = 0 | 1: process default of the 256^2 kernel's epilogue form for the outputs EPI 1 can take (per call: plan_flags bits 4, 5)
inline int g_epi_env() {
    static const int v = getenv("UNITE_GEMM_EPI") ? atoi(getenv("UNITE_GEMM_EPI")) : 1;      // default 1: sustained -1 .. -8 %, the step -0.6 % (round 4)
    return v != 0;
}
// UNITE_GEMM_SCHED = 0 | 1: process default of the main-loop schedule (per call: plan_flags bits 2, 3)
inline int g_sched_env() {
    static const int v = getenv("UNITE_GEMM_SCHED") ? atoi(getenv("UNITE_GEMM_SCHED")) : 1;
    return v != 0;
}

inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// Joint choice of kernel and split-K factor from a cost model fitted on MI355X (us):
//   time(kind, S) = rounds(tiles(kind) * S over resident workgroups) * (ceil(K-tiles / S) * c + e) + slab reduction,
// kind 1: 128^2 tile, 512 resident (2 / CU), (c, e) = (1.006, 4.4);  kind 2: 256^2, 256 resident, (1.51, 9.8);
// kind 3: 128 x 256, 512 resident, (1.58, 9.0).  S > 1 only for plain f32 outputs with a workspace (weight gradients).
struct Plan { int kind, splitk; double cost; };
// 0 .. 1: weight of the CU time a launch takes against its latency when the planner picks tile size and split-K: the PROCESS DEFAULT
// (unite_gemm_set_sharing; UNITE_GEMM_PLAN_WORK pins it for experiments).  A launch that carries its own hint (unite_gemm_args.plan_*)
// does not read it.
double g_plan_work = getenv("UNITE_GEMM_PLAN_WORK") ? atof(getenv("UNITE_GEMM_PLAN_WORK")) : 0.0;

// split-K slabs are summed inside the launch by the last slice of each tile: ONE workgroup reads S write-through slabs of its tile, a few
// memory round trips of 16 loads per thread each -- measured (tools/wgrad_time.py sweeps, round 3) 2.4 us per 128^2 slab (64 KB) and
// 6.5 us per 256^2 slab (256 KB), i.e. 27-40 GB/s for that one workgroup; the other slices' workgroups have left their CUs by then
inline double splitk_tail_us(int kind, int S) { return 1.5 + S * (kind == 2 ? 6.5 : 2.4); }

// `tt`: both operands k-strided (weight gradients dY^T X): transposing fragment reads for A and B; fitted on the K = 10 240 sweeps of
// tools/wgrad_time.py (128^2: 0.76 us per K-tile with one workgroup per CU, 1.15 with two; 256^2: 1.64)
inline Plan plan_gemm(int M, int N, int K, bool can_split, size_t ws_bytes, int only_kind, bool deep_only, double work_weight, bool tt = false) {
    const int tiles[4] = {0, ((M + 127) / 128) * ((N + 127) / 128), ((M + 255) / 256) * ((N + 255) / 256), ((M + 127) / 128) * ((N + 255) / 256)};
    const int slots[4] = {0, 512, 256, 512};
    double cc[4] = {0, 1.006, 1.51, 1.58}, ee[4] = {0, 4.4, 9.8, 9.0};
    // Cost model of products that split K.  2 (default): generic constants, the slabs streamed once more by a second launch.  3: constants fitted on
    // stand-alone sweeps of the weight-gradient layout with the in-launch reduction's serial tail (tools/wgrad_time.py).  Model 3 predicts the
    // stand-alone launch better and gives the SLOWER step (20.60 vs 20.44 ms, same call: it trades CU time of the side stream's launches for
    // latency nobody waits for); UNITE_PLAN_MODEL=3 keeps it for A/B runs.
    static const int model = getenv("UNITE_PLAN_MODEL") ? atoi(getenv("UNITE_PLAN_MODEL")) : 2;
    if (model == 2) tt = false;
    if (tt) { cc[1] = 1.15; cc[2] = 1.64; }
    const int kt = (K + BK - 1) / BK;
    Plan best = {1, 1, 1e30};
    for (int kind = 1; kind <= 3; ++kind) {
        if (only_kind && kind != only_kind) continue;
        if (deep_only && kind == 3) continue;
        for (int S = 1; S <= 16; ++S) {
            // split-K: deep kernels only (they carry the in-launch reduction), one counter per tile in the workspace header
            if (S > 1 && (!can_split || kind == 3 || tiles[kind] > 4096 || (size_t)S * M * N * sizeof(float) > ws_bytes || kt / S < 4)) break;
            const int kts = (kt + S - 1) / S;
            const double rounds = (double)((tiles[kind] * S + slots[kind] - 1) / slots[kind]);
            const double c_eff = (tt && kind == 1 && tiles[kind] * S <= 256) ? 0.76 : cc[kind];      // a 128^2 workgroup alone on its CU
            double cost = rounds * (kts * c_eff + ee[kind]);
            const double stream_us = (double)S * M * N * 8.0 / 5.0e6;       // slab write + read at ~5 TB/s
            if (S > 1) cost += model == 2 ? 3.0 + stream_us : splitk_tail_us(kind, S);
            // work_weight > 0: the launch shares the GPU with an independent stream (teacher one batch ahead), so what it costs the step is
            // less its own latency than the CU time it takes: workgroups x time each, over the resident slots (+ the reducing workgroups' tails)
            if (work_weight > 0.0) {
                double work = (double)tiles[kind] * S * (kts * cc[kind] + ee[kind]) / slots[kind];      // CU time at full occupancy
                if (S > 1) work += model == 2 ? stream_us : (double)tiles[kind] * splitk_tail_us(kind, S) / slots[kind];
                cost = (1.0 - work_weight) * cost + work_weight * work;
            }
            if (cost < best.cost - 1e-9) best = {kind, S, cost};
        }
    }
    return best;
}

// launch-timing pool for bench.py's roofline leg (off by default; the only global state of this file)
struct ProfState {
    bool on = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
    size_t used = 0;
    double flops = 0.0;
    double bytes = 0.0;           // algorithmic HBM bytes of the timed launches: every operand / output / residual / aux matrix once
};
ProfState g_prof;

// operands once, outputs once, f32 or bf16 residual / saved pre-activations / old output once: the floor the PMC traffic is compared with
inline double algo_bytes(const unite_gemm_args& g) {
    const double mn = (double)g.M * g.N;
    double b = 2.0 * ((double)g.M * g.K + (double)g.N * g.K) + mn * (g.out_f32 ? 4.0 : 2.0);
    if (g.bias) b += 4.0 * g.N;
    if (g.residual) b += mn * (g.residual_bf16 ? 2.0 : 4.0);
    if ((g.act == UNITE_ACT_DGELU || g.act == UNITE_ACT_MULAUX) && g.aux_in) b += mn * 2.0;
    if ((g.act == UNITE_ACT_GELU || g.act == UNITE_ACT_GELU_DSAVE) && g.aux_out) b += mn * 2.0;
    if (g.accumulate) b += mn * 4.0;
    if (g.out_bf16_copy) b += mn * 2.0;
    return b;
}

}  // namespace

namespace {
// argument checks shared by the single and the grouped entry point; returns the operand extents for the buffer descriptors
int check_problem(const unite_gemm_args& g, int64_t& a_bytes, int64_t& b_bytes) {
    if (g.M <= 0 || g.N <= 0 || g.K <= 0 || !g.A || !g.B || !g.out) return UNITE_EINVAL;
    if ((g.lda & 7) || (g.ldb & 7) || (g.N & 7) || (g.ldc & 7)) return UNITE_EINVAL;
    if (!aligned16(g.A) || !aligned16(g.B) || !aligned16(g.out)) return UNITE_EINVAL;
    if ((g.trans_a ? (g.M & 7) : (g.K & 7)) || (g.trans_b ? 0 : (g.K & 7))) return UNITE_EINVAL;
    if (g.accumulate && !g.out_f32) return UNITE_EINVAL;
    if (g.act < UNITE_ACT_NONE || g.act > UNITE_ACT_MULAUX) return UNITE_EINVAL;
    if ((g.act == UNITE_ACT_DGELU || g.act == UNITE_ACT_MULAUX) && (!g.aux_in || (g.ld_aux_in & 7) || !aligned16(g.aux_in))) return UNITE_EINVAL;
    if (g.act == UNITE_ACT_GELU_DSAVE && !g.aux_out) return UNITE_EINVAL;
    if (g.aux_out && (g.ld_aux_out & 7)) return UNITE_EINVAL;
    if (g.row_scale && g.rows_per_scale <= 0) return UNITE_EINVAL;
    if (g.residual && ((g.ldr & (g.residual_bf16 ? 7 : 3)) || !aligned16(g.residual))) return UNITE_EINVAL;
    if (g.residual_bf16 < 0 || g.residual_bf16 > 2) return UNITE_EINVAL;
    // the f16 stream: f16 residual rows in, f16 rows out -- nothing else of the epilogue knows that type
    if (g.residual_bf16 == 2 && (!g.residual || g.out_f32 || g.out_bf16_copy || g.colsum_out || g.act != UNITE_ACT_NONE)) return UNITE_EINVAL;
    if ((g.plan_flags & 1) && (g.plan_persistent < 0 || g.plan_persistent > 2)) return UNITE_EINVAL;
    if (g.out_bf16_copy && (g.ld_copy & 7)) return UNITE_EINVAL;
    const int64_t a_rows = g.trans_a ? g.K : g.M, a_cols = g.trans_a ? g.M : g.K;
    const int64_t b_rows = g.trans_b ? g.K : g.N, b_cols = g.trans_b ? g.N : g.K;
    a_bytes = ((a_rows - 1) * g.lda + a_cols) * 2;
    b_bytes = ((b_rows - 1) * g.ldb + b_cols) * 2;
    if (a_bytes >= (int64_t)OOB_OFFSET || b_bytes >= (int64_t)OOB_OFFSET) return UNITE_ENOSUP;
    return UNITE_OK;
}
}  // namespace

// Several independent problems with the same operand layouts in ONE launch of the 128 x 128 deep kernel: the four weight
// gradients of a transformer block (K = tokens, outputs of 36..144 tiles each) fill the chip together (432 tiles over 512
// resident workgroups) instead of each going through split-K slabs and a reduction pass.
extern "C" int unite_gemm_bf16_grouped(const unite_gemm_args* args, int32_t count, void* stream) {
    if (!args || count <= 0 || count > MAX_GROUP) return UNITE_EINVAL;
    if (count == 1) return unite_gemm_bf16(args, stream);
    Params p;
    memset(&p, 0, sizeof(p));
    int tiles = 0, kmax = 0;
    for (int i = 0; i < count; ++i) {
        const unite_gemm_args& g = args[i];
        int64_t ab, bb;
        const int rc = check_problem(g, ab, bb);
        if (rc != UNITE_OK) return rc;
        if (g.trans_a != args[0].trans_a || g.trans_b != args[0].trans_b) return UNITE_EINVAL;
        // the grouped launch runs the plain kernel form: the fused bias sums (row sums of op(A), column sums of the output) are products of the
        // single-problem launch only -- refuse them instead of returning UNITE_OK with the sums unwritten
        if (g.rowsum_a_out || g.colsum_out) return UNITE_EINVAL;
        if (i == 0) { p.a = g; p.a_bytes = (uint32_t)ab; p.b_bytes = (uint32_t)bb; }
        else { p.more[i - 1] = g; p.more_a_bytes[i - 1] = (uint32_t)ab; p.more_b_bytes[i - 1] = (uint32_t)bb; }
        tiles += ((g.M + 127) / 128) * ((g.N + 127) / 128);
        p.tile_end[i] = tiles;
        kmax = g.K > kmax ? g.K : kmax;
    }
    p.ngroups = count;
    p.splitk = 1;
    p.k_chunk = (kmax + BK - 1) / BK * BK;
    p.slab = nullptr;
    static const int dbg = getenv("UNITE_GEMM_DEBUG_SKIP") ? atoi(getenv("UNITE_GEMM_DEBUG_SKIP")) : 0;
    p.debug_skip = dbg;
    p.nt_store = 0;
    p.colsum_partial = nullptr;
    hipStream_t s = (hipStream_t)stream;
    const bool prof = g_prof.on && g_prof.used < g_prof.ev.size();
    if (prof) (void)hipEventRecord(g_prof.ev[g_prof.used].first, s);
    const bool ta = args[0].trans_a, tb = args[0].trans_b;
    if (!ta && !tb) hipLaunchKernelGGL((gemm_deep_kernel<64, false, false>), dim3(tiles), dim3(256), 8 * 64 * 128, s, p);
    else if (!ta && tb) hipLaunchKernelGGL((gemm_deep_kernel<64, false, true>), dim3(tiles), dim3(256), 8 * 64 * 128, s, p);
    else if (ta && !tb) hipLaunchKernelGGL((gemm_deep_kernel<64, true, false>), dim3(tiles), dim3(256), 8 * 64 * 128, s, p);
    else hipLaunchKernelGGL((gemm_deep_kernel<64, true, true>), dim3(tiles), dim3(256), 8 * 64 * 128, s, p);
    if (prof) {
        (void)hipEventRecord(g_prof.ev[g_prof.used].second, s);
        g_prof.used++;
        for (int i = 0; i < count; ++i) {
            g_prof.flops += 2.0 * args[i].M * args[i].N * args[i].K;
            g_prof.bytes += algo_bytes(args[i]);
        }
    }
    UNITE_LAUNCH_CHECK();
    return UNITE_OK;
}

// gemm_pp.hip: persistent 256 x 128 kernel with the epilogue of one tile hidden under the main loop of the next
namespace {
int g_pp_policy = -1;      // -1: UNITE_GEMM_PP (default 1); set by unite_gemm_set_policy (tests / A-B runs in one process)
int g_pp_env() {
    static const int v = getenv("UNITE_GEMM_PP") ? atoi(getenv("UNITE_GEMM_PP")) : 1;
    return v;
}
}  // namespace
extern "C" int unite_gemm_set_policy(int32_t persistent) {
    if (persistent < -1 || persistent > 2) return UNITE_EINVAL;
    g_pp_policy = persistent;
    return UNITE_OK;
}
extern "C" int unite_gemm_get_policy(void) { return g_pp_policy; }
extern "C" int unite_gemm_set_sharing(float work_weight) {
    if (!(work_weight >= 0.f && work_weight <= 1.f)) return UNITE_EINVAL;
    static const bool pinned = getenv("UNITE_GEMM_PLAN_WORK") != nullptr;
    if (!pinned) g_plan_work = (double)work_weight;
    return UNITE_OK;
}
extern "C" float unite_gemm_get_sharing(void) { return (float)g_plan_work; }

// What the planner would choose for a product that may split K (plain f32 output, a workspace of `slab_bytes` for the slabs): tile kernel
// (1: 128^2, 2: 256^2, 3: 128 x 256) and split factor.  Host arithmetic only -- no device is touched -- so the planner can be inspected and
// pinned by tests on a box without a GPU.
extern "C" int unite_gemm_plan(int32_t M, int32_t N, int32_t K, int32_t trans_a, int32_t trans_b, float sharing, int64_t slab_bytes, int32_t rowsum,
                               int32_t* kind, int32_t* splitk) {
    if (M <= 0 || N <= 0 || K <= 0 || !kind || !splitk || !(sharing >= 0.f && sharing <= 1.f) || slab_bytes < 0) return UNITE_EINVAL;
    const bool deep_only = rowsum != 0;
    const Plan plan = plan_gemm(M, N, K, slab_bytes > 0, (size_t)slab_bytes, 0, deep_only, (double)sharing, trans_a && trans_b);
    *kind = plan.kind;
    *splitk = plan.splitk;
    return UNITE_OK;
}
int unite_gemm_pp_supported(const unite_gemm_args& g);
int unite_gemm_pp_launch(const unite_gemm_args& g, int64_t a_bytes, int64_t b_bytes, hipStream_t s);

extern "C" int unite_gemm_bf16(const unite_gemm_args* args, void* stream) {
    if (!args) return UNITE_EINVAL;
    const unite_gemm_args& g = *args;
    int64_t a_bytes, b_bytes;
    {
        const int rc = check_problem(g, a_bytes, b_bytes);
        if (rc != UNITE_OK) return rc;
    }
    {
        // The persistent kernel (gemm_pp.hip) takes the shapes it measured faster on than the tile kernels below (MI355X, round 2, both
        // in one process: profiles/r02_gemm_ab.txt): k-contiguous B with an f32 output (proj / fc2 / out_proj / c_proj / decoder heads:
        // -5 .. -21 %, the residual epilogue hides under the next tile and whole 128-byte lines move) and the student's plain and GELU
        // bf16 products (qkv -14 %, fc1 -3 %).  The teacher's 310-MB bf16 outputs (QuickGELU c_fc, qkv: +1 .. +5 %) and the k-strided B of
        // the input gradients (+10 .. +18 %: transposing fragment reads with rebuilt addresses) stay on the tile kernels.
        // UNITE_GEMM_PP = 0 never / 1 measured shapes (default) / 2 whenever supported; UNITE_GEMM_PP_MIN_TILES moves the size floor.
        const int pp = (g.plan_flags & 1) ? g.plan_persistent : (g_pp_policy >= 0 ? g_pp_policy : g_pp_env());
        static const int pp_min = getenv("UNITE_GEMM_PP_MIN_TILES") ? atoi(getenv("UNITE_GEMM_PP_MIN_TILES")) : 64;
        static const char* force_k = getenv("UNITE_GEMM_KERNEL");
        // a launch of 257 .. ~1000 tiles gives the 256 persistent workgroups 1-4 tiles each: badly balanced, most of the epilogues unhidden,
        // and -- unlike tile kernels, whose workgroups retire tile by tile -- nothing from another stream can slip in meanwhile (the
        // teacher runs three frame ranges on three streams: 396 tiles per launch; measured +0.5 ms per step there).  UNITE_GEMM_PP_BAND=lo,hi
        static const char* band_s = getenv("UNITE_GEMM_PP_BAND");
        static const int band_lo = band_s ? atoi(band_s) : 256, band_hi = (band_s && strchr(band_s, ',')) ? atoi(strchr(band_s, ',') + 1) : 1000;
        const int pp_tiles = ((g.M + 255) / 256) * ((g.N + 127) / 128);
        bool measured = !g.trans_b && (pp_tiles <= band_lo || pp_tiles >= band_hi) &&
                        (g.out_f32 || (g.act != UNITE_ACT_QUICKGELU && (int64_t)g.M * g.N <= (int64_t)32 << 20));
        // Round 4, in-step A/B (profiles/r04_clock_notes.txt section 20): the student's fc2 at B = 32 (10 240 x 768, K = 3072: 240 persistent tiles, every
        // CU held for the whole launch) is better off on the tile kernels, which at the step's sharing weight take 120 CUs and leave the rest to the
        // teacher's stream: -0.10 ms per step (3 of 3 rounds); ViT-L's fc2 (5120 x 1024, K = 4096: 160 tiles) the other way round (+0.2 ms of 32.6).
        // Tried in the same call and left alone: the teacher's QuickGELU c_fc per frame range on the persistent kernel (equal), the qkv input
        // gradient on it (+0.2 ms).  UNITE_GEMM_PP_FC2=0 switches the rule off.
        static const bool fc2_rule = !getenv("UNITE_GEMM_PP_FC2") || atoi(getenv("UNITE_GEMM_PP_FC2")) != 0;
        if (fc2_rule && g.out_f32 && g.K >= 2048 && pp_tiles > 192 && pp_tiles <= band_lo) measured = false;
        if (pp && (pp == 2 || measured) && !force_k && !g.rowsum_a_out && ((g.M + 255) / 256) * ((g.N + 127) / 128) >= pp_min && unite_gemm_pp_supported(g)) {
            hipStream_t s = (hipStream_t)stream;
            const bool prof = g_prof.on && g_prof.used < g_prof.ev.size();
            if (prof) (void)hipEventRecord(g_prof.ev[g_prof.used].first, s);
            const int rc = unite_gemm_pp_launch(g, a_bytes, b_bytes, s);
            if (prof) {
                (void)hipEventRecord(g_prof.ev[g_prof.used].second, s);
                g_prof.used++;
                g_prof.flops += 2.0 * g.M * g.N * g.K;
                g_prof.bytes += algo_bytes(g);
            }
            return rc;
        }
    }
    Params p;
    memset(&p, 0, sizeof(p));
    p.a = g;
    p.a_bytes = (uint32_t)a_bytes;
    p.b_bytes = (uint32_t)b_bytes;
    p.ngroups = 1;
    // kernel + split-K choice; UNITE_GEMM_KERNEL = simple | deep128 | deep256 | wide pins the kernel (A/B experiments)
    static const char* force = getenv("UNITE_GEMM_KERNEL");
    const int only = !force ? 0 : !strcmp(force, "deep256") ? 2 : !strcmp(force, "wide") ? 3 : 1;
    const bool plain = g.out_f32 && !g.bias && g.act == UNITE_ACT_NONE && !g.row_scale && !g.residual && !g.out_bf16_copy;
    // workspace: [header: tile arrival counters][per-slice row sums of op(A), 16 x M floats][slabs]
    const bool want_rowsum = g.rowsum_a_out != nullptr;
    const size_t rs_bytes = want_rowsum ? (((size_t)16 * g.M * sizeof(float) + 255) & ~(size_t)255) : 0;
    const size_t ws_head = UNITE_WS_HEADER_BYTES + rs_bytes;
    const bool can_split = plain && g.workspace && aligned16(g.workspace) && (size_t)g.workspace_bytes > ws_head;
    // column sums of the output come out of the deep kernels' epilogue (per-tile-row partials in the workspace, then one small
    // reduction): no split-K, no wide / simple kernel for such a product
    const bool want_colsum = g.colsum_out != nullptr;
    if (want_colsum && (!g.workspace || !aligned16(g.workspace) ||
                        (size_t)g.workspace_bytes < unite_gemm_colsum_workspace(g.M, g.N) || (((uintptr_t)g.colsum_out) & 3))) return UNITE_EINVAL;
    if (want_rowsum && (want_colsum || (((uintptr_t)g.rowsum_a_out) & 3) || g.rowsum_zero_lo > g.rowsum_zero_hi)) return UNITE_EINVAL;
    const bool deep_only = want_colsum || want_rowsum;
    const double work_weight = (g.plan_flags & 2) ? (double)g.plan_sharing : g_plan_work;
    if (!(work_weight >= 0.0 && work_weight <= 1.0)) return UNITE_EINVAL;
    Plan plan = plan_gemm(g.M, g.N, g.K, can_split && !want_colsum, can_split ? (size_t)g.workspace_bytes - ws_head : 0,
                          deep_only && only != 2 ? (only == 3 ? 0 : only) : only, deep_only, work_weight, g.trans_a && g.trans_b);
    // tuning aid: UNITE_GEMM_FORCE_PLAN="kind,S" pins tile kernel (1: 128^2, 2: 256^2) and split factor for products that may split
    static const char* force_plan = getenv("UNITE_GEMM_FORCE_PLAN");
    if (force_plan && can_split && !want_colsum) {
        const int fk = atoi(force_plan), fs = strchr(force_plan, ',') ? atoi(strchr(force_plan, ',') + 1) : 1;
        const int kt = (g.K + BK - 1) / BK;
        if ((fk == 1 || fk == 2) && fs >= 1 && fs <= 16 && kt / fs >= 1 &&
            (size_t)fs * g.M * g.N * sizeof(float) <= (size_t)g.workspace_bytes - ws_head) plan = {fk, fs, 0.0};
    }
    static const bool plan_dbg = getenv("UNITE_GEMM_PLAN_DEBUG") != nullptr;
    if (plan_dbg) fprintf(stderr, "[unite_gemm] M %d N %d K %d ta %d tb %d w %.2f -> kind %d S %d\n", g.M, g.N, g.K, g.trans_a, g.trans_b, work_weight, plan.kind, plan.splitk);
    const int kind = (force && !strcmp(force, "simple") && !deep_only) ? 0 : plan.kind;      // 0 simple 128^2, 1 deep 128^2, 2 deep 256^2, 3 wide
    p.colsum_partial = want_colsum ? (float*)((char*)g.workspace + UNITE_WS_HEADER_BYTES) : nullptr;      // the header stays zero for split-K users of the same buffer
    const int tiles = kind == 2 ? ((g.M + 255) / 256) * ((g.N + 255) / 256)
                    : kind == 3 ? ((g.M + 127) / 128) * ((g.N + 255) / 256) : ((g.M + 127) / 128) * ((g.N + 127) / 128);
    p.splitk = 1;
    p.k_chunk = (g.K + BK - 1) / BK * BK;
    p.slab = nullptr;
    static const int dbg = getenv("UNITE_GEMM_DEBUG_SKIP") ? atoi(getenv("UNITE_GEMM_DEBUG_SKIP")) : 0;
    p.debug_skip = dbg;
    // Large bf16 outputs (e.g. 232-310 MB per teacher GEMM) are written through and dropped from the XCD L2 (sc1 stores):
    // left in the L2 they evict the B operand, which every XCD then re-fetches each round (+3 % measured).  f32 outputs
    // (32 B per lane) measured slower with sc1 and keep plain stores.  UNITE_GEMM_NT = 0 | 1 (nt) | 2 (sc1) overrides.
    static const int nt = getenv("UNITE_GEMM_NT") ? atoi(getenv("UNITE_GEMM_NT")) : -1;
    p.nt_store = nt >= 0 ? nt : ((!g.out_f32 && (size_t)g.M * g.N * 2 > (32u << 20)) ? 2 : 0);
    if (plan.splitk > 1 && kind != 0) {
        p.k_chunk = ((g.K + plan.splitk - 1) / plan.splitk + BK - 1) / BK * BK;
        p.splitk = (g.K + p.k_chunk - 1) / p.k_chunk;      // drop empty trailing slices
        p.counters = (uint32_t*)g.workspace;
        p.rowsum_slab = (float*)((char*)g.workspace + UNITE_WS_HEADER_BYTES);
        p.slab = (float*)((char*)g.workspace + ws_head);
        // who adds the slabs up (UNITE_SPLITK_SEPARATE): 1 (default) a second launch, splitk_finish_kernel; 0 the last slice of every tile inside the
        // launch; 2 the launch itself for 128^2 tiles and a second launch for 256^2 ones, whose last workgroup would read S x 256 KiB on its own.
        // Same sums in the same order either way.  In the step the three are within noise of each other (21.24 / 21.29 / 21.25 ms, same call); the
        // second launch is the default because the kernel form without the reduction code needs fewer registers (see gemm_deep_kernel)
        static const int sep_env = getenv("UNITE_SPLITK_SEPARATE") ? atoi(getenv("UNITE_SPLITK_SEPARATE")) : 1;
        p.separate_reduce = sep_env == 1 || (sep_env == 2 && kind == 2);
    }
    const int nb = tiles * p.splitk;
    // kernel form: 0 plain, 1 with the row sums compiled in, 2 with the in-launch reduction too (weight-gradient layout only; elsewhere a split
    // product falls back to the second launch)
    if (p.splitk > 1 && !p.separate_reduce && !(g.trans_a && g.trans_b)) p.separate_reduce = 1;
    const int wgf = (p.splitk > 1 && !p.separate_reduce) ? 2 : (want_rowsum ? 1 : 0);
    {
        // tile rows walked in groups (column-major inside a group): pays where ONE tile row's A panel is a large part of an XCD's 4-MiB L2, i.e.
        // for very deep products (8192^3: 1117 -> 1367 TF/s with groups of four); at the depths of the training step (K <= 3072) it changes
        // neither kernel nor step time, so it is off there.  UNITE_GEMM_GROUP_ROWS=r forces r (0: never).
        static const int gr_env = getenv("UNITE_GEMM_GROUP_ROWS") ? atoi(getenv("UNITE_GEMM_GROUP_ROWS")) : -1;
        const int tile_e = kind == 2 ? 256 : 128;
        const bool deep = g.K >= 4096 && (g.M + tile_e - 1) / tile_e >= 8 && (g.N + tile_e - 1) / tile_e >= 8;
        // round 4: the same walk in groups of eight tile rows for the teacher's SHALLOW, WIDE products (c_fc, qkv: K = 768, 9-12 tile columns, 197 tile
        // rows).  Timed in short bursts it changes nothing (round 3); in a sustained loop -- where the chip sits at its power limit and time measures
        // energy -- it is -2 .. -3 % (294 vs 301 us c_fc, 193 vs 199 us qkv: fewer weight-panel re-reads from beyond the L2), and +1 .. +8 % on the
        // three-column c_proj, which therefore keeps the plain order (profiles/r04_clock_notes.txt)
        const bool shallow_wide = kind == 2 && g.K <= 1024 && (g.M + tile_e - 1) / tile_e >= 64 && (g.N + tile_e - 1) / tile_e >= 8;
        p.group_rows = gr_env >= 0 ? gr_env : (deep ? 4 : shallow_wide ? 8 : 0);
    }
    hipStream_t s = (hipStream_t)stream;
    const bool prof = g_prof.on && g_prof.used < g_prof.ev.size();
    if (prof) (void)hipEventRecord(g_prof.ev[g_prof.used].first, s);
    if (kind == 3) {
        // register-direct epilogue measured slower than the LDS-staged one (narrower stores): opt-in for experiments only
        static const bool direct = getenv("UNITE_GEMM_WIDE_DIRECT") && atoi(getenv("UNITE_GEMM_WIDE_DIRECT"));
        if (direct) {
            if (!g.trans_a && !g.trans_b) hipLaunchKernelGGL((gemm_wide_kernel<false, false, true>), dim3(nb), dim3(256), WIDE_LDS, s, p);
            else if (!g.trans_a && g.trans_b) hipLaunchKernelGGL((gemm_wide_kernel<false, true, true>), dim3(nb), dim3(256), WIDE_LDS, s, p);
            else if (g.trans_a && !g.trans_b) hipLaunchKernelGGL((gemm_wide_kernel<true, false, true>), dim3(nb), dim3(256), WIDE_LDS, s, p);
            else hipLaunchKernelGGL((gemm_wide_kernel<true, true, true>), dim3(nb), dim3(256), WIDE_LDS, s, p);
        } else {
            if (!g.trans_a && !g.trans_b) hipLaunchKernelGGL((gemm_wide_kernel<false, false, false>), dim3(nb), dim3(256), WIDE_LDS, s, p);
            else if (!g.trans_a && g.trans_b) hipLaunchKernelGGL((gemm_wide_kernel<false, true, false>), dim3(nb), dim3(256), WIDE_LDS, s, p);
            else if (g.trans_a && !g.trans_b) hipLaunchKernelGGL((gemm_wide_kernel<true, false, false>), dim3(nb), dim3(256), WIDE_LDS, s, p);
            else hipLaunchKernelGGL((gemm_wide_kernel<true, true, false>), dim3(nb), dim3(256), WIDE_LDS, s, p);
        }
    } else if (kind == 2 || kind == 1) {
        // kernel form x operand layouts x main-loop schedule (0: fragment reads at the head of each phase, 1: software-pipelined, see gemm_deep_kernel)
        const int sched = (wgf == 2) ? 0 : (g.plan_flags & 4) ? ((g.plan_flags >> 3) & 1) : g_sched_env();
        // epilogue form 1 (transposed accumulators, bf16 image): 256^2 tiles, schedule 1, a bf16 output that needs nothing read or written beside it
        const bool epi1_ok = kind == 2 && sched == 1 && wgf == 0 && p.splitk == 1 && !g.out_f32 && !g.residual && !g.aux_out && !g.out_bf16_copy &&
                             !want_colsum && (g.act == UNITE_ACT_NONE || g.act == UNITE_ACT_GELU || g.act == UNITE_ACT_QUICKGELU) && p.ngroups == 1 &&
                             (!g.bias || (((uintptr_t)g.bias) & 15) == 0);
        // (an f32 form of the same idea -- 16-byte stores straight from the transposed accumulators, no LDS: 16 rows x 64 bytes per instruction -- was
        // built and measured in round 4: bit-identical in its tests and SLOWER, c_proj 308 vs 282 us sustained, out_proj 169 vs 141, the step +0.35 ms;
        // removed.  Half-line pieces cost more than the LDS round trip saves.)
        const int epi = epi1_ok ? ((g.plan_flags & 16) ? ((g.plan_flags >> 5) & 1) : g_epi_env()) : 0;
        const int rc = epi ? launch_deep_epi1(g.trans_a != 0, g.trans_b != 0, nb, s, p)
                     : kind == 2 ? launch_deep<128>(wgf, g.trans_a != 0, g.trans_b != 0, sched, nb, s, p)
                                 : launch_deep<64>(wgf, g.trans_a != 0, g.trans_b != 0, sched, nb, s, p);
        if (rc != UNITE_OK) return rc;
    } else if (!g.trans_a && !g.trans_b) hipLaunchKernelGGL((gemm_bf16_kernel<false, false>), dim3(nb), dim3(256), LDS_BYTES, s, p);
    else if (!g.trans_a && g.trans_b) hipLaunchKernelGGL((gemm_bf16_kernel<false, true>), dim3(nb), dim3(256), LDS_BYTES, s, p);
    else if (g.trans_a && !g.trans_b) hipLaunchKernelGGL((gemm_bf16_kernel<true, false>), dim3(nb), dim3(256), LDS_BYTES, s, p);
    else hipLaunchKernelGGL((gemm_bf16_kernel<true, true>), dim3(nb), dim3(256), LDS_BYTES, s, p);
    if (p.splitk > 1 && p.separate_reduce) {
        const size_t total4 = (size_t)g.M * g.N / 4;
        const unsigned grid = (unsigned)((total4 + 255) / 256 < 2048 ? (total4 + 255) / 256 : 2048);
        hipLaunchKernelGGL(splitk_finish_kernel, dim3(grid), dim3(256), 0, s, (const float*)p.slab, (const float*)p.rowsum_slab, p.splitk, g.M, g.N,
                           (float*)g.out, g.ldc, g.accumulate, g.rowsum_a_out, g.rowsum_accumulate, g.rowsum_zero_lo, g.rowsum_zero_hi);
    }
    if (want_colsum) {
        const int tile_m = kind == 2 ? 256 : 128;
        hipLaunchKernelGGL(colsum_rows_kernel, dim3((g.N + 63) / 64), dim3(256), 0, s, (const float*)p.colsum_partial, (g.M + tile_m - 1) / tile_m,
                           g.N, g.colsum_out, g.colsum_accumulate);
    }
    if (prof) {
        (void)hipEventRecord(g_prof.ev[g_prof.used].second, s);
        g_prof.used++;
        g_prof.flops += 2.0 * g.M * g.N * g.K;
        g_prof.bytes += algo_bytes(g);
    }
    UNITE_LAUNCH_CHECK();
    return UNITE_OK;
}

extern "C" size_t unite_gemm_colsum_workspace(int32_t M, int32_t N) {
    return UNITE_WS_HEADER_BYTES + (size_t)((M + 127) / 128) * (size_t)N * sizeof(float);
}

// the same timing pool for the other MFMA kernel of the step (teacher_fused.hip); not part of the C ABI
bool unite_prof_begin(hipStream_t s) {
    const bool prof = g_prof.on && g_prof.used < g_prof.ev.size();
    if (prof) (void)hipEventRecord(g_prof.ev[g_prof.used].first, s);
    return prof;
}
void unite_prof_end(hipStream_t s, double flops, double bytes) {
    (void)hipEventRecord(g_prof.ev[g_prof.used].second, s);
    g_prof.used++;
    g_prof.flops += flops;
    g_prof.bytes += bytes;
}

// ---- diagnostics (bench.py): HIP events around every GEMM launch, on the stream it is launched on
extern "C" int unite_prof_enable(int32_t on, int32_t max_launches) {
    if (on) {
        while ((int)g_prof.ev.size() < max_launches) {
            hipEvent_t a, b;
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return UNITE_EINVAL;
            g_prof.ev.emplace_back(a, b);
        }
        g_prof.used = 0;
        g_prof.flops = 0.0;
        g_prof.bytes = 0.0;
    }
    g_prof.on = on != 0;
    return UNITE_OK;
}

extern "C" int unite_prof_summary(double* total_ms, int64_t* launches, double* total_flops, double* total_bytes) {
    double ms = 0.0;
    for (size_t i = 0; i < g_prof.used; ++i) {
        float t = 0.f;
        if (hipEventSynchronize(g_prof.ev[i].second) != hipSuccess) return UNITE_EINVAL;
        if (hipEventElapsedTime(&t, g_prof.ev[i].first, g_prof.ev[i].second) != hipSuccess) return UNITE_EINVAL;
        ms += t;
    }
    if (total_ms) *total_ms = ms;
    if (launches) *launches = (int64_t)g_prof.used;
    if (total_flops) *total_flops = g_prof.flops;
    if (total_bytes) *total_bytes = g_prof.bytes;
    return UNITE_OK;
}
