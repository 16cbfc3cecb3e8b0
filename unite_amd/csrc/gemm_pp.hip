// Persistent bf16 MFMA GEMM with the epilogue of tile t hidden under the main loop of tile t+1 (gfx950 / MI355X).
//
//   out[M,N] = epilogue( A[M,K] * op(B)[K,N] )       A k-contiguous ([M][K]); B [N][K] (TB = false) or [K][N] (TB = true)
//
// Why: the 256 x 256 kernel of gemm.hip spends ~10 us per tile outside its main loop (pipeline fill, accumulators -> LDS -> stores,
// store drain before the workgroup can retire) next to 18 us of MFMA work at K = 768, and every launch is quantised to whole rounds
// of 256 workgroups.  Here ONE workgroup per CU walks over its tiles:
//   * tile 256 (M) x 128 (N) x 64 (K): 8 waves as 4 (M) x 2 (N), 64 x 64 outputs = 16 MFMA tiles = 64 accumulator registers per wave,
//     which leaves room for a SECOND accumulator set: while the MFMAs of tile t+1 fill `acc`, the finished tile t sits in `accp` and is
//     written out two MFMA tiles at a time during the first eight K-tiles of t+1 (bias / activation / residual math on the VALU beside the
//     other wave's MFMAs, loads and stores between the LDS-DMAs of the operand stream);
//   * the products are computed TRANSPOSED (B fragment in the MFMA's A slot), so a lane holds 4 consecutive columns of one output
//     row and the stores leave straight from the registers: no LDS round trip, no barrier, nothing that waits for the tile's end;
//     for bf16 outputs the two MFMA tiles of a pair trade halves between neighbouring lane rows (v_permlane16_swap) so that a lane's 8
//     values are 8 consecutive columns (16-byte stores, 64 contiguous bytes per row and instruction);
//   * the operand stream (LDS-DMA, three K-tiles of LDS, 3 K-tiles ahead) runs across tile boundaries: no pipeline fill per tile;
//   * fragments are double-buffered in registers (k-step 0 / 1), one s_barrier per K-tile.
// vmcnt bookkeeping: LDS-DMAs, epilogue loads (inline asm, invisible to hipcc's wait insertion) and stores retire in issue order, so every
// wait below is an exact count of the younger operations that may stay in flight (see ITER).
// Supported: K a multiple of 64 and >= 768 (twelve K-tiles: the drain schedule is static), no split-K / accumulate / colsum / bf16 copy --
// gemm.hip keeps those.
#include "common.h"
#include <stdlib.h>
#include <string.h>

namespace {

constexpr int PBM = 256, PBN = 128, PBK = 64;
constexpr int A_BYTES = PBM * PBK * 2;        // 32 KiB
constexpr int B_BYTES = PBN * PBK * 2;        // 16 KiB
constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
constexpr int NSTAGE = 3;
constexpr int PP_LDS = NSTAGE * STAGE_BYTES;  // 144 KiB
constexpr int NDRAIN = 8;                     // drain steps per tile: (m-tile i, n-tile pair p), two MFMA tiles each
constexpr int MIN_KT = NDRAIN + 4;          // K-tiles 0..8 drain, 9 settles the counts, the last two prefetch the next bias

struct PPParams {
    unite_gemm_args g;
    uint32_t a_bytes, b_bytes, out_bytes, res_bytes, bias_bytes, aux_in_bytes, aux_out_bytes, scale_bytes;
    int32_t nbm, nbn, ntiles, nk;
    uint32_t rps_magic;                       // ceil(2^32 / rows_per_scale): row / rows_per_scale = umulhi(row, magic), exact for row < 2^20, divisor in [2, 2^12]
    int32_t wt_store;                         // large bf16 outputs are written through (sc1) so that they do not evict the operand panels from the L2
    int32_t res16;                            // 1: bf16 output + bf16 residual, 2: the same in IEEE half (unite_gemm_args.residual_bf16): the residual rows travel in the aux_in slot
    int32_t ld_aux;                           // row stride of whatever the aux_in slot reads (saved pre-activations or the bf16 residual)
};

__device__ __forceinline__ int swz256p(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }
// swizzle key of B row n ([n][k] image, 128-B rows): natural fragments read rows 16 j + c, permuted ones 32 (j>>1) + 8 (c>>2) + 4 (j&1) + (c&3)
template <bool PERM>
__device__ __forceinline__ int bkey(int n) { return PERM ? ((n & 3) | (((n >> 3) & 1) << 2)) : (n & 7); }

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rs, char* lds, uint32_t voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (LDS_AS void*)lds, 16, voff, 0, 0, 0);
}

// register-destination loads the compiler does not count (it would drain the whole LDS-DMA queue at their first use)
__device__ __forceinline__ void asm_load_b128(f32x4& dst, uint32_t voff, __amdgpu_buffer_rsrc_t rs) {
    asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(dst) : "v"(voff), "s"(rs) : "memory");
}
__device__ __forceinline__ void asm_load_b32(float& dst, uint32_t voff, __amdgpu_buffer_rsrc_t rs) {
    asm volatile("s_nop 4\n\tbuffer_load_dword %0, %1, %2, 0 offen" : "=v"(dst) : "v"(voff), "s"(rs) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// ---- operand staging: one K-tile (A 256 x 64, B 128 x 64) = 48 LDS-DMA pieces of 1 KiB, six per wave.
// The tile / K-tile position goes into the DESCRIPTOR (base advanced, record count reduced: scalar arithmetic, once per K-tile), the lane's
// place inside the K-tile is a loop-invariant VGPR: a piece costs one v_add.  Rows past the matrix fall off the descriptor's end
// (K is a multiple of 64 here, and N a multiple of 128 for the k-strided B), so nothing is compared per lane.
struct LaneOffs { uint32_t a, b0, b1; };       // byte offsets of this lane's 16 bytes inside the K-tile's A block (piece 0) / B block (both pieces)

template <bool TB, bool PERM>
__device__ __forceinline__ LaneOffs lane_offsets(const unite_gemm_args& g, int wave, int lane) {
    LaneOffs o;
    {   // A image [256][64 k]: 128-B rows, 16-B chunk c at c ^ (row & 7); piece it = wave * 4 + x covers rows it * 8 .. + 7 (same row & 7 for every x)
        const int r = wave * 32 + (lane >> 3);
        const int lc = (lane & 7) ^ ((lane >> 3) & 7);
        o.a = (uint32_t)(r * g.lda + lc * 8) * 2u;
    }
    if (!TB) {      // B image [128 n][64 k]: chunk c at c ^ bkey(n); piece it = wave * 2 + x covers rows it * 8 .. + 7
        const int r = wave * 16 + (lane >> 3);
        o.b0 = (uint32_t)(r * g.ldb + ((lane & 7) ^ bkey<PERM>(r)) * 8) * 2u;
        o.b1 = (uint32_t)((r + 8) * g.ldb + ((lane & 7) ^ bkey<PERM>(r + 8)) * 8) * 2u;
    } else {        // B image [64 k][128 n]: 256-B rows, chunk c at c ^ swz256p(k); piece it covers k rows it * 4 .. + 3
        const int kr = wave * 8 + (lane >> 4);
        o.b0 = (uint32_t)(kr * g.ldb + ((lane & 15) ^ swz256p(kr)) * 8) * 2u;
        o.b1 = (uint32_t)((kr + 4) * g.ldb + ((lane & 15) ^ swz256p(kr + 4)) * 8) * 2u;
    }
    return o;
}

template <bool TB, bool PERM>
__device__ __forceinline__ void issue_ktile(const unite_gemm_args& g, __amdgpu_buffer_rsrc_t rsA, __amdgpu_buffer_rsrc_t rsB, char* stage,
                                            const LaneOffs& lo, int wave) {
#pragma unroll
    for (int x = 0; x < 4; ++x) dma16(rsA, stage + (wave * 4 + x) * 1024, lo.a + (uint32_t)(x * 8 * g.lda) * 2u);
    dma16(rsB, stage + A_BYTES + (wave * 2) * 1024, lo.b0);
    dma16(rsB, stage + A_BYTES + (wave * 2 + 1) * 1024, lo.b1);
}

__device__ __forceinline__ bf16x8 frag_a(const char* at, int row0, int ks, int lane) {
    const int row = row0 + (lane & 15), lc = ks * 4 + (lane >> 4);
    return *(const bf16x8*)(at + row * 128 + ((lc ^ (row & 7)) << 4));
}
// B fragment of the wave's n-tile j (wave column base nb): fragment row c <-> local column nloc(j, c)
template <bool TB, bool PERM>
__device__ __forceinline__ bf16x8 frag_b(const char* bt, int nb, int j, int ks, int lane) {
    if (!TB) {
        const int c = lane & 15;
        const int n = PERM ? nb + (j >> 1) * 32 + (c >> 2) * 8 + (j & 1) * 4 + (c & 3) : nb + j * 16 + c;
        const int lc = ks * 4 + (lane >> 4);
        return *(const bf16x8*)(bt + n * 128 + ((lc ^ bkey<PERM>(n)) << 4));
    } else {
        // transposing read: in each 16-lane group lane 4q + pp addresses k-row q, four consecutive columns; lane i gets column i
        asm volatile("" : "+v"(lane));      // opaque: the sixteen (j, k half, k-step) addresses are rebuilt with a few VALU ops each instead of living in registers across the loop (spills)
        const int G = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
        const int k_lo = ks * 32 + 8 * G + q, k_hi = k_lo + 4;
        int ch, sub;
        if (PERM) { ch = (nb >> 3) + (j >> 1) * 4 + pp; sub = 8 * (j & 1); }       // columns nb + 32 (j>>1) + 8 pp + 4 (j&1) ..+3
        else { ch = (nb >> 3) + j * 2 + (pp >> 1); sub = 8 * (pp & 1); }           // columns nb + 16 j + 4 pp ..+3
        return tr_join(lds_read_tr16_raw(bt + 256 * k_lo + ((ch ^ swz256p(k_lo)) << 4) + sub),
                       lds_read_tr16_raw(bt + 256 * k_hi + ((ch ^ swz256p(k_hi)) << 4) + sub));
    }
}

struct Frags { bf16x8 a[4], b[4]; };

template <bool TB, bool PERM>
__device__ __forceinline__ void load_frags(Frags& f, const char* stage, int arow, int nb, int ks, int lane) {
#pragma unroll
    for (int j = 0; j < 4; ++j) f.b[j] = frag_b<TB, PERM>(stage + A_BYTES, nb, j, ks, lane);
#pragma unroll
    for (int i = 0; i < 4; ++i) f.a[i] = frag_a(stage, arow + i * 16, ks, lane);
}

template <bool TB, bool PERM>
__device__ __forceinline__ void load_frags_b(Frags& f, const char* stage, int nb, int ks, int lane) {
#pragma unroll
    for (int j = 0; j < 4; ++j) f.b[j] = frag_b<TB, PERM>(stage + A_BYTES, nb, j, ks, lane);
}
__device__ __forceinline__ void load_frags_a(Frags& f, const char* stage, int arow, int ks, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) f.a[i] = frag_a(stage, arow + i * 16, ks, lane);
}
// the four MFMAs of m-tile i: D[n][m] (B fragment in the A slot): a lane holds 4 consecutive n of row m
__device__ __forceinline__ void mma_row(f32x4 (&acc)[4][4], const Frags& f, int i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(f.b[j], f.a[i], acc[i][j]);
}
#define PIN() __builtin_amdgcn_sched_barrier(0)

// ---- epilogue state of one drain step (loaded in phase B of K-tile e, consumed in phase A of K-tile e + 1)
// The bias is not here: it is the INITIAL VALUE of the accumulators (loaded one K-tile before the tile switch), so the product
// leaves the MFMA chain as acc = bias + A B and a drain step only needs the residual rows / saved pre-activations.
struct EpiRegs { f32x4 r0, r1; float sc; };

struct Descs { __amdgpu_buffer_rsrc_t out, res, bias, aux_in, aux_out, scale; };

template <bool F32OUT>
struct EpiCounts {
    static constexpr int L = F32OUT ? 3 : 1;      // loads per drain step: residual x 2 + row scale | aux_in
    static constexpr int S = 2;                   // stores per drain step: out x 2 (f32) | out + aux_out (bf16)
};
constexpr int NBIAS = 4;                          // bias loads per tile and lane (one 16-byte piece per n-tile)

// output coordinates of drain step E for this lane: row gm, first column gn (bf16: 8 consecutive columns; f32: 4 at gn and 4 at gn + 16)
template <bool F32OUT, int E>
__device__ __forceinline__ void epi_coords(int m0, int n0, int wm, int wn, int lane, int& gm, int& gn) {
    constexpr int i = E >> 1, pr = E & 1;
    asm volatile("" : "+v"(lane));      // opaque: keeps hipcc from hoisting sixteen per-lane row offsets out of the tile loop (they would be spilled)
    const int G = lane >> 4, c = lane & 15;
    gm = m0 + wm * 64 + i * 16 + c;
    // f32: n-tiles 2 pr and 2 pr + 1 in the MFMA's own layout (4 columns at 4 G of each).  bf16: after the lane-row exchange of epi_finish
    // lane row G holds columns 8 (G >> 1) .. + 7 of n-tile 2 pr + (G & 1)
    gn = F32OUT ? n0 + wn * 64 + pr * 32 + 4 * G : n0 + wn * 64 + (2 * pr + (G & 1)) * 16 + 8 * (G >> 1);
}

// f32 outputs leave (and residual rows arrive) in WHOLE 128-byte lines: the MFMA's own layout gives a lane 4 columns of n-tile 2 pr and 4 of
// n-tile 2 pr + 1 in ONE row, i.e. a store instruction would write 16 rows x 64 bytes (half lines: measured at half the HBM rate).  Lanes c
// and c ^ 8 of a lane row trade one of their two pieces (DPP row rotate by 8), after which lane (G, c) holds, for memory operation A: row
// (c & 7) of the m-tile, and for operation B: row (c & 7) + 8 -- both at n-tile 2 pr + (c >> 3): 8 lanes x 16 bytes = one full line per row.
__device__ __forceinline__ void f32_line_coords(int m0, int n0, int wm, int wn, int lane, int i, int pr, int& gmA, int& gn) {
    asm volatile("" : "+v"(lane));
    const int G = lane >> 4, c = lane & 15;
    gmA = m0 + wm * 64 + i * 16 + (c & 7);            // operation B: gmA + 8
    gn = n0 + wn * 64 + pr * 32 + (c >> 3) * 16 + 4 * G;
}
__device__ __forceinline__ uint32_t rot8(uint32_t x) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x128, 0xF, 0xF, true); }   // row_ror:8

// first column of the 4 consecutive columns lane (G, c) holds of its n-tile j
template <bool PERM>
__device__ __forceinline__ int ncol(int n0, int wn, int j, int lane) {
    const int G = lane >> 4;
    return PERM ? n0 + wn * 64 + (j >> 1) * 32 + G * 8 + (j & 1) * 4 : n0 + wn * 64 + j * 16 + 4 * G;
}

template <bool PERM>
__device__ __forceinline__ void bias_issue_loads(const unite_gemm_args& g, const Descs& d, f32x4 (&bv)[4], int n0, int wn, int lane, bool valid) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int gn = ncol<PERM>(n0, wn, j, lane);
        asm_load_b128(bv[j], (valid && gn < g.N) ? (uint32_t)gn * 4u : OOB_OFFSET, d.bias);       // zero-sized descriptor without a bias: zeros
    }
}

template <bool F32OUT, int E>
__device__ __forceinline__ void epi_issue_loads(const PPParams& p, const Descs& d, EpiRegs& e, int m0, int n0, int wm, int wn, int lane) {
    const unite_gemm_args& g = p.g;
    int gm, gn;
    epi_coords<F32OUT, E>(m0, n0, wm, wn, lane, gm, gn);
    const bool okm = gm < g.M;
    if (F32OUT) {
        int gmA, gl;
        f32_line_coords(m0, n0, wm, wn, lane, E >> 1, E & 1, gmA, gl);
        const bool okn = gl < g.N;
        asm_load_b128(e.r0, (okn && gmA < g.M) ? (uint32_t)(gmA * g.ldr + gl) * 4u : OOB_OFFSET, d.res);           // line layout, operation A
        asm_load_b128(e.r1, (okn && gmA + 8 < g.M) ? (uint32_t)((gmA + 8) * g.ldr + gl) * 4u : OOB_OFFSET, d.res);   // operation B
        asm_load_b32(e.sc, okm ? __umulhi((uint32_t)gm, p.rps_magic) * 4u : OOB_OFFSET, d.scale);
    } else {
        asm_load_b128(e.r0, (okm && gn < g.N) ? (uint32_t)(gm * p.ld_aux + gn) * 2u : OOB_OFFSET, d.aux_in);      // 8 bf16 pre-activations / residual values
    }
}

// results of one drain step, ready to store (f32: two 16-byte pieces; bf16: the output and the saved pre-activation)
struct EpiOut { u32x4 o0, o1; };

template <bool F32OUT, int E>
__device__ __forceinline__ EpiOut epi_math(const PPParams& p, EpiRegs& e, const f32x4 (&accp)[4][4]) {
    const unite_gemm_args& g = p.g;
    constexpr int i = E >> 1, pr = E & 1;
    float v[8];
#pragma unroll
    for (int r = 0; r < 4; ++r) { v[r] = accp[i][2 * pr][r]; v[4 + r] = accp[i][2 * pr + 1][r]; }
    EpiOut out;
    if (F32OUT) {
        if (g.row_scale) {
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] *= e.sc;
        }
        // to the line layout: a lane keeps the piece of its own half (c < 8: n-tile 2 pr, c >= 8: n-tile 2 pr + 1) and trades the other one
        // with lane c ^ 8; operation A then carries row c & 7, operation B row (c & 7) + 8
        const bool hi = (__lane_id() & 8) != 0;
        float a[4], b[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float got = __uint_as_float(rot8(__float_as_uint(hi ? v[r] : v[4 + r])));
            a[r] = hi ? got : v[r];
            b[r] = hi ? v[4 + r] : got;
        }
        if (g.residual) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { a[r] += e.r0[r]; b[r] += e.r1[r]; }
        }
        out.o0 = (u32x4){__float_as_uint(a[0]), __float_as_uint(a[1]), __float_as_uint(a[2]), __float_as_uint(a[3])};
        out.o1 = (u32x4){__float_as_uint(b[0]), __float_as_uint(b[1]), __float_as_uint(b[2]), __float_as_uint(b[3])};
    } else {
        // v_permlane16_swap: the odd lane rows of the first operand trade places with the even lane rows of the second.  Before: lane row
        // G holds columns 4 G .. + 3 of both tiles.  After: (v[r] | v[4 + r]) are 8 consecutive columns of ONE tile (epi_coords).
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[r]), __float_as_uint(v[4 + r]), false, false);
            v[r] = __uint_as_float(sw[0]);
            v[4 + r] = __uint_as_float(sw[1]);
        }
        out.o1 = (u32x4){0u, 0u, 0u, 0u};
        if (g.act == UNITE_ACT_GELU) {
            out.o1 = (u32x4){pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] = gelu_erf(v[r]);
        } else if (g.act == UNITE_ACT_QUICKGELU) {
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] = quick_gelu(v[r]);
        } else if (g.act == UNITE_ACT_DGELU) {
            const u32x4 z = __builtin_bit_cast(u32x4, e.r0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[2 * r] *= gelu_erf_grad(__uint_as_float(z[r] << 16));
                v[2 * r + 1] *= gelu_erf_grad(__uint_as_float(z[r] & 0xFFFF0000u));
            }
        } else if (p.res16 == 2) {            // the f16 stream: half rows in, half rows out
            const u32x4 z = __builtin_bit_cast(u32x4, e.r0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[2 * r] += unpack_f16_lo(z[r]);
                v[2 * r + 1] += unpack_f16_hi(z[r]);
            }
        } else if (p.res16) {
            const u32x4 z = __builtin_bit_cast(u32x4, e.r0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[2 * r] += __uint_as_float(z[r] << 16);
                v[2 * r + 1] += __uint_as_float(z[r] & 0xFFFF0000u);
            }
        }
        if (p.res16 == 2) out.o0 = (u32x4){pack_f16x2(v[0], v[1]), pack_f16x2(v[2], v[3]), pack_f16x2(v[4], v[5]), pack_f16x2(v[6], v[7])};
        else out.o0 = (u32x4){pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
    }
    return out;
}

// always TWO stores per step (the counted waits assume it): a store through a zero-sized descriptor is dropped
template <bool F32OUT, int E>
__device__ __forceinline__ void epi_store(const PPParams& p, const Descs& d, const EpiOut& out, int m0, int n0, int wm, int wn, int lane) {
    const unite_gemm_args& g = p.g;
    int gm, gn;
    epi_coords<F32OUT, E>(m0, n0, wm, wn, lane, gm, gn);
    const bool okm = gm < g.M;
    if (F32OUT) {
        int gmA, gl;
        f32_line_coords(m0, n0, wm, wn, lane, E >> 1, E & 1, gmA, gl);
        const uint32_t o0 = (gl < g.N && gmA < g.M) ? (uint32_t)(gmA * g.ldc + gl) * 4u : OOB_OFFSET;
        const uint32_t o1 = (gl < g.N && gmA + 8 < g.M) ? (uint32_t)((gmA + 8) * g.ldc + gl) * 4u : OOB_OFFSET;
        __builtin_amdgcn_raw_buffer_store_b128(out.o0, d.out, o0, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(out.o1, d.out, o1, 0, 0);
    } else {
        const bool ok = okm && gn < g.N;
        __builtin_amdgcn_raw_buffer_store_b128(out.o1, d.aux_out, ok ? (uint32_t)(gm * g.ld_aux_out + gn) * 2u : OOB_OFFSET, 0, 0);
        const uint32_t oo = ok ? (uint32_t)(gm * g.ldc + gn) * 2u : OOB_OFFSET;
        if (p.wt_store) __builtin_amdgcn_raw_buffer_store_b128(out.o0, d.out, oo, 0, 16);
        else __builtin_amdgcn_raw_buffer_store_b128(out.o0, d.out, oo, 0, 0);
    }
}

template <bool F32OUT, int N>
__device__ __forceinline__ void epi_wait_loads(EpiRegs& e) {       // at most N younger operations stay in flight; the registers are pinned across the wait
    if (F32OUT) asm volatile("s_waitcnt vmcnt(%3)" : "+v"(e.r0), "+v"(e.r1), "+v"(e.sc) : "n"(N) : "memory");
    else asm volatile("s_waitcnt vmcnt(%1)" : "+v"(e.r0) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void bias_wait_loads(f32x4 (&bv)[4]) {
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(bv[0]), "+v"(bv[1]), "+v"(bv[2]), "+v"(bv[3]) : "n"(N) : "memory");
}

// linear tile number -> tile coordinates: groups of four row panels, columns outside, rows inside, so that the 32 workgroups of an
// XCD work on ~8 column tiles x 4 row panels at a time (A panels and B columns both stay in the 4 MiB L2)
__device__ __forceinline__ void tile_coords(const PPParams& p, int e, int& m0, int& n0) {
    const int per_group = 4 * p.nbn;
    const int rg = e / per_group, rem = e - rg * per_group;
    const int rows_g = min(4, p.nbm - rg * 4);
    const int tn = rem / rows_g, tm = rg * 4 + (rem - tn * rows_g);
    m0 = tm * PBM;
    n0 = tn * PBN;
}

// descriptor of operand X restricted to what lies at or behind byte offset `off` (nothing if off is past the end)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rebase(const void* base, uint32_t bytes, uint32_t off, bool valid) {
    const uint32_t rem = (valid && off < bytes) ? bytes - off : 0u;
    return __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)base + (valid ? off : 0u)), 0, (int)rem, 0x00020000);
}

template <bool TB, bool F32OUT>
__global__ __launch_bounds__(512, 2) void gemm_pp_kernel(const PPParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr bool PERM = false;      // natural fragment rows; bf16 outputs are widened by a lane-row exchange in the epilogue instead
    constexpr int L = EpiCounts<F32OUT>::L, S = EpiCounts<F32OUT>::S;
    const unite_gemm_args& g = p.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int arow = wm * 64, nb = wn * 64;

    // this workgroup's tiles: label x = blockIdx & 7 (the blocks that share an XCD) owns a contiguous chunk of the tile order
    const int nwg = (int)gridDim.x, bid = (int)blockIdx.x, x = bid & 7, slot = bid >> 3;
    const int nslots = (nwg - x + 7) >> 3;
    const int qq = p.ntiles >> 3, rr = p.ntiles & 7;
    const int lo = x < rr ? x * (qq + 1) : rr * (qq + 1) + (x - rr) * qq, cnt = qq + (x < rr ? 1 : 0);
    if (slot >= cnt) return;
    const int ntl = (cnt - slot + nslots - 1) / nslots;
    const int nk = p.nk;

    Descs d;
    d.out = __builtin_amdgcn_make_buffer_rsrc((void*)g.out, 0, (int)p.out_bytes, 0x00020000);
    d.res = __builtin_amdgcn_make_buffer_rsrc((void*)g.residual, 0, (int)p.res_bytes, 0x00020000);
    d.bias = __builtin_amdgcn_make_buffer_rsrc((void*)g.bias, 0, (int)p.bias_bytes, 0x00020000);
    d.aux_in = __builtin_amdgcn_make_buffer_rsrc((void*)(p.res16 ? g.residual : g.aux_in), 0, (int)p.aux_in_bytes, 0x00020000);
    d.aux_out = __builtin_amdgcn_make_buffer_rsrc((void*)g.aux_out, 0, (int)p.aux_out_bytes, 0x00020000);
    d.scale = __builtin_amdgcn_make_buffer_rsrc((void*)g.row_scale, 0, (int)p.scale_bytes, 0x00020000);
    const LaneOffs lofs = lane_offsets<TB, PERM>(g, wave, lane);

    // operand stream: position q = (tile u = q / nk, K-tile q % nk) -> stage q % 3; issued three positions ahead of the MFMAs.
    // The descriptors of the NEXT position are kept as (pointer, bytes left) scalars and advanced by one K-tile per issue.
    int ld_u = 0, ld_kt = 0, ld_stage = 0, ld_m0, ld_n0;
    tile_coords(p, lo + slot, ld_m0, ld_n0);
    const uint32_t stepB = TB ? (uint32_t)(PBK * g.ldb) * 2u : (uint32_t)PBK * 2u;
    uint32_t offA, offB;          // byte offsets of the next position inside A / B (always inside the operand while ld_u < ntl)
    auto point_at_tile = [&]() {
        offA = (uint32_t)(ld_m0 * g.lda) * 2u;
        offB = TB ? (uint32_t)ld_n0 * 2u : (uint32_t)(ld_n0 * g.ldb) * 2u;
    };
    point_at_tile();
    // descriptor of what lies at / behind `off` (empty past the last tile): readfirstlane makes the words provably wave-uniform, or hipcc
    // wraps every LDS-DMA in a waterfall loop
    auto desc_at = [&](const void* base, uint32_t bytes, uint32_t off) {
        const bool valid = ld_u < ntl;
        const uint32_t o = __builtin_amdgcn_readfirstlane(valid ? off : 0u);
        const uint32_t rem = __builtin_amdgcn_readfirstlane(valid ? bytes - off : 0u);
        return __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)base + o), 0, (int)rem, 0x00020000);
    };
    auto issue_a = [&]() {        // the four A pieces of the next position
        const __amdgpu_buffer_rsrc_t rsA = desc_at(g.A, p.a_bytes, offA);
        char* stage = smem + ld_stage * STAGE_BYTES;
#pragma unroll
        for (int x2 = 0; x2 < 4; ++x2) dma16(rsA, stage + (wave * 4 + x2) * 1024, lofs.a + (uint32_t)(x2 * 8 * g.lda) * 2u);
    };
    auto issue_b = [&]() {        // the two B pieces, then advance to the following position
        const __amdgpu_buffer_rsrc_t rsB = desc_at(g.B, p.b_bytes, offB);
        char* stage = smem + ld_stage * STAGE_BYTES + A_BYTES;
        dma16(rsB, stage + (wave * 2) * 1024, lofs.b0);
        dma16(rsB, stage + (wave * 2 + 1) * 1024, lofs.b1);
        ld_stage = ld_stage == NSTAGE - 1 ? 0 : ld_stage + 1;
        if (++ld_kt == nk) {
            ld_kt = 0;
            ++ld_u;
            if (ld_u < ntl) tile_coords(p, lo + slot + ld_u * nslots, ld_m0, ld_n0);
            point_at_tile();
        } else {
            offA += PBK * 2;
            offB += stepB;
        }
    };
    auto issue_next = [&]() { issue_a(); issue_b(); };

    f32x4 acc[4][4], accp[4][4], bv[4];
    Frags f0, f1;
    EpiRegs er;
    er.r0 = er.r1 = (f32x4){0.f, 0.f, 0.f, 0.f};
    er.sc = 1.f;

    int cm0, cn0;                                                     // coordinates of the tile being multiplied
    tile_coords(p, lo + slot, cm0, cn0);
    bias_issue_loads<PERM>(g, d, bv, cn0, wn, lane, true);
    issue_next();
    issue_next();
    issue_next();
    bias_wait_loads<18>(bv);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = bv[j];
    wait_vm<12>();
    __builtin_amdgcn_s_barrier();
    int cs = 0;                                                       // stage of the K-tile being multiplied
    load_frags<TB, PERM>(f0, smem, arow, nb, 0, lane);
    lds_tr_fence<true>();

    int pm0 = 0, pn0 = 0;                                             // coordinates of the tile held in accp
    int nx_n0 = 0;
    bool nx_valid = false;
    // One K-tile.  WMID = operations younger than the NEXT K-tile's DMAs at the middle wait (the six DMAs of its successor + the epilogue /
    // bias loads and stores issued since).  LD / ST = drain step whose loads are issued in phase B / whose results are stored in phase A
    // (-1: none).  BIAS: phase B also loads the next tile's bias (consumed at the tile switch).
    auto iter = [&](auto wmid_c, auto ld_c, auto st_c, auto bias_c) {
        constexpr int WMID = decltype(wmid_c)::value, LD = decltype(ld_c)::value, ST = decltype(st_c)::value;
        constexpr bool BIAS = decltype(bias_c)::value != 0;
        const char* st0 = smem + cs * STAGE_BYTES;
        const int ns = cs == NSTAGE - 1 ? 0 : cs + 1;
        const char* st1 = smem + ns * STAGE_BYTES;
        // ---- phase A: k-step 0 of this K-tile from f0; fragments of k-step 1 into f1; finish drain step ST of the previous tile.
        // The order below is pinned (PIN = sched_barrier): four MFMAs, then the next piece of other work, so that the MFMA pipe is fed
        // from the first instruction after the barrier and the rest issues in the MFMAs' shadow (and in the other wave's).
        mma_row(acc, f0, 0);
        PIN();
        load_frags_b<TB, PERM>(f1, st0, nb, 1, lane);
        PIN();
        mma_row(acc, f0, 1);
        PIN();
        load_frags_a(f1, st0, arow, 1, lane);
        PIN();
        mma_row(acc, f0, 2);
        PIN();
        // drain: wait for the loads of step ST (issued one K-tile ago, just ahead of the stores of step ST - 1 and the six DMAs), do its
        // math, request the loads of step LD = ST + 1 and only then store: a wait never covers a store younger than two K-tiles
        EpiOut eo;
        if constexpr (ST >= 0) {
            epi_wait_loads<F32OUT, 6 + (ST >= 1 ? S : 0)>(er);
            eo = epi_math<F32OUT, ST>(p, er, accp);
        }
        if constexpr (LD >= 0) epi_issue_loads<F32OUT, LD>(p, d, er, pm0, pn0, wm, wn, lane);
        if constexpr (ST >= 0) epi_store<F32OUT, ST>(p, d, eo, pm0, pn0, wm, wn, lane);
        PIN();
        mma_row(acc, f0, 3);
        lds_tr_fence<true>();                                         // f1 complete; every read of this stage retired
        wait_vm<WMID>();                                              // the next K-tile has landed (this wave's pieces)
        __builtin_amdgcn_s_barrier();                                 // ... everybody's; this stage is free for position q + 3
        // ---- phase B: k-step 1 from f1; k-step 0 of the next K-tile into f0; epilogue / bias loads, then the next DMAs
        mma_row(acc, f1, 0);
        PIN();
        load_frags_b<TB, PERM>(f0, st1, nb, 0, lane);
        if constexpr (BIAS) bias_issue_loads<PERM>(g, d, bv, nx_n0, wn, lane, nx_valid);
        PIN();
        mma_row(acc, f1, 1);
        PIN();
        load_frags_a(f0, st1, arow, 0, lane);
        PIN();
        mma_row(acc, f1, 2);
        PIN();
        issue_a();
        PIN();
        mma_row(acc, f1, 3);
        PIN();
        issue_b();
        lds_tr_fence<true>();
        cs = ns;
    };
    using std::integral_constant;
    using IC = integral_constant<int, 0>;
    auto plain = [&]() { iter(integral_constant<int, 6>{}, integral_constant<int, -1>{}, integral_constant<int, -1>{}, IC{}); };

    // the last two K-tiles of a tile: the next tile's bias is requested (four loads ahead of the DMAs), the last one counts them; then the
    // tile is complete: it becomes the previous tile and the accumulators restart from the next tile's bias
    auto finish_tile = [&](int u) {
        iter(integral_constant<int, 6>{}, integral_constant<int, -1>{}, integral_constant<int, -1>{}, integral_constant<int, 1>{});
        iter(integral_constant<int, 6 + NBIAS>{}, integral_constant<int, -1>{}, integral_constant<int, -1>{}, IC{});
        bias_wait_loads<12>(bv);                                      // behind the bias loads: the DMAs of the last two K-tiles
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) { accp[i][j] = acc[i][j]; acc[i][j] = bv[j]; }
        pm0 = cm0;
        pn0 = cn0;
        if (nx_valid) tile_coords(p, lo + slot + (u + 1) * nslots, cm0, cn0);
    };
    auto look_ahead = [&](int u) {
        nx_valid = u + 1 < ntl;
        if (nx_valid) {
            int t0;
            tile_coords(p, lo + slot + (u + 1) * nslots, t0, nx_n0);
        }
    };

    // first tile: no previous tile to drain (peeled, so that accp is not live into it)
    look_ahead(0);
    for (int kt = 0; kt < nk - 2; ++kt) plain();
    finish_tile(0);
    for (int u = 1; u < ntl; ++u) {
        look_ahead(u);
        // phase A of K-tile kt: wait + math of step kt - 1, loads of step kt, stores of step kt - 1 (kt = 0 .. 8).  Middle waits = the six DMAs
        // of the successor + every epilogue operation issued since the awaited K-tile's DMAs (header): phase A(kt - 1) and phase A(kt)
        iter(integral_constant<int, 6 + L>{}, integral_constant<int, 0>{}, integral_constant<int, -1>{}, IC{});
        iter(integral_constant<int, 6 + 2 * L + S>{}, integral_constant<int, 1>{}, integral_constant<int, 0>{}, IC{});
        static_for<2, NDRAIN>([&](auto kc) {
            constexpr int k2 = decltype(kc)::value;
            iter(integral_constant<int, 6 + 2 * L + 2 * S>{}, integral_constant<int, k2>{}, integral_constant<int, k2 - 1>{}, IC{});
        });
        iter(integral_constant<int, 6 + L + 2 * S>{}, integral_constant<int, -1>{}, integral_constant<int, NDRAIN - 1>{}, IC{});
        iter(integral_constant<int, 6 + S>{}, integral_constant<int, -1>{}, integral_constant<int, -1>{}, IC{});
        for (int kt = NDRAIN + 2; kt < nk - 2; ++kt) plain();
        finish_tile(u);
    }
    // ---- last tile: nothing left to hide under
    wait_vm<0>();
    static_for<0, NDRAIN>([&](auto ec) {
        constexpr int E = decltype(ec)::value;
        epi_issue_loads<F32OUT, E>(p, d, er, pm0, pn0, wm, wn, lane);
        epi_wait_loads<F32OUT, 0>(er);
        const EpiOut eo = epi_math<F32OUT, E>(p, er, accp);
        epi_store<F32OUT, E>(p, d, eo, pm0, pn0, wm, wn, lane);
    });
}

inline bool aligned16p(const void* q) { return (((uintptr_t)q) & 15) == 0; }

}  // namespace

// 1 if unite_gemm_pp_launch can run the problem (the caller, unite_gemm_bf16, has already validated the arguments)
int unite_gemm_pp_supported(const unite_gemm_args& g) {
    if (g.trans_a) return 0;
    if (g.out_f32 && g.trans_b) return 0;
    if ((g.K % PBK) || g.K < MIN_KT * PBK) return 0;                 // whole K-tiles; the drain / bias schedule needs twelve of them
    if (g.trans_b && (g.N % PBN)) return 0;                          // k-strided B: a ragged last column tile would read the next k-row
    if (g.accumulate || g.colsum_out || g.out_bf16_copy) return 0;
    if ((g.lda & 7) || (g.ldb & 7) || (g.ldc & 7) || (g.N & 7)) return 0;
    if (g.lda < g.K || (g.trans_b ? g.ldb < g.N : g.ldb < g.K)) return 0;
    const int64_t mx = 0x7FFFFFF0;
    if (g.out_f32) {
        if (g.act != UNITE_ACT_NONE || g.aux_out || g.aux_in) return 0;
        if (g.residual && (g.residual_bf16 || (g.ldr & 3) || !aligned16p(g.residual))) return 0;
        if (((int64_t)(g.M - 1) * g.ldc + g.N) * 4 >= mx) return 0;
        if (g.residual && ((int64_t)(g.M - 1) * g.ldr + g.N) * 4 >= mx) return 0;
        if (g.row_scale && (g.rows_per_scale < 2 || g.rows_per_scale > 4096 || g.M >= (1 << 20))) return 0;
    } else {
        if (g.row_scale) return 0;
        if (g.act > UNITE_ACT_DGELU) return 0;                       // the saved-derivative forms are the tile kernels' (gemm.hip)
        if (g.residual && (!g.residual_bf16 || g.act != UNITE_ACT_NONE || g.aux_out || (g.ldr & 7) || !aligned16p(g.residual) ||
                           ((int64_t)(g.M - 1) * g.ldr + g.N) * 2 >= mx)) return 0;
        if (g.act == UNITE_ACT_DGELU && (!g.aux_in || !aligned16p(g.aux_in))) return 0;
        if (g.aux_out && (g.act != UNITE_ACT_GELU || !aligned16p(g.aux_out))) return 0;
        if (((int64_t)(g.M - 1) * g.ldc + g.N) * 2 >= mx) return 0;
    }
    if (g.bias && !aligned16p(g.bias)) return 0;
    return 1;
}

int unite_gemm_pp_launch(const unite_gemm_args& g, int64_t a_bytes, int64_t b_bytes, hipStream_t s) {
    PPParams p;
    memset(&p, 0, sizeof(p));
    p.g = g;
    p.a_bytes = (uint32_t)a_bytes;
    p.b_bytes = (uint32_t)b_bytes;
    const int esz = g.out_f32 ? 4 : 2;
    p.out_bytes = (uint32_t)(((int64_t)(g.M - 1) * g.ldc + g.N) * esz);
    p.res16 = (!g.out_f32 && g.residual && g.residual_bf16) ? g.residual_bf16 : 0;      // 1 bf16 rows, 2 f16 rows
    p.res_bytes = (g.residual && !p.res16) ? (uint32_t)(((int64_t)(g.M - 1) * g.ldr + g.N) * 4) : 0u;
    p.bias_bytes = g.bias ? (uint32_t)g.N * 4u : 0u;
    p.ld_aux = p.res16 ? g.ldr : g.ld_aux_in;
    p.aux_in_bytes = p.res16 ? (uint32_t)(((int64_t)(g.M - 1) * g.ldr + g.N) * 2)
                   : (g.aux_in && g.act == UNITE_ACT_DGELU) ? (uint32_t)(((int64_t)(g.M - 1) * g.ld_aux_in + g.N) * 2) : 0u;
    p.aux_out_bytes = (g.aux_out && g.act == UNITE_ACT_GELU) ? (uint32_t)(((int64_t)(g.M - 1) * g.ld_aux_out + g.N) * 2) : 0u;
    p.scale_bytes = g.row_scale ? (uint32_t)((g.M + g.rows_per_scale - 1) / g.rows_per_scale) * 4u : 0u;
    p.rps_magic = g.row_scale ? (uint32_t)((0x100000000ull + (uint64_t)g.rows_per_scale - 1) / (uint64_t)g.rows_per_scale) : 0u;
    static const int wt = getenv("UNITE_GEMM_NT") ? atoi(getenv("UNITE_GEMM_NT")) : -1;
    p.wt_store = wt >= 0 ? (wt == 2) : ((!g.out_f32 && (size_t)g.M * g.N * 2 > (32u << 20)) ? 1 : 0);
    static const int dbg = getenv("UNITE_PP_DEBUG") ? atoi(getenv("UNITE_PP_DEBUG")) : 0;      // timing experiments: 1 drops the stores, 2 the epilogue loads
    if (dbg & 1) p.out_bytes = p.aux_out_bytes = 0;
    if (dbg & 2) p.res_bytes = p.aux_in_bytes = p.scale_bytes = 0;
    p.nbm = (g.M + PBM - 1) / PBM;
    p.nbn = (g.N + PBN - 1) / PBN;
    p.ntiles = p.nbm * p.nbn;
    p.nk = g.K / PBK;
    static bool lds_ok = false;
    if (!lds_ok) {
        const void* ks[3] = {(const void*)gemm_pp_kernel<false, false>, (const void*)gemm_pp_kernel<true, false>,
                             (const void*)gemm_pp_kernel<false, true>};
        for (const void* k : ks) {
            hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS);
            if (e != hipSuccess) return (int)e;
        }
        lds_ok = true;
    }
    // UNITE_PP_WALK=k (experiment): workgroups of at most k tiles instead of one workgroup per CU for the whole product -- the finished tile
    // still drains under the next one inside a workgroup, but CUs change hands every k tiles (a launch beside another stream's kernels)
    static const int walk = getenv("UNITE_PP_WALK") ? atoi(getenv("UNITE_PP_WALK")) : 0;
    int grid = p.ntiles < 256 ? p.ntiles : 256;
    if (walk > 0) {
        int g2 = ((p.ntiles + walk - 1) / walk + 7) & ~7;
        if (g2 > p.ntiles) g2 = p.ntiles;
        if (g2 > grid) grid = g2;
    }
    if (g.out_f32) hipLaunchKernelGGL((gemm_pp_kernel<false, true>), dim3(grid), dim3(512), PP_LDS, s, p);
    else if (g.trans_b) hipLaunchKernelGGL((gemm_pp_kernel<true, false>), dim3(grid), dim3(512), PP_LDS, s, p);
    else hipLaunchKernelGGL((gemm_pp_kernel<false, false>), dim3(grid), dim3(512), PP_LDS, s, p);
    UNITE_LAUNCH_CHECK();
    return UNITE_OK;
}
