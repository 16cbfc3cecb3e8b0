// Fused multi-head self-attention (forward + backward) for short sequences on gfx950: head_dim 64, N <= 320 keys,
// i.e. the 320 visible student tokens and the 197 teacher tokens of UNITE stage 1.  A whole head's K and V
// (<= 40 KiB each) live in LDS, so the softmax is exact over the full row held in MFMA accumulators: no online
// rescaling, no N x N matrix in HBM.
//
// MFMA orientation (v_mfma_f32_16x16x32_bf16):  S^T = K Q^T puts the key index on the accumulator ROWS and the
// query on the LANES, so (a) row statistics are per-lane scalars + two cross-lane shuffles, and (b) the bf16-packed
// accumulators ARE the B operand of the next product (O^T = V^T P^T, dQ^T = K^T dS^T) with no LDS round trip:
// element j of lane group G is key 4G+j of the first 16-key tile (j<4) / of the second (j>=4), and the other operand
// is fetched with the same k order by two ds_read_b64_tr_b16 transposing reads.  The backward recomputes P from
// (Q, K, LSE): one kernel per query block for dQ (also emits delta = rowsum(dO*O)), one per key block for dK, dV
// (key on the lanes there, so P and dS are again operands as they stand).
//
// One LDS image serves row reads (ds_read_b128) and transposed reads: 128-B rows, 32-B chunk c of row r stored at
// chunk c ^ ((r>>1)&3); conflict-free for both access kinds.  Tiles are staged by LDS-DMA with the swizzle on the
// source address; rows >= N read as zero through the buffer descriptor's range check.
#include "attn_common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

// ------------------------------------------------------------------------------------ forward
// NW waves per workgroup share one head's K / V image (two workgroups fit a CU either way: the LDS image decides).  NW = 8 doubles the waves per
// SIMD (2 -> 4) at a 128-register budget, so the q fragments are fetched one tile ahead instead of all up front.  Round 4, alone on MI355X:
// N = 197 (the teacher's last block, 3072 heads) 105 -> 91 us; N = 320 (student, 384 heads) does not fit 128 registers with the whole score row
// held and a two-range form with a rescale measured 26.6 against 25.7 us -- it stays at NW = 4.  (Shader-clock stamps at N = 320: 12.1 k cycles of
// K / V staging at the CU's fill rate before the first product, then 4.7-5.9 k per q-tile: scores 1.6 k, soft-max 0.6 k, P V 2.7 k -- the
// transposing V reads --, store 1.0 k.)
template <int NT16, int NW>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(NW / 2))) void attn_fwd_kernel(const uint16_t* __restrict__ qkv, uint16_t* __restrict__ out, float* __restrict__ lse,
                                                          int N, int H, float scale, uint32_t qkv_bytes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NK = NT16 * 16;
    char* Ks = smem;
    char* Vs = smem + NK * 128;
    const int lane = threadIdx.x & 63, G = lane >> 4, c = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.x / H, h = blockIdx.x % H, HD = H * 64, ld = 3 * HD;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)qkv, 0, (int)qkv_bytes, 0x00020000);
    const uint32_t base_k = (uint32_t)(b * N) * ld + HD + h * 64;
    stage_rows(rs, Ks, NK, N, base_k, ld, wave, NW, lane);
    stage_rows(rs, Vs, NK, N, base_k + HD, ld, wave, NW, lane);
    // this wave's query fragments (q-tiles wave, wave + NW, ...) are fetched under the K / V staging (NW = 4: all of them; NW = 8: the first,
    // then one tile ahead): a global load at the top of every q-tile would expose a full HBM round trip (~2 us) per ~0.6 us of work
    constexpr int MAXQ = (NT16 + NW - 1) / NW;
    constexpr int HELD = NW == 4 ? MAXQ : 1;
    bf16x8 qfr[HELD][2];
    auto fetch_q = [&](int qt, bf16x8& f0, bf16x8& f1) {
        const int qc = min(qt * 16 + c, N - 1);
        const uint16_t* qp = qkv + (size_t)(b * N + qc) * ld + h * 64 + 8 * G;
        f0 = *(const bf16x8*)qp;
        f1 = *(const bf16x8*)(qp + 32);
    };
#pragma unroll
    for (int qi = 0; qi < HELD; ++qi) fetch_q(wave + NW * qi, qfr[qi][0], qfr[qi][1]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const float sl2 = scale * LOG2E;
    const int nqt = (N + 15) / 16;
    // NW = 4: the q-tile loop is unrolled (its fragments sit in registers by index); NW = 8: a real loop -- unrolled, the tail of one tile
    // overlaps the head of the next and the kernel no longer fits 128 registers
    constexpr int QUNROLL = NW == 4 ? MAXQ : 1;
#pragma unroll QUNROLL
    for (int qi = 0; qi < MAXQ; ++qi) {
        const int qt = wave + NW * qi;
        if (qt >= nqt) break;
        const int q = qt * 16 + c;
        const bf16x8 qf0 = qfr[NW == 4 ? qi : 0][0], qf1 = qfr[NW == 4 ? qi : 0][1];
        if (NW != 4 && qi + 1 < MAXQ) fetch_q(min(qt + NW, nqt - 1), qfr[0][0], qfr[0][1]);      // lands under this tile's work
        f32x4 st[NT16];
        float m = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NT16; ++kt) {
            f32x4 s = mfma16(row_frag(Ks, kt * 16, 0, lane), qf0, (f32x4){0.f, 0.f, 0.f, 0.f});
            s = mfma16(row_frag(Ks, kt * 16, 1, lane), qf1, s);
            if (kt >= NT16 - 2) {        // 16 (NT16 - 2) < N <= 16 NT16 (host): only the last two key tiles can hold keys beyond N
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (kt * 16 + 4 * G + r >= N) s[r] = -INFINITY;
            }
            m = max3(max3(m, s[0], s[1]), s[2], s[3]);
            st[kt] = s;
        }
        m = group_max(m);
        const float ml2 = m * sl2;
        float sum4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < NT16; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(st[kt][r], sl2, -ml2));   // one FMA + v_exp_f32; exp2(-inf) = 0
                st[kt][r] = p;
                sum4[r] += p;
            }
        float sum = group_sum((sum4[0] + sum4[1]) + (sum4[2] + sum4[3]));
        f32x4 o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < NT16 / 2; ++kk) {
            const bf16x8 pf = pack_pair(st[2 * kk], st[2 * kk + 1]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) o[dt] = mfma16(tr_frag(Vs, kk * 32, dt, lane), pf, o[dt]);   // O^T[d][q]
        }
        if (q < N) {
            const float inv = 1.0f / sum;
            uint16_t* op = out + (size_t)(b * N + q) * HD + h * 64 + 4 * G;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) store_bf16x4(op + dt * 16, o[dt], inv);
            if (G == 0) lse[((size_t)b * H + h) * N + q] = m * scale + __logf(sum);
        }
    }
}

// ------------------------------------------------------------------------------------ backward: dQ (+ delta)
// NW as in the forward kernel: 8 waves per workgroup (4 per SIMD; the kernel needs 112-126 registers) measured 37.5 against 41.5 us at N = 320
template <int NT16, int NW>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(NW / 2))) void attn_bwd_dq_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ out,
                                                             const uint16_t* __restrict__ dout, const float* __restrict__ lse,
                                                             float* __restrict__ delta, uint16_t* __restrict__ dqkv, int N, int H, float scale,
                                                             uint32_t qkv_bytes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NK = NT16 * 16;
    char* Ks = smem;
    char* Vs = smem + NK * 128;
    const int lane = threadIdx.x & 63, G = lane >> 4, c = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.x / H, h = blockIdx.x % H, HD = H * 64, ld = 3 * HD;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)qkv, 0, (int)qkv_bytes, 0x00020000);
    const uint32_t base_k = (uint32_t)(b * N) * ld + HD + h * 64;
    stage_rows(rs, Ks, NK, N, base_k, ld, wave, NW, lane);
    stage_rows(rs, Vs, NK, N, base_k + HD, ld, wave, NW, lane);

    const float sl2 = scale * LOG2E;
    const int nqt = (N + 15) / 16;
    // per-q-tile operands (Q, dO, O rows and the LSE) are fetched one q-tile ahead: the loads of tile i+1 fly under the ~1 us of
    // MFMA / softmax work of tile i instead of stalling every iteration for an HBM round trip
    bf16x8 nq0, nq1, nd0, nd1, no0, no1;
    float nl2;
    auto fetch = [&](int qt) {
        const int qc = min(qt * 16 + c, N - 1);
        const uint16_t* qp = qkv + (size_t)(b * N + qc) * ld + h * 64 + 8 * G;
        const uint16_t* dop = dout + (size_t)(b * N + qc) * HD + h * 64 + 8 * G;
        const uint16_t* oop = out + (size_t)(b * N + qc) * HD + h * 64 + 8 * G;
        nq0 = *(const bf16x8*)qp; nq1 = *(const bf16x8*)(qp + 32);
        nd0 = *(const bf16x8*)dop; nd1 = *(const bf16x8*)(dop + 32);
        no0 = *(const bf16x8*)oop; no1 = *(const bf16x8*)(oop + 32);
        nl2 = lse[((size_t)b * H + h) * N + qc] * LOG2E;
    };
    fetch(wave);             // under the K / V staging (waited for below together with it)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int qt = wave; qt < nqt; qt += NW) {
        const int q = qt * 16 + c;
        const bf16x8 qf0 = nq0, qf1 = nq1, df0 = nd0, df1 = nd1, of0 = no0, of1 = no1;
        const float l2 = nl2;
        if (qt + NW < nqt) fetch(qt + NW);
        float dl = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) dl += (float)df0[j] * (float)of0[j] + (float)df1[j] * (float)of1[j];
        dl = group_sum(dl);
        if (q < N && G == 0) delta[((size_t)b * H + h) * N + q] = dl;
        f32x4 dq[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dq[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // 32 keys per step.  NOT fully unrolled: with all NT16 / 2 steps in one block the scheduler hoists every K / V fragment read to
        // the top and the kernel spills (168 VGPRs at NT16 = 20: 100 us instead of 45); keys beyond N only exist in the last step.
        auto kstep = [&](int kk, auto masked_c) {
            constexpr bool MASKED = decltype(masked_c)::value;
            f32x4 ds[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int kt = 2 * kk + t;
                f32x4 s = mfma16(row_frag(Ks, kt * 16, 0, lane), qf0, (f32x4){0.f, 0.f, 0.f, 0.f});
                s = mfma16(row_frag(Ks, kt * 16, 1, lane), qf1, s);
                f32x4 dp = mfma16(row_frag(Vs, kt * 16, 0, lane), df0, (f32x4){0.f, 0.f, 0.f, 0.f});
                dp = mfma16(row_frag(Vs, kt * 16, 1, lane), df1, dp);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], sl2, -l2));
                    if (MASKED && kt * 16 + 4 * G + r >= N) p = 0.f;
                    ds[t][r] = p * (dp[r] - dl);
                }
            }
            const bf16x8 dsf = pack_pair(ds[0], ds[1]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) dq[dt] = mfma16(tr_frag(Ks, kk * 32, dt, lane), dsf, dq[dt]);   // dQ^T[d][q]
        };
        int nsteps = NT16 / 2 - 1;
        asm volatile("" : "+s"(nsteps));            // opaque trip count: a constant odd one is unrolled completely whatever the pragma says
        for (int kk = 0; kk < nsteps; ++kk) kstep(kk, std::false_type{});
        kstep(NT16 / 2 - 1, std::true_type{});
        if (q < N) {
            uint16_t* op = dqkv + (size_t)(b * N + q) * ld + h * 64 + 4 * G;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) store_bf16x4(op + dt * 16, dq[dt], scale);
        }
    }
}

// ------------------------------------------------------------------------------------ backward: dK, dV
// 8 waves; wave w owns key tiles [w*KTW, (w+1)*KTW); Q and dO of the head live in LDS; loop over 32-query steps.
template <int NT16, int KTW>
__global__ __launch_bounds__(512, 2) void attn_bwd_dkv_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ dout,
                                                              const float* __restrict__ lse, const float* __restrict__ delta,
                                                              uint16_t* __restrict__ dqkv, int N, int H, float scale, uint32_t qkv_bytes,
                                                              uint32_t dout_bytes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NQ = NT16 * 16;
    char* Qs = smem;
    char* Ds = smem + NQ * 128;
    float* lse_s = (float*)(smem + 2 * NQ * 128);
    float* dl_s = lse_s + NQ;
    const int lane = threadIdx.x & 63, G = lane >> 4, c = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.x / H, h = blockIdx.x % H, HD = H * 64, ld = 3 * HD;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)qkv, 0, (int)qkv_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsd = __builtin_amdgcn_make_buffer_rsrc((void*)dout, 0, (int)dout_bytes, 0x00020000);
    stage_rows(rs, Qs, NQ, N, (uint32_t)(b * N) * ld + h * 64, ld, wave, 8, lane);
    stage_rows(rsd, Ds, NQ, N, (uint32_t)(b * N) * HD + h * 64, HD, wave, 8, lane);
    for (int i = threadIdx.x; i < NQ; i += 512) {
        lse_s[i] = (i < N) ? lse[((size_t)b * H + h) * N + i] * LOG2E : 0.f;
        dl_s[i] = (i < N) ? delta[((size_t)b * H + h) * N + i] : 0.f;
    }
    // this wave's K / V fragments (B-operand layout: key on the lane)
    bf16x8 kf[KTW][2], vf[KTW][2];
#pragma unroll
    for (int i = 0; i < KTW; ++i) {
        const int key = (wave * KTW + i) * 16 + c, kc = min(key, N - 1);
        const uint16_t* kp = qkv + (size_t)(b * N + kc) * ld + HD + h * 64 + 8 * G;
        kf[i][0] = *(const bf16x8*)kp;
        kf[i][1] = *(const bf16x8*)(kp + 32);
        vf[i][0] = *(const bf16x8*)(kp + HD);
        vf[i][1] = *(const bf16x8*)(kp + HD + 32);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    f32x4 dk[KTW][4], dv[KTW][4];
#pragma unroll
    for (int i = 0; i < KTW; ++i)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dk[i][dt] = dv[i][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float sl2 = scale * LOG2E;

    auto qstep = [&](int kq, auto masked_c) {
        constexpr bool MASKED = decltype(masked_c)::value;      // query rows beyond N exist only in the last 32-query step
        f32x4 P[2][KTW], dS[2][KTW];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int q0 = kq * 32 + t * 16;
            const bf16x8 qa0 = row_frag(Qs, q0, 0, lane), qa1 = row_frag(Qs, q0, 1, lane);
            const bf16x8 da0 = row_frag(Ds, q0, 0, lane), da1 = row_frag(Ds, q0, 1, lane);
            const f32x4 l4 = *(const f32x4*)(lse_s + q0 + 4 * G), d4 = *(const f32x4*)(dl_s + q0 + 4 * G);
#pragma unroll
            for (int i = 0; i < KTW; ++i) {
                f32x4 s = mfma16(qa0, kf[i][0], (f32x4){0.f, 0.f, 0.f, 0.f});     // S[q][key]
                s = mfma16(qa1, kf[i][1], s);
                f32x4 dp = mfma16(da0, vf[i][0], (f32x4){0.f, 0.f, 0.f, 0.f});    // dP[q][key]
                dp = mfma16(da1, vf[i][1], dp);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], sl2, -l4[r]));
                    if (MASKED && q0 + 4 * G + r >= N) p = 0.f;
                    P[t][i][r] = p;
                    dS[t][i][r] = p * (dp[r] - d4[r]);
                }
            }
        }
        bf16x8 dot[4], qtr[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            dot[dt] = tr_frag(Ds, kq * 32, dt, lane);    // dO^T[d][q]
            qtr[dt] = tr_frag(Qs, kq * 32, dt, lane);    // Q^T[d][q]
        }
#pragma unroll
        for (int i = 0; i < KTW; ++i) {
            const bf16x8 pf = pack_pair(P[0][i], P[1][i]), dsf = pack_pair(dS[0][i], dS[1][i]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                dv[i][dt] = mfma16(dot[dt], pf, dv[i][dt]);     // dV^T[d][key]
                dk[i][dt] = mfma16(qtr[dt], dsf, dk[i][dt]);    // dK^T[d][key]
            }
        }
    };
    for (int kq = 0; kq < NQ / 32 - 1; ++kq) qstep(kq, std::false_type{});
    qstep(NQ / 32 - 1, std::true_type{});
#pragma unroll
    for (int i = 0; i < KTW; ++i) {
        const int key = (wave * KTW + i) * 16 + c;
        if (key < N) {
            uint16_t* kp = dqkv + (size_t)(b * N + key) * ld + HD + h * 64 + 4 * G;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                store_bf16x4(kp + dt * 16, dk[i][dt], scale);
                store_bf16x4(kp + HD + dt * 16, dv[i][dt], 1.0f);
            }
        }
    }
}

// key tiles of 16, even count, tight: 16 (NT16 - 2) < N <= 16 NT16, so that only the last two tiles need boundary masks
inline int nt16_for(int N) { return N <= 320 ? 2 * ((N + 31) / 32) : 0; }

// dynamic LDS above 64 KiB has to be opted into once per kernel
template <auto Kern>
inline int set_lds(int bytes) {
    static int granted = 0;
    if (bytes > 64 * 1024 && bytes > granted) {
        hipError_t e = hipFuncSetAttribute((const void*)Kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) return (int)e;
        granted = bytes;
    }
    return 0;
}

}  // namespace

#define ATTN_CASE(V, W, ...) case V: { constexpr int NT16 = V; constexpr int KTW = W; (void)KTW; __VA_ARGS__; } break;
#define ATTN_DISPATCH(NT, ...)                                                                                                   \
    switch (NT) {                                                                                                                \
        ATTN_CASE(2, 1, __VA_ARGS__) ATTN_CASE(4, 1, __VA_ARGS__) ATTN_CASE(6, 1, __VA_ARGS__) ATTN_CASE(8, 1, __VA_ARGS__)      \
        ATTN_CASE(10, 2, __VA_ARGS__) ATTN_CASE(12, 2, __VA_ARGS__) ATTN_CASE(14, 2, __VA_ARGS__) ATTN_CASE(16, 2, __VA_ARGS__) \
        ATTN_CASE(18, 3, __VA_ARGS__)                                                                                            \
        default: { constexpr int NT16 = 20; constexpr int KTW = 3; (void)KTW; __VA_ARGS__; } break;                              \
    }

int attn_fwd_tiled_launch(const void* qkv, void* out, float* lse, int B, int N, int H, float scale, hipStream_t stream);
int attn_bwd_tiled_launch(const void* qkv, const void* out, const void* dout, const float* lse, float* delta, void* dqkv, int B, int N, int H,
                          float scale, hipStream_t stream);

extern "C" int unite_attn_fwd(const void* qkv, void* out, float* lse, int32_t B, int32_t N, int32_t H, float scale, void* stream) {
    if (!qkv || !out || !lse || B <= 0 || N <= 0 || H <= 0) return UNITE_EINVAL;
    static const bool force_tiled = getenv("UNITE_ATTN_TILED") && atoi(getenv("UNITE_ATTN_TILED"));
    const int nt = force_tiled ? 0 : nt16_for(N);
    if (!nt) return attn_fwd_tiled_launch(qkv, out, lse, B, N, H, scale, (hipStream_t)stream);
    const int64_t bytes = (int64_t)B * N * 3 * H * 64 * 2;
    if (bytes >= (int64_t)OOB_OFFSET) return UNITE_ENOSUP;
    const int lds = nt * 16 * 128 * 2;
    static const int nw_env = getenv("UNITE_ATTN_WAVES") ? atoi(getenv("UNITE_ATTN_WAVES")) : 0;      // 4 / 8: pin the waves per workgroup (A/B)
    const bool eight = nt <= 16 && (nw_env ? nw_env == 8 : nt >= 14);      // above 256 keys the score row of a q-tile does not fit 128 registers
    ATTN_DISPATCH(nt, {
        if constexpr (NT16 <= 16) {
            if (eight) {
                int e = set_lds<attn_fwd_kernel<NT16, 8>>(lds);
                if (e) return e;
                hipLaunchKernelGGL((attn_fwd_kernel<NT16, 8>), dim3(B * H), dim3(512), lds, (hipStream_t)stream, (const uint16_t*)qkv,
                                   (uint16_t*)out, lse, N, H, scale, (uint32_t)bytes);
                break;
            }
        }
        int e = set_lds<attn_fwd_kernel<NT16, 4>>(lds);
        if (e) return e;
        hipLaunchKernelGGL((attn_fwd_kernel<NT16, 4>), dim3(B * H), dim3(256), lds, (hipStream_t)stream, (const uint16_t*)qkv, (uint16_t*)out, lse,
                           N, H, scale, (uint32_t)bytes);
    });
    UNITE_LAUNCH_CHECK();
    return UNITE_OK;
}

extern "C" int unite_attn_bwd(const void* qkv, const void* out, const void* dout, const float* lse, float* delta, void* dqkv, int32_t B,
                              int32_t N, int32_t H, float scale, void* stream) {
    if (!qkv || !out || !dout || !lse || !delta || !dqkv || B <= 0 || N <= 0 || H <= 0) return UNITE_EINVAL;
    static const bool force_tiled = getenv("UNITE_ATTN_TILED") && atoi(getenv("UNITE_ATTN_TILED"));
    const int nt = force_tiled ? 0 : nt16_for(N);
    if (!nt) return attn_bwd_tiled_launch(qkv, out, dout, lse, delta, dqkv, B, N, H, scale, (hipStream_t)stream);
    const int64_t bytes = (int64_t)B * N * 3 * H * 64 * 2;
    if (bytes >= (int64_t)OOB_OFFSET) return UNITE_ENOSUP;
    const int lds = nt * 16 * 128 * 2;
    const int lds2 = lds + nt * 16 * 4 * 2;
    static const int nw_env = getenv("UNITE_ATTN_WAVES") ? atoi(getenv("UNITE_ATTN_WAVES")) : 0;
    const bool eight = nw_env ? nw_env == 8 : nt >= 14;
    ATTN_DISPATCH(nt, {
        int e;
        if (eight) {
            e = set_lds<attn_bwd_dq_kernel<NT16, 8>>(lds);
            if (e) return e;
            hipLaunchKernelGGL((attn_bwd_dq_kernel<NT16, 8>), dim3(B * H), dim3(512), lds, (hipStream_t)stream, (const uint16_t*)qkv,
                               (const uint16_t*)out, (const uint16_t*)dout, lse, delta, (uint16_t*)dqkv, N, H, scale, (uint32_t)bytes);
        } else {
            e = set_lds<attn_bwd_dq_kernel<NT16, 4>>(lds);
            if (e) return e;
            hipLaunchKernelGGL((attn_bwd_dq_kernel<NT16, 4>), dim3(B * H), dim3(256), lds, (hipStream_t)stream, (const uint16_t*)qkv,
                               (const uint16_t*)out, (const uint16_t*)dout, lse, delta, (uint16_t*)dqkv, N, H, scale, (uint32_t)bytes);
        }
        UNITE_LAUNCH_CHECK();
        e = set_lds<attn_bwd_dkv_kernel<NT16, KTW>>(lds2);
        if (e) return e;
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<NT16, KTW>), dim3(B * H), dim3(512), lds2, (hipStream_t)stream, (const uint16_t*)qkv,
                           (const uint16_t*)dout, lse, delta, (uint16_t*)dqkv, N, H, scale, (uint32_t)bytes, (uint32_t)(bytes / 3));
    });
    UNITE_LAUNCH_CHECK();
    return UNITE_OK;
}
