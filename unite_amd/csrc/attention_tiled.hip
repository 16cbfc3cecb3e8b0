// Flash-style (tiled, online-softmax) attention forward / backward for sequences that do not fit a whole head in LDS:
// the all-token passes of UNITE stage 2 (N = 1568 / 3136 tokens, reference modeling_finetune.py:111-116 materialises a
// (B,H,N,N) probability tensor there) and of stage 3 (run_stage3.py:475-483).  Same MFMA orientation and LDS image as the
// short-sequence kernels (attention.hip): S^T = K Q^T with the key on the accumulator rows, so the bf16-packed
// accumulators are the next product's B operand and V^T / K^T come from transposing LDS reads.
//   forward : workgroup = (batch, head, 128 queries); 4 waves x 2 query tiles of 16; K/V tiles of 64 keys double-buffered
//             in LDS by LDS-DMA; running max / sum per query, O rescaled when the max moves.
//   backward: dQ kernel (same tiling, P recomputed from LSE, also writes delta = rowsum(dO*O));
//             dK/dV kernel: workgroup = (batch, head, 256 keys), 8 waves x 2 key tiles with K,V fragments in registers,
//             Q/dO tiles of 32 queries double-buffered in LDS.
#include "attn_common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int TK = 64;                 // keys per LDS tile (forward, dQ)
constexpr int TILE_B = TK * 128;       // 8 KiB per K or V tile
constexpr int NBUF = 4;                // LDS ring: tile kt + 3 is in flight while tile kt is consumed (an HBM round trip is ~3 tiles long)

// tile kt has landed for this wave once at most 4 * min(2, tiles issued after it) of its LDS-DMA instructions are outstanding
// (each stage() is 4 per wave in forward / dQ: 2 K pieces + 2 V pieces)
template <int NB, int PER = 4>
__device__ __forceinline__ void wait_tile(int rem) {          // NB - 2 younger tiles (PER LDS-DMA instructions each) may stay in flight
    if (NB >= 4 && rem >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER) : "memory");
    else if (NB >= 3 && rem >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ------------------------------------------------------------------------------------ forward
template <int NB, int SUB>
__global__ __launch_bounds__(256, 2) void attn_fwd_tiled_kernel(const uint16_t* __restrict__ qkv, uint16_t* __restrict__ out,
                                                                float* __restrict__ lse, int N, int H, float scale, uint32_t qkv_bytes,
                                                                int nqb) {
    constexpr int TKS = TK * SUB, TB = TKS * 128;      // keys per staged tile (consumed as SUB sub-tiles of 64), bytes per K or V tile
    __shared__ __attribute__((aligned(16))) char smem[NB * 2 * TB];      // ring of [K | V] tiles
    const int lane = threadIdx.x & 63, G = lane >> 4, c = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int bh = blockIdx.x / nqb, qb = blockIdx.x % nqb;
    const int b = bh / H, h = bh % H, HD = H * 64, ld = 3 * HD;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)qkv, 0, (int)qkv_bytes, 0x00020000);
    const uint32_t base_k = (uint32_t)(b * N) * ld + HD + h * 64;
    const int nkt = (N + TKS - 1) / TKS;
    auto stage = [&](int kt) {
        char* buf = smem + (kt % NB) * 2 * TB;
        stage_rows(rs, buf, TKS, N - kt * TKS, base_k + (uint32_t)(kt * TKS) * ld, ld, wave, 4, lane);
        stage_rows(rs, buf + TB, TKS, N - kt * TKS, base_k + HD + (uint32_t)(kt * TKS) * ld, ld, wave, 4, lane);
    };
#pragma unroll
    for (int i = 0; i < NB - 1; ++i)
        if (i < nkt) stage(i);

    const int q0 = qb * 128 + wave * 32;
    bf16x8 qf[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int qc = min(q0 + t * 16 + c, N - 1);
        const uint16_t* qp = qkv + (size_t)(b * N + qc) * ld + h * 64 + 8 * G;
        qf[t][0] = *(const bf16x8*)qp;
        qf[t][1] = *(const bf16x8*)(qp + 32);
    }
    f32x4 o[2][4];
    float m[2] = {-INFINITY, -INFINITY}, l[2] = {0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[t][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float sl2 = scale * LOG2E;

    // one key tile; MASKED (keys beyond N) only for the last one.  The V^T fragments are read (asm, untracked) right behind the K
    // fragments so that their LDS latency hides under QK^T and the softmax; the next tile's LDS-DMA stays in flight throughout.
    auto ktile = [&](int kt, auto masked_c) {
        constexpr bool MASKED = decltype(masked_c)::value;
        wait_tile<NB, 4 * SUB>(nkt - 1 - kt);
        __syncthreads();                       // tile kt landed for every wave; everyone is done with tile kt - 1's buffer
        if (kt + NB - 1 < nkt) stage(kt + NB - 1);
#pragma unroll
        for (int sub = 0; sub < SUB; ++sub) {
        const char* Ks = smem + (kt % NB) * 2 * TB + sub * TILE_B;
        const char* Vs = Ks + TB;
        const int key0 = kt * TKS + sub * TK;
        bf16x8 kf[4][2], vf[2][4];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            kf[s4][0] = row_frag(Ks, s4 * 16, 0, lane);
            kf[s4][1] = row_frag(Ks, s4 * 16, 1, lane);
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) vf[kk][dt] = tr_frag_raw(Vs, kk * 32, dt, lane);
        bf16x8 pf[2][2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x4 s[4];
            float mt = -INFINITY;
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                f32x4 v = mfma16(kf[s4][0], qf[t][0], (f32x4){0.f, 0.f, 0.f, 0.f});
                v = mfma16(kf[s4][1], qf[t][1], v);
                if (MASKED) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (key0 + s4 * 16 + 4 * G + r >= N) v[r] = -INFINITY;
                }
                mt = max3(max3(mt, v[0], v[1]), v[2], v[3]);
                s[s4] = v;
            }
            mt = group_max(mt);
            const float mn = fmaxf(m[t], mt), mn2 = mn * sl2;
            const float alpha = __builtin_amdgcn_exp2f(__builtin_fmaf(m[t], sl2, -mn2));      // 0 on the first tile (m = -inf)
            float ls4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[s4][r], sl2, -mn2));
                    s[s4][r] = p;
                    ls4[r] += p;
                }
            l[t] = l[t] * alpha + ((ls4[0] + ls4[1]) + (ls4[2] + ls4[3]));      // per-lane partial over this lane's keys; summed across G at the end
            m[t] = mn;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) o[t][dt] = (f32x4){o[t][dt][0] * alpha, o[t][dt][1] * alpha, o[t][dt][2] * alpha, o[t][dt][3] * alpha};
            pf[t][0] = pack_pair(s[0], s[1]);
            pf[t][1] = pack_pair(s[2], s[3]);
        }
        lds_tr_fence<true>();
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                o[0][dt] = mfma16(vf[kk][dt], pf[0][kk], o[0][dt]);
                o[1][dt] = mfma16(vf[kk][dt], pf[1][kk], o[1][dt]);
            }
        }
    };
    for (int kt = 0; kt < nkt - 1; ++kt) ktile(kt, std::false_type{});
    ktile(nkt - 1, std::true_type{});
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int q = q0 + t * 16 + c;
        const float lt = group_sum(l[t]);
        if (q < N) {
            const float inv = 1.0f / lt;
            uint16_t* op = out + (size_t)(b * N + q) * HD + h * 64 + 4 * G;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) store_bf16x4(op + dt * 16, o[t][dt], inv);
            if (G == 0) lse[((size_t)b * H + h) * N + q] = m[t] * scale + __logf(lt);
        }
    }
}

// ------------------------------------------------------------------------------------ backward: dQ (+ delta)
template <int NB>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_tiled_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ out,
                                                                   const uint16_t* __restrict__ dout, const float* __restrict__ lse,
                                                                   float* __restrict__ delta, uint16_t* __restrict__ dqkv, int N, int H,
                                                                   float scale, uint32_t qkv_bytes, int nqb) {
    __shared__ __attribute__((aligned(16))) char smem[NB * 2 * TILE_B];
    const int lane = threadIdx.x & 63, G = lane >> 4, c = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int bh = blockIdx.x / nqb, qb = blockIdx.x % nqb;
    const int b = bh / H, h = bh % H, HD = H * 64, ld = 3 * HD;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)qkv, 0, (int)qkv_bytes, 0x00020000);
    const uint32_t base_k = (uint32_t)(b * N) * ld + HD + h * 64;
    const int nkt = (N + TK - 1) / TK;
    auto stage = [&](int kt) {
        char* buf = smem + (kt % NB) * 2 * TILE_B;
        stage_rows(rs, buf, TK, N - kt * TK, base_k + (uint32_t)(kt * TK) * ld, ld, wave, 4, lane);
        stage_rows(rs, buf + TILE_B, TK, N - kt * TK, base_k + HD + (uint32_t)(kt * TK) * ld, ld, wave, 4, lane);
    };
#pragma unroll
    for (int i = 0; i < NB - 1; ++i)
        if (i < nkt) stage(i);

    const int q0 = qb * 128 + wave * 32;
    bf16x8 qf[2][2], df[2][2];
    float l2[2], dl[2];
    f32x4 dq[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int q = q0 + t * 16 + c, qc = min(q, N - 1);
        const uint16_t* qp = qkv + (size_t)(b * N + qc) * ld + h * 64 + 8 * G;
        const uint16_t* dop = dout + (size_t)(b * N + qc) * HD + h * 64 + 8 * G;
        const uint16_t* oop = out + (size_t)(b * N + qc) * HD + h * 64 + 8 * G;
        qf[t][0] = *(const bf16x8*)qp;
        qf[t][1] = *(const bf16x8*)(qp + 32);
        df[t][0] = *(const bf16x8*)dop;
        df[t][1] = *(const bf16x8*)(dop + 32);
        const bf16x8 of0 = *(const bf16x8*)oop, of1 = *(const bf16x8*)(oop + 32);
        float d = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) d += (float)df[t][0][j] * (float)of0[j] + (float)df[t][1][j] * (float)of1[j];
        dl[t] = group_sum(d);
        l2[t] = lse[((size_t)b * H + h) * N + qc] * LOG2E;
        if (q < N && G == 0) delta[((size_t)b * H + h) * N + q] = dl[t];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dq[t][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const float sl2 = scale * LOG2E;

    auto ktile = [&](int kt, auto masked_c) {
        constexpr bool MASKED = decltype(masked_c)::value;          // keys beyond N: last tile only
        wait_tile<NB>(nkt - 1 - kt);
        __syncthreads();
        if (kt + NB - 1 < nkt) stage(kt + NB - 1);
        const char* Ks = smem + (kt % NB) * 2 * TILE_B;
        const char* Vs = Ks + TILE_B;
        bf16x8 ktr[2][4];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) ktr[kk][dt] = tr_frag_raw(Ks, kk * 32, dt, lane);      // K^T for dQ: latency hides under S / dP
        bf16x8 dsf[2][2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x4 ds[4];
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                f32x4 s = mfma16(row_frag(Ks, s4 * 16, 0, lane), qf[t][0], (f32x4){0.f, 0.f, 0.f, 0.f});
                s = mfma16(row_frag(Ks, s4 * 16, 1, lane), qf[t][1], s);
                f32x4 dp = mfma16(row_frag(Vs, s4 * 16, 0, lane), df[t][0], (f32x4){0.f, 0.f, 0.f, 0.f});
                dp = mfma16(row_frag(Vs, s4 * 16, 1, lane), df[t][1], dp);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], sl2, -l2[t]));
                    if (MASKED && kt * TK + s4 * 16 + 4 * G + r >= N) p = 0.f;
                    ds[s4][r] = p * (dp[r] - dl[t]);
                }
            }
            dsf[t][0] = pack_pair(ds[0], ds[1]);
            dsf[t][1] = pack_pair(ds[2], ds[3]);
        }
        lds_tr_fence<true>();
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                dq[0][dt] = mfma16(ktr[kk][dt], dsf[0][kk], dq[0][dt]);
                dq[1][dt] = mfma16(ktr[kk][dt], dsf[1][kk], dq[1][dt]);
            }
    };
    for (int kt = 0; kt < nkt - 1; ++kt) ktile(kt, std::false_type{});
    ktile(nkt - 1, std::true_type{});
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int q = q0 + t * 16 + c;
        if (q < N) {
            uint16_t* op = dqkv + (size_t)(b * N + q) * ld + h * 64 + 4 * G;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) store_bf16x4(op + dt * 16, dq[t][dt], scale);
        }
    }
}

// ------------------------------------------------------------------------------------ backward: dK, dV
constexpr int TQ = 64;                 // queries per staged LDS tile (consumed as two sub-steps of 32)
constexpr int STEP_B = TQ * 128;       // 8 KiB per Q or dO tile
constexpr int DKV_BUF = 2 * STEP_B + 512;      // Q | dO | lse (64 f32) | delta (64 f32)
__global__ __launch_bounds__(512, 2) void attn_bwd_dkv_tiled_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ dout,
                                                                    const float* __restrict__ lse, const float* __restrict__ delta,
                                                                    uint16_t* __restrict__ dqkv, int N, int H, float scale,
                                                                    uint32_t qkv_bytes, uint32_t dout_bytes, int nkb) {
    __shared__ __attribute__((aligned(16))) char smem[NBUF * DKV_BUF];      // ring of 4 query tiles
    const int lane = threadIdx.x & 63, G = lane >> 4, c = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int bh = blockIdx.x / nkb, kb = blockIdx.x % nkb;
    const int b = bh / H, h = bh % H, HD = H * 64, ld = 3 * HD;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)qkv, 0, (int)qkv_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsd = __builtin_amdgcn_make_buffer_rsrc((void*)dout, 0, (int)dout_bytes, 0x00020000);
    const uint32_t stat_bytes = (uint32_t)((size_t)(N) * 4);      // one (b, h) row of lse / delta
    const __amdgpu_buffer_rsrc_t rsl = __builtin_amdgcn_make_buffer_rsrc((void*)(lse + ((size_t)b * H + h) * N), 0, (int)stat_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsm = __builtin_amdgcn_make_buffer_rsrc((void*)(delta + ((size_t)b * H + h) * N), 0, (int)stat_bytes, 0x00020000);
    const int nqs = (N + TQ - 1) / TQ;
    // Every wave issues exactly 4 LDS-DMA instructions per stage (its 1-KiB piece of Q and of dO, and -- redundantly, same bytes
    // to the same place -- the 256-B lse and delta rows), so one counted vmcnt serves all eight waves.
    auto stage = [&](int qs) {
        char* buf = smem + (qs % NBUF) * DKV_BUF;
        stage_rows(rs, buf, TQ, N - qs * TQ, (uint32_t)(b * N + qs * TQ) * ld + h * 64, ld, wave, 8, lane);
        stage_rows(rsd, buf + STEP_B, TQ, N - qs * TQ, (uint32_t)(b * N + qs * TQ) * HD + h * 64, HD, wave, 8, lane);
        const int q = qs * TQ + lane;
        const uint32_t voff = q < N ? (uint32_t)q * 4u : OOB_OFFSET;      // rows beyond N read as 0
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsl, (LDS_AS void*)(buf + 2 * STEP_B), 4, voff, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsm, (LDS_AS void*)(buf + 2 * STEP_B + 256), 4, voff, 0, 0, 0);
    };
    stage(0);
    if (nqs > 1) stage(1);
    if (nqs > 2) stage(2);
    // this wave's K / V fragments (B-operand layout: key on the lane): 2 key tiles of 16
    const int key0 = kb * 256 + wave * 32;
    bf16x8 kf[2][2], vf[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int kc = min(key0 + i * 16 + c, N - 1);
        const uint16_t* kp = qkv + (size_t)(b * N + kc) * ld + HD + h * 64 + 8 * G;
        kf[i][0] = *(const bf16x8*)kp;
        kf[i][1] = *(const bf16x8*)(kp + 32);
        vf[i][0] = *(const bf16x8*)(kp + HD);
        vf[i][1] = *(const bf16x8*)(kp + HD + 32);
    }
    f32x4 dk[2][4], dv[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dk[i][dt] = dv[i][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float sl2 = scale * LOG2E;

    auto qstep = [&](int qs, auto masked_c) {
        constexpr bool MASKED = decltype(masked_c)::value;          // query rows beyond N: last tile only
        wait_tile<NBUF>(nqs - 1 - qs);
        __syncthreads();
        if (qs + 3 < nqs) stage(qs + 3);
        const char* Qt = smem + (qs % NBUF) * DKV_BUF;
        const char* Dt = Qt + STEP_B;
        const float* st = (const float*)(Qt + 2 * STEP_B);
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const char* Qs = Qt + sub * 32 * 128;
            const char* Ds = Dt + sub * 32 * 128;
            bf16x8 dot[4], qtr[4];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                dot[dt] = tr_frag_raw(Ds, 0, dt, lane);       // dO^T, Q^T for dV / dK: asm reads, fenced before those products
                qtr[dt] = tr_frag_raw(Qs, 0, dt, lane);
            }
            f32x4 P[2][2], dS[2][2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const bf16x8 qa0 = row_frag(Qs, t * 16, 0, lane), qa1 = row_frag(Qs, t * 16, 1, lane);
                const bf16x8 da0 = row_frag(Ds, t * 16, 0, lane), da1 = row_frag(Ds, t * 16, 1, lane);
                const f32x4 lraw = *(const f32x4*)(st + sub * 32 + t * 16 + 4 * G), d4 = *(const f32x4*)(st + 64 + sub * 32 + t * 16 + 4 * G);
                const f32x4 l4 = {lraw[0] * LOG2E, lraw[1] * LOG2E, lraw[2] * LOG2E, lraw[3] * LOG2E};
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    f32x4 s = mfma16(qa0, kf[i][0], (f32x4){0.f, 0.f, 0.f, 0.f});
                    s = mfma16(qa1, kf[i][1], s);
                    f32x4 dp = mfma16(da0, vf[i][0], (f32x4){0.f, 0.f, 0.f, 0.f});
                    dp = mfma16(da1, vf[i][1], dp);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], sl2, -l4[r]));
                        if (MASKED && qs * TQ + sub * 32 + t * 16 + 4 * G + r >= N) p = 0.f;
                        P[t][i][r] = p;
                        dS[t][i][r] = p * (dp[r] - d4[r]);
                    }
                }
            }
            lds_tr_fence<true>();
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    dv[i][dt] = mfma16(dot[dt], pack_pair(P[0][i], P[1][i]), dv[i][dt]);
                    dk[i][dt] = mfma16(qtr[dt], pack_pair(dS[0][i], dS[1][i]), dk[i][dt]);
                }
            }
        }
    };
    for (int qs = 0; qs < nqs - 1; ++qs) qstep(qs, std::false_type{});
    qstep(nqs - 1, std::true_type{});
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int key = key0 + i * 16 + c;
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

}  // namespace

int attn_fwd_tiled_launch(const void* qkv, void* out, float* lse, int B, int N, int H, float scale, hipStream_t stream) {
    const int64_t bytes = (int64_t)B * N * 3 * H * 64 * 2;
    if (bytes >= (int64_t)OOB_OFFSET) return UNITE_ENOSUP;
    const int nqb = (N + 127) / 128;
    // ring depth 2, 64-key tiles: 32 KiB of LDS and 139 VGPRs keep three workgroups per CU, which hides the tile latency better
    // than a deeper ring (3 / 4 buffers: +10 %) or 128-key staged tiles (+9 %) at two workgroups per CU
    hipLaunchKernelGGL((attn_fwd_tiled_kernel<2, 1>), dim3(B * H * nqb), dim3(256), 0, stream, (const uint16_t*)qkv, (uint16_t*)out, lse, N, H, scale,
                       (uint32_t)bytes, nqb);
    UNITE_LAUNCH_CHECK();
    return UNITE_OK;
}

int attn_bwd_tiled_launch(const void* qkv, const void* out, const void* dout, const float* lse, float* delta, void* dqkv, int B, int N, int H,
                          float scale, hipStream_t stream) {
    const int64_t bytes = (int64_t)B * N * 3 * H * 64 * 2;
    if (bytes >= (int64_t)OOB_OFFSET) return UNITE_ENOSUP;
    const int nqb = (N + 127) / 128, nkb = (N + 255) / 256;
    hipLaunchKernelGGL(attn_bwd_dq_tiled_kernel<2>, dim3(B * H * nqb), dim3(256), 0, stream, (const uint16_t*)qkv, (const uint16_t*)out,
                       (const uint16_t*)dout, lse, delta, (uint16_t*)dqkv, N, H, scale, (uint32_t)bytes, nqb);
    UNITE_LAUNCH_CHECK();
    hipLaunchKernelGGL(attn_bwd_dkv_tiled_kernel, dim3(B * H * nkb), dim3(512), 0, stream, (const uint16_t*)qkv, (const uint16_t*)dout, lse,
                       delta, (uint16_t*)dqkv, N, H, scale, (uint32_t)bytes, (uint32_t)(bytes / 3), nkb);
    UNITE_LAUNCH_CHECK();
    return UNITE_OK;
}
