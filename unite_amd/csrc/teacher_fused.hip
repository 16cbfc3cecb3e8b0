// Fused QKV projection + attention of ONE (frame, head) per workgroup for the frozen CLIP teacher (clip.py:34-64, forward only).
//
// Unfused, every teacher block writes qkv [B*T*197, 3*D] bf16 (232 MB for CLIP-B/16 at B*T = 256) and the attention kernel reads
// it back as 128-byte row slices at a 4.6-KB stride: 464 MB of HBM traffic per block plus the projection GEMM's epilogue.  Here
// the workgroup computes its head's  [197 x 192] = LN(x)_frame [197 x D] . W_head^T  (q | k | v, 64 columns each) with MFMAs
// from LDS-DMA-staged K-tiles (3-deep ring of 52 KiB: 224 activation rows + 192 weight rows of 64 k), adds the bias, drops the
// bf16 q, k, v into LDS images and runs the exact-softmax attention of attention.hip on them; only O [197 x 64] leaves the CU.
// The MFMAs are issued with the operands swapped (weights as A, activations as B): lane (G, c16) then holds four CONSECUTIVE
// output columns of row c16, which leave as one packed 8-byte LDS store.
// Arithmetic is that of the unfused path (f32 accumulation over k in the same order, + bias, round to bf16, same attention code).
#include "attn_common.h"
#include <stdlib.h>

namespace {

constexpr int FQ_MROWS = 224, FQ_NT16 = 14;             // activation rows per frame (197 valid for a 14 x 14 grid + CLS), key tiles
constexpr int FQ_A_BYTES = FQ_MROWS * 128, FQ_B_BYTES = 192 * 128, FQ_STAGE = FQ_A_BYTES + FQ_B_BYTES;      // 28 + 24 = 52 KiB
constexpr int FQ_LDS = 3 * FQ_STAGE;
#ifndef FQ_DEBUG
#define FQ_DEBUG 0      // experiment builds only (-DFQ_DEBUG=n): 1 no attention, 2 no projection loop, 3 no MFMAs, 4 no MFMAs / fragment reads, 5 no DMA
#endif                    // 156 KiB ring; the q / k / v images (84 KiB) reuse it afterwards

__global__ __launch_bounds__(512, 2) void teacher_qkv_attn_kernel(const uint16_t* __restrict__ hin, const uint16_t* __restrict__ w_in,
                                                                  const float* __restrict__ b_in, uint16_t* __restrict__ out, int L, int H,
                                                                  int D, float scale, uint32_t h_bytes, uint32_t w_bytes, int order) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, G = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // XCD-aware order (workgroups b, b + 8, ... share an XCD): each XCD gets a contiguous range of (frame, head) units, so the 12
    // heads of a frame read its activation rows from one L2 instead of eight
    const int nwg = (int)gridDim.x, bid = blockIdx.x, xcd = bid & 7, qq = nwg >> 3, rr = nwg & 7;
    const int unit = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
    // Inside an XCD's range the units run in blocks of (8 frames x 4 heads), the heads of a frame group block after block: the 32 workgroups an
    // XCD holds at a time then read 8 x 302 KB of activation rows + 4 x 295 KB of weight rows (3.6 MB, inside its 4-MiB L2), and the next 32 --
    // the same frames, the next four heads -- find the activation rows still there.  Frame-major (the 12 heads of 2.7 frames at a time) every
    // set of 32 streams ALL of W (3.5 MB) past 0.8 MB of activations, nothing survives to the next set, and the L2 fetches 433 MB per launch
    // for 81 MB of operands (rocprofv3 FETCH_SIZE, profiles/r04_clock_notes.txt section 9).  FQ_ORDER 0 keeps the frame-major order (A/B).
    int frame, h;
    if (order == 0 || (H & 3)) { frame = unit / H; h = unit % H; }
    else {
        const int BT = nwg / H, per_fg = 8 * H;
        const int fg = unit / per_fg, v = unit - fg * per_fg;
        const int nf = min(8, BT - fg * 8);                 // the last frame group may be short
        const int hb = v / (nf * 4), w = v - hb * (nf * 4);
        frame = fg * 8 + w % nf;
        h = hb * 4 + w / nf;
    }
    const int wm = wave >> 2, wn = wave & 3;               // 8 waves: m-tiles wm*7 .. +7 (of 14), n-tiles wn*3 .. +3 (of 12)
    const __amdgpu_buffer_rsrc_t rsH = __builtin_amdgcn_make_buffer_rsrc((void*)hin, 0, (int)h_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)w_in, 0, (int)w_bytes, 0x00020000);
    const int nk = FQ_DEBUG == 2 ? 0 : D / 64;

    // one K-tile: 224 activation rows (rows >= L read as zero; 28 pieces of 1 KiB: 7 per wave of waves 0-3) + the head's 3 x 64
    // weight rows (24 pieces: 6 per wave of waves 4-7).  Lane (r8 = lane >> 3, p = lane & 7) of piece `it` moves 16 B of row
    // it * 8 + r8; its swizzled chunk index depends on the lane only, so a piece's offset is a per-lane constant + it * 8 rows.
    const int r8 = lane >> 3, p8 = lane & 7;
    const int lc16 = ((((p8 >> 1) ^ ((r8 >> 1) & 3)) << 1) | (p8 & 1));
    const int wq = wave & 3;
    const uint32_t a_off0 = ((uint32_t)(frame * L + wq * 8 + r8) * D + lc16 * 8) * 2u;             // piece wq, K-tile 0
    const uint32_t b_off0 = ((uint32_t)(h * 64 + wq * 8 + r8) * D + lc16 * 8) * 2u;                // q rows; k / v rows are + D * D elements
    // i-th DMA of this wave for K-tile t: activation stagers i < 7, weight stagers i < 6
    auto piece = [&](int t, int i, bool a_side) {
        if (FQ_DEBUG == 5) return;
        char* buf = smem + (t % 3) * FQ_STAGE;
        if (a_side) {
            const int it = wq + 4 * i;
            // L > 192: only the last piece (rows 192 + 8 wq ..) can run past the frame; those rows read as zero
            const uint32_t voff = (i < 6 || it * 8 + r8 < L) ? a_off0 + (uint32_t)(i * 32 * D + t * 64) * 2u : OOB_OFFSET;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsH, (LDS_AS void*)(buf + it * 1024), 16, voff, 0, 0, 0);
        } else {
            const int which = i >> 1, it = wq + 4 * (i & 1);
            const uint32_t voff = b_off0 + (uint32_t)(which * D * D + (i & 1) * 32 * D + t * 64) * 2u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (LDS_AS void*)(buf + FQ_A_BYTES + which * 8192 + it * 1024), 16, voff, 0, 0, 0);
        }
    };
    auto stage = [&](int t) {
        if (wave < 4) {
#pragma unroll
            for (int i = 0; i < 7; ++i) piece(t, i, true);
        } else {
#pragma unroll
            for (int i = 0; i < 6; ++i) piece(t, i, false);
        }
    };
    // K-tile `t` has landed for this wave once only the DMAs of the `younger` tiles issued after it are outstanding
    auto landed = [&](int younger) {
        if (younger <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (younger == 1) { if (wave < 4) asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
        else { if (wave < 4) asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); }
    };
    // fragment reads through untracked asm (common.h): lane-constant address per ks, m- / n-tile as immediate offset
    const int c = lane & 15;
    uint32_t la[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) la[ks] = lds_address(smem) + (uint32_t)row_addr(c, ks * 4 + G);
    const uint32_t a_tiles = (uint32_t)(wm * 7 * 2048), b_tiles = (uint32_t)(FQ_A_BYTES + wn * 3 * 2048);
    auto frags = [&](int t, int ks, bf16x8 (&xa)[7], bf16x8 (&wb)[3]) {
        if (FQ_DEBUG == 4) return;
        const uint32_t st = (uint32_t)((t % 3) * FQ_STAGE) + la[ks];
        const uint32_t pa = st + a_tiles, pb = st + b_tiles;
        static_for<0, 7>([&](auto i) { xa[decltype(i)::value] = lds_read_b128_raw<decltype(i)::value * 2048>(pa); });
        static_for<0, 3>([&](auto j) { wb[decltype(j)::value] = lds_read_b128_raw<decltype(j)::value * 2048>(pb); });
        __builtin_amdgcn_sched_barrier(0);                 // the MFMAs that follow stay behind the reads they are meant to cover
    };
    f32x4 acc[7][3];
#pragma unroll
    for (int i = 0; i < 7; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto mm = [&](const bf16x8 (&xa)[7], const bf16x8 (&wb)[3]) {
#pragma unroll
        for (int i = 0; i < 7; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                if (FQ_DEBUG == 3 || FQ_DEBUG == 4) asm volatile("" ::"v"(wb[j]), "v"(xa[i]));
                else acc[i][j] = mfma16(wb[j], xa[i], acc[i][j]);      // C^T: rows = output columns
            }
    };

    // the same with this wave's DMAs for K-tile t_new spread between the MFMA rows (one DMA per three MFMAs)
    auto mm_dma = [&](const bf16x8 (&xa)[7], const bf16x8 (&wb)[3], int t_new, auto a_side_c) {
        constexpr bool A_SIDE = decltype(a_side_c)::value;
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            if (A_SIDE || i < 6) piece(t_new, i, A_SIDE);
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                if (FQ_DEBUG == 3 || FQ_DEBUG == 4) asm volatile("" ::"v"(wb[j]), "v"(xa[i]));
                else acc[i][j] = mfma16(wb[j], xa[i], acc[i][j]);
            }
        }
    };

    // Three K-tiles in the ring; the fragments of the next half K-tile are read from LDS while the MFMAs of the current one run
    // (two register sets), one barrier per K-tile, after which the tile just finished (t) may be refilled with K-tile t + 3.
    // Waves w and w + 4 share a SIMD: the activation stagers (0-3) spread their DMAs over the second-half MFMAs of iteration t,
    // the weight stagers (4-7) over the first-half MFMAs of iteration t + 1, so one of the two is always issuing plain MFMAs.
    // Either way K-tiles up to t + 2 have been issued when a wave waits for K-tile t + 1.
    // Raw barriers: __syncthreads() would put "s_waitcnt vmcnt(0)" in front (the compiler makes the workgroup fence wait for the
    // LDS-DMA it sees in flight), i.e. drain the whole prefetch queue every K-tile.
    stage(0);
    if (nk > 1) stage(1);
    if (nk > 2) stage(2);
    landed(min(nk - 1, 2));
    __builtin_amdgcn_s_barrier();
    bf16x8 xa0[7], wb0[3], xa1[7], wb1[3];
    frags(0, 0, xa0, wb0);
    lds_tr_fence<true>();
    // one K-tile step; DMA_FIRST: weight stager, spreads K-tile t + 2 over the first half; DMA_SECOND: activation stager, K-tile t + 3
    auto step = [&](int t, auto dma_first_c, auto dma_second_c) {
        frags(t, 1, xa1, wb1);                             // in flight under the MFMAs of the first half
        if constexpr (decltype(dma_first_c)::value) mm_dma(xa0, wb0, t + 2, std::false_type{});      // buffer of K-tile t - 1
        else mm(xa0, wb0);
        landed(min(nk - 2 - t, 1));                        // K-tile t + 1; only t + 2 was issued after it
        lds_tr_fence<true>();                              // second-half fragments are in; this wave is done reading K-tile t
        __builtin_amdgcn_s_barrier();
        frags(t + 1, 0, xa0, wb0);                         // in flight under the MFMAs of the second half
        if constexpr (decltype(dma_second_c)::value) mm_dma(xa1, wb1, t + 3, std::true_type{});      // buffer of K-tile t
        else mm(xa1, wb1);
        lds_tr_fence<true>();
    };
    // (every wave passes the same number of barriers: nk - 1 steps in either branch)
    if (wave < 4) {
        int t = 0;
        for (; t + 3 < nk; ++t) step(t, std::false_type{}, std::true_type{});
        for (; t + 1 < nk; ++t) step(t, std::false_type{}, std::false_type{});
    } else {
        int t = 0;
        if (nk > 1) step(t++, std::false_type{}, std::false_type{});
        for (; t + 2 < nk; ++t) step(t, std::true_type{}, std::false_type{});
        for (; t + 1 < nk; ++t) step(t, std::false_type{}, std::false_type{});
    }
    frags(nk - 1, 1, xa1, wb1);
    mm(xa0, wb0);
    lds_tr_fence<true>();
    mm(xa1, wb1);
    __syncthreads();                                       // the ring is free: q | k | v images [224][64] bf16 go to its start

    // ---- bias, bf16, LDS images (attention.hip layout: 128-B rows, 32-B chunk c at c ^ ((row >> 1) & 3))
    char* const Qs = smem;
    char* const Ks = smem + FQ_MROWS * 128;
    char* const Vs = smem + 2 * FQ_MROWS * 128;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int nt = wn * 3 + j, which = nt >> 2, dchunk = nt & 3;
        const f32x4 bias = *(const f32x4*)(b_in + which * D + h * 64 + dchunk * 16 + 4 * G);
        char* img = which == 0 ? Qs : (which == 1 ? Ks : Vs);
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const int row = (wm * 7 + i) * 16 + c;
            const f32x4 v = acc[i][j];
            *(u32x2*)(img + row_addr(row, dchunk * 2 + (G >> 1)) + (G & 1) * 8) =
                (u32x2){pack_bf16x2(v[0] + bias[0], v[1] + bias[1]), pack_bf16x2(v[2] + bias[2], v[3] + bias[3])};
        }
    }
    __syncthreads();

    // ---- attention over the LDS-resident head (attention.hip's forward with Q from LDS as well; no LSE: the teacher has no backward)
    constexpr int NT16 = FQ_NT16;
    const float sl2 = scale * LOG2E;
    const int nqt = (L + 15) / 16;
    for (int qt = wave; qt < (FQ_DEBUG == 1 ? 0 : nqt); qt += 8) {
        const int q = qt * 16 + c;
        const bf16x8 qf0 = row_frag(Qs, qt * 16, 0, lane), qf1 = row_frag(Qs, qt * 16, 1, lane);
        f32x4 st[NT16];
        float m = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NT16; ++kt) {
            f32x4 s = mfma16(row_frag(Ks, kt * 16, 0, lane), qf0, (f32x4){0.f, 0.f, 0.f, 0.f});
            s = mfma16(row_frag(Ks, kt * 16, 1, lane), qf1, s);
            if (kt >= NT16 - 2) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (kt * 16 + 4 * G + r >= L) s[r] = -INFINITY;
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
                const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(st[kt][r], sl2, -ml2));
                st[kt][r] = p;
                sum4[r] += p;
            }
        const float sum = group_sum((sum4[0] + sum4[1]) + (sum4[2] + sum4[3]));
        f32x4 o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < NT16 / 2; ++kk) {
            const bf16x8 pf = pack_pair(st[2 * kk], st[2 * kk + 1]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) o[dt] = mfma16(tr_frag(Vs, kk * 32, dt, lane), pf, o[dt]);
        }
        if (q < L) {
            const float inv = 1.0f / sum;
            uint16_t* op = out + (size_t)(frame * L + q) * D + h * 64 + 4 * G;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) store_bf16x4(op + dt * 16, o[dt], inv);
        }
    }
}

}  // namespace

bool unite_prof_begin(hipStream_t s);                  // gemm.hip: bench.py's launch-timing pool
void unite_prof_end(hipStream_t s, double flops, double bytes);

extern "C" int unite_teacher_qkv_attn(const void* h, const void* w_in, const float* b_in, void* out, int32_t BT, int32_t L, int32_t H,
                                      int32_t D, float scale, void* stream) {
    if (!h || !w_in || !b_in || !out || BT <= 0 || H <= 0 || D != H * 64) return UNITE_EINVAL;
    if (L <= 192 || L > FQ_MROWS) return UNITE_ENOSUP;              // built for 14 key tiles (197 tokens: 14 x 14 patches + CLS)
    const int64_t h_bytes = (int64_t)BT * L * D * 2, w_bytes = (int64_t)3 * D * D * 2;
    if (h_bytes >= (int64_t)OOB_OFFSET || w_bytes >= (int64_t)OOB_OFFSET || (((uintptr_t)b_in) & 15)) return UNITE_ENOSUP;
    static bool lds_ok = false;
    if (!lds_ok) {
        hipError_t e = hipFuncSetAttribute((const void*)teacher_qkv_attn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, FQ_LDS);
        if (e != hipSuccess) return (int)e;
        lds_ok = true;
    }
    static const int order = getenv("UNITE_TEACHER_FQ_ORDER") ? atoi(getenv("UNITE_TEACHER_FQ_ORDER")) : 1;      // 0: frame-major units, 1: (8 frames x 4 heads) blocks
    const bool prof = unite_prof_begin((hipStream_t)stream);
    hipLaunchKernelGGL(teacher_qkv_attn_kernel, dim3(BT * H), dim3(512), FQ_LDS, (hipStream_t)stream, (const uint16_t*)h, (const uint16_t*)w_in,
                       b_in, (uint16_t*)out, L, H, D, scale, (uint32_t)h_bytes, (uint32_t)w_bytes, order);
    if (prof) unite_prof_end((hipStream_t)stream, 2.0 * BT * L * 3.0 * D * D + 4.0 * BT * H * (double)L * L * 64.0,
                             2.0 * (2.0 * BT * L * D + 3.0 * D * D) + 12.0 * D);      // h in, out, w_in, b_in
    UNITE_LAUNCH_CHECK();
    return UNITE_OK;
}
