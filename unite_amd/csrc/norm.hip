// Row-wise normalisation kernels for gfx950: one 64-lane wavefront per row, the row held in registers
// (D <= 1024 -> <= 16 floats per lane, read as float4), fp32 statistics by wavefront shuffles.
// All of them are HBM-bound streaming kernels: 16-byte loads, 8/16-byte stores, no LDS except for the
// per-workgroup partial sums of the gamma/beta gradients.
#include "common.h"
#include <stdint.h>

namespace {

// rows per workgroup of the LayerNorm / decoder-tail backward (4 waves, one row each per iteration): 16 up to 16 k rows, then
// grown so that at most ~1024 partial-sum rows reach the second-stage reduction (50 176 tokens of stage 2: 52 rows per workgroup,
// 965 partials instead of 3 136 -- the reduction over those took 189 us per call)
__host__ __device__ inline int bwd_rows_per_block(int M) {
    const int r = (M + 1023) / 1024;
    return r <= 16 ? 16 : (r + 3) / 4 * 4;
}

template <int NV>
__device__ __forceinline__ void load_row(const float* row, int D, int lane, f32x4 (&v)[NV]) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        v[i] = (c < D) ? *(const f32x4*)(row + c) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
}

// the same row from a bf16 matrix (the frozen teacher's bf16 residual stream): 8 bytes per lane and piece
template <int NV>
__device__ __forceinline__ void load_row_bf16(const uint16_t* row, int D, int lane, f32x4 (&v)[NV]) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < D) {
            const u32x2 w = *(const u32x2*)(row + c);
            v[i] = (f32x4){__uint_as_float(w[0] << 16), __uint_as_float(w[0] & 0xFFFF0000u), __uint_as_float(w[1] << 16),
                           __uint_as_float(w[1] & 0xFFFF0000u)};
        } else v[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
}

template <int NV>
__device__ __forceinline__ void load_row_f16(const uint16_t* row, int D, int lane, f32x4 (&v)[NV]) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < D) {
            const u32x2 w = *(const u32x2*)(row + c);
            v[i] = (f32x4){unpack_f16_lo(w[0]), unpack_f16_hi(w[0]), unpack_f16_lo(w[1]), unpack_f16_hi(w[1])};
        } else v[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
}

template <int NV>
__device__ __forceinline__ void row_stats(const f32x4 (&v)[NV], int D, int lane, float eps, float& mean, float& rstd) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
    mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < D) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mean; q += d * d; }
        }
    }
    rstd = rsqrtf(wave_sum(q) / (float)D + eps);
}

__device__ __forceinline__ void store4(void* y, int y_f32, size_t off, f32x4 o) {
    if (y_f32 == 1) *(f32x4*)((float*)y + off) = o;
    else if (y_f32 == 2) *(u32x2*)((uint16_t*)y + off) = (u32x2){pack_f16x2(o[0], o[1]), pack_f16x2(o[2], o[3])};      // IEEE half (teacher stream)
    else *(u32x2*)((uint16_t*)y + off) = (u32x2){pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
}

// ------------------------------------------------------------------------------------ forward
template <int NV, int XK>      // XK: the input rows are 0 f32, 1 bf16, 2 IEEE half
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const void* __restrict__ x, int ldx, const int32_t* __restrict__ row_index,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                            const float* __restrict__ post_add, void* __restrict__ y, int y_f32,
                                                            float* __restrict__ mean_out, float* __restrict__ rstd_out, int M, int D) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int src = row_index ? row_index[row] : row;
    f32x4 v[NV];
    if (XK == 1) load_row_bf16<NV>((const uint16_t*)x + (size_t)src * ldx, D, lane, v);
    else if (XK == 2) load_row_f16<NV>((const uint16_t*)x + (size_t)src * ldx, D, lane, v);
    else load_row<NV>((const float*)x + (size_t)src * ldx, D, lane, v);
    float mean, rstd;
    row_stats<NV>(v, D, lane, eps, mean, rstd);
    if (lane == 0) {
        if (mean_out) mean_out[row] = mean;
        if (rstd_out) rstd_out[row] = rstd;
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < D) {
            const f32x4 g = *(const f32x4*)(gamma + c), b = *(const f32x4*)(beta + c);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean) * rstd * g[e] + b[e];
            if (post_add) {
                const f32x4 a = *(const f32x4*)(post_add + (size_t)row * D + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] += a[e];
            }
            store4(y, y_f32, (size_t)row * D + c, o);
        }
    }
}

// ------------------------------------------------------------------------------------ backward
// dx = rstd * (g*dy - mean_D(g*dy) - xhat * mean_D(g*dy*xhat));  dgamma = sum_rows dy*xhat; dbeta = sum_rows dy
// FLAGS: -1 = every option is a run-time test (generic); otherwise bit 0 dy is f32, bit 1 a residual gradient is added, bit 2 the f32 result is
// stored, bit 3 the bf16 copy (+ its column sums) is stored, bit 4 a row scale applies to the copy -- known at compile time for the combinations the
// training steps use (round 4: run-time tests on kernel arguments inside unrolled per-element loops had turned out to be a branch per element in
// the GEMM epilogues, profiles/r04_clock_notes.txt section 18)
template <int NV, int FLAGS = -1>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const void* __restrict__ dy, int dy_f32_rt, const float* __restrict__ x, int ldx,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma, const float* __restrict__ dx_residual,
                                                            float* __restrict__ dx_out, void* __restrict__ dx_bf16,
                                                            const float* __restrict__ row_scale, int rows_per_scale,
                                                            float* __restrict__ partial, int M, int D) {
    __shared__ float red[4][3][NV * 256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool dy_f32 = FLAGS >= 0 ? (FLAGS & 1) != 0 : dy_f32_rt != 0;
    if (FLAGS >= 0) {                                      // the pointers a known combination does not use are dead from here on
        if (!(FLAGS & 2)) dx_residual = nullptr;
        if (!(FLAGS & 4)) dx_out = nullptr;
        if (!(FLAGS & 8)) dx_bf16 = nullptr;
        if (!(FLAGS & 16)) row_scale = nullptr;
    }
    const bool has_res = FLAGS >= 0 ? (FLAGS & 2) != 0 : dx_residual != nullptr;
    const bool has_out = FLAGS >= 0 ? (FLAGS & 4) != 0 : dx_out != nullptr;
    const bool has_bf = FLAGS >= 0 ? (FLAGS & 8) != 0 : dx_bf16 != nullptr;
    const bool has_sc = FLAGS >= 0 ? (FLAGS & 16) != 0 : row_scale != nullptr;
    f32x4 gam[NV], dg[NV], db[NV], ds[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        gam[i] = (c < D) ? *(const f32x4*)(gamma + c) : (f32x4){0.f, 0.f, 0.f, 0.f};
        dg[i] = db[i] = ds[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const int rpb = bwd_rows_per_block(M);
    for (int r = 0; r < rpb / 4; ++r) {
        const int row = blockIdx.x * rpb + r * 4 + wave;
        if (row >= M) break;
        f32x4 xv[NV], dyv[NV], resv[NV];
        load_row<NV>(x + (size_t)row * ldx, D, lane, xv);
        if (has_res) load_row<NV>(dx_residual + (size_t)row * D, D, lane, resv);      // with the other loads: one memory round trip per row
        if (dy_f32) load_row<NV>((const float*)dy + (size_t)row * D, D, lane, dyv);
        else {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = (i * 64 + lane) * 4;
                if (c < D) {
                    const u32x2 w = *(const u32x2*)((const uint16_t*)dy + (size_t)row * D + c);
                    dyv[i] = (f32x4){__uint_as_float(w[0] << 16), __uint_as_float(w[0] & 0xFFFF0000u),
                                     __uint_as_float(w[1] << 16), __uint_as_float(w[1] & 0xFFFF0000u)};
                } else dyv[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
        const float mu = mean[row], rs = rstd[row];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float xh = (xv[i][e] - mu) * rs;
                const float gd = gam[i][e] * dyv[i][e];
                s1 += gd;
                s2 += gd * xh;
                dg[i][e] += dyv[i][e] * xh;
                db[i][e] += dyv[i][e];
                xv[i][e] = xh;
            }
        s1 = wave_sum(s1) / (float)D;
        s2 = wave_sum(s2) / (float)D;
        const float sc = has_sc ? row_scale[row / rows_per_scale] : 1.0f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            if (c < D) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = rs * (gam[i][e] * dyv[i][e] - s1 - xv[i][e] * s2);
                if (has_res) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] += resv[i][e];
                }
                if (has_out) *(f32x4*)(dx_out + (size_t)row * D + c) = o;
                if (has_bf) {
                    const u32x2 w = (u32x2){pack_bf16x2(sc * o[0], sc * o[1]), pack_bf16x2(sc * o[2], sc * o[3])};
                    *(u32x2*)((uint16_t*)dx_bf16 + (size_t)row * D + c) = w;
                    // column sums of exactly what the consuming GEMMs read (its bias gradient)
                    ds[i][0] += __uint_as_float(w[0] << 16);
                    ds[i][1] += __uint_as_float(w[0] & 0xFFFF0000u);
                    ds[i][2] += __uint_as_float(w[1] << 16);
                    ds[i][3] += __uint_as_float(w[1] & 0xFFFF0000u);
                }
            }
        }
    }
    // workgroup partial sums -> partial[block][3][D]
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            red[wave][0][(i * 64 + lane) * 4 + e] = dg[i][e];
            red[wave][1][(i * 64 + lane) * 4 + e] = db[i][e];
            red[wave][2][(i * 64 + lane) * 4 + e] = ds[i][e];
        }
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += 256) {
#pragma unroll
        for (int k = 0; k < 3; ++k)
            partial[((size_t)blockIdx.x * 3 + k) * D + c] = red[0][k][c] + red[1][k][c] + red[2][k][c] + red[3][k][c];
    }
}

// out_k[c] (+)= sum_b partial[b][k][c], k = 0..2; fixed order -> deterministic.
// Grid (D / 32, 3): one workgroup per output vector k and 32 columns; thread (r, c) = (tid >> 5, tid & 31) sums partial rows
// r, r + 8, ... (128-B coalesced, 8 independent loads in flight per thread), then the 8 row groups are combined through LDS in a
// fixed order.  Round 3 tried to fold this second stage into the backward kernels themselves (groups of 32 workgroups, the last one of
// a group to finish adds the group's partials, the last group adds the groups: common.h arrive_last): layernorm_bwd went from 28.9 to
// 73.5 us per call at 10 240 rows -- 640 x 9 KB of partials want ~70 workgroups reading in parallel, and only the ONE workgroup that
// arrives last can know that the others are done -- against 28.9 + 5.7 us for the two launches.  Reverted; the split-K slabs of the
// weight-gradient GEMMs (a few tiles per slab set) and the column sums do fold into their kernels (gemm.hip, misc.hip).
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ partial, int nblocks, int D,
                                                              float* __restrict__ out_a, float* __restrict__ out_b, float* __restrict__ out_c,
                                                              int accumulate) {
    __shared__ float red[8][33];
    const int k = blockIdx.y;
    float* out = k == 0 ? out_a : (k == 1 ? out_b : out_c);
    if (!out) return;                                   // workgroup-uniform
    const int cl = threadIdx.x & 31, r = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    float acc = 0.f;
    if (c < D) {
        const float* p = partial + (size_t)k * D + c;
        const size_t stride = (size_t)3 * D;
        int i = r;
        for (; i + 56 < nblocks; i += 64) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(i + 8 * u) * stride];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
        }
        for (; i < nblocks; i += 8) acc += p[(size_t)i * stride];
    }
    red[r][cl] = acc;
    __syncthreads();
    if (threadIdx.x < 32 && c < D) {
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) v += red[i][cl];
        // accumulate: bit 0 -> out_a/out_b (gamma/beta gradients), bit 1 -> out_c (bias column sums)
        const int accf = k < 2 ? (accumulate & 1) : (accumulate & 2);
        out[c] = accf ? out[c] + v : v;
    }
}

// ------------------------------------------------------------------------------------ decoder tail
// u = LN(y)*g+b ; o = u/||u|| ; loss += 2 - 2<o,t>
template <int NV>
__global__ __launch_bounds__(256) void decoder_tail_fwd_kernel(const float* __restrict__ y, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float eps, const float* __restrict__ tgt,
                                                               float* __restrict__ out, float* __restrict__ loss_sum, int M, int C) {
    __shared__ float wl[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float l = 0.f;
    // grid-stride over rows: one atomicAdd per WORKGROUP at the end (one per 4 rows serialised 2 560 atomics on one address:
    // 39 us per call for 42 MB of traffic)
    for (int row = blockIdx.x * 4 + wave; row < M; row += gridDim.x * 4) {
        f32x4 v[NV];
        load_row<NV>(y + (size_t)row * C, C, lane, v);
        float mean, rstd;
        row_stats<NV>(v, C, lane, eps, mean, rstd);
        float nn = 0.f, dt = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            if (c < C) {
                const f32x4 g = *(const f32x4*)(gamma + c), b = *(const f32x4*)(beta + c);
                f32x4 t = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (tgt) t = *(const f32x4*)(tgt + (size_t)row * C + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float u = (v[i][e] - mean) * rstd * g[e] + b[e];
                    v[i][e] = u;
                    nn += u * u;
                    dt += u * t[e];
                }
            }
        }
        nn = wave_sum(nn);
        dt = wave_sum(dt);
        const float inv = rsqrtf(nn);
        if (out) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = (i * 64 + lane) * 4;
                if (c < C) *(f32x4*)(out + (size_t)row * C + c) = (f32x4){v[i][0] * inv, v[i][1] * inv, v[i][2] * inv, v[i][3] * inv};
            }
        }
        l += 2.0f - 2.0f * dt * inv;
    }
    if (loss_sum && tgt) {
        if (lane == 0) wl[wave] = l;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(loss_sum, wl[0] + wl[1] + wl[2] + wl[3]);
    }
}

template <int NV>
__global__ __launch_bounds__(256) void decoder_tail_bwd_kernel(const float* __restrict__ y, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float eps, const float* __restrict__ tgt,
                                                               float loss_scale, const float* __restrict__ loss_scale_dev,
                                                               const float* __restrict__ dout, void* __restrict__ dy_bf16,
                                                               float* __restrict__ partial, int M, int C) {
    __shared__ float red[4][3][NV * 256];
    if (loss_scale_dev) loss_scale *= *loss_scale_dev;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 gam[NV], bet[NV], dg[NV], db[NV], ds[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        gam[i] = (c < C) ? *(const f32x4*)(gamma + c) : (f32x4){0.f, 0.f, 0.f, 0.f};
        bet[i] = (c < C) ? *(const f32x4*)(beta + c) : (f32x4){0.f, 0.f, 0.f, 0.f};
        dg[i] = db[i] = ds[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const int rpb = bwd_rows_per_block(M);
    for (int r = 0; r < rpb / 4; ++r) {
        const int row = blockIdx.x * rpb + r * 4 + wave;
        if (row >= M) break;
        f32x4 v[NV], u[NV], dov[NV];
        load_row<NV>(y + (size_t)row * C, C, lane, v);
        float mean, rstd;
        row_stats<NV>(v, C, lane, eps, mean, rstd);
        if (dout) load_row<NV>(dout + (size_t)row * C, C, lane, dov);
        else {
            load_row<NV>(tgt + (size_t)row * C, C, lane, dov);
#pragma unroll
            for (int i = 0; i < NV; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) dov[i][e] *= -2.0f * loss_scale;
        }
        float nn = 0.f, od = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float xh = (v[i][e] - mean) * rstd;
                const float uu = (c < C) ? xh * gam[i][e] + bet[i][e] : 0.f;
                v[i][e] = (c < C) ? xh : 0.f;
                u[i][e] = uu;
                nn += uu * uu;
                od += uu * dov[i][e];
            }
        }
        nn = wave_sum(nn);
        od = wave_sum(od);
        const float inv = rsqrtf(nn);
        // o = u*inv ; du = inv * (do - o <o,do>) = inv*do - u * inv^3 * <u,do>
        const float k2 = inv * inv * inv * od;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float du = inv * dov[i][e] - u[i][e] * k2;
                dg[i][e] += du * v[i][e];
                db[i][e] += du;
                const float gd = gam[i][e] * du;
                u[i][e] = gd;
                s1 += gd;
                s2 += gd * v[i][e];
            }
        s1 = wave_sum(s1) / (float)C;
        s2 = wave_sum(s2) / (float)C;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            if (c < C) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = rstd * (u[i][e] - s1 - v[i][e] * s2);
                const u32x2 w = (u32x2){pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
                *(u32x2*)((uint16_t*)dy_bf16 + (size_t)row * C + c) = w;
                ds[i][0] += __uint_as_float(w[0] << 16);
                ds[i][1] += __uint_as_float(w[0] & 0xFFFF0000u);
                ds[i][2] += __uint_as_float(w[1] << 16);
                ds[i][3] += __uint_as_float(w[1] & 0xFFFF0000u);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            red[wave][0][(i * 64 + lane) * 4 + e] = dg[i][e];
            red[wave][1][(i * 64 + lane) * 4 + e] = db[i][e];
            red[wave][2][(i * 64 + lane) * 4 + e] = ds[i][e];
        }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
#pragma unroll
        for (int k = 0; k < 3; ++k)
            partial[((size_t)blockIdx.x * 3 + k) * C + c] = red[0][k][c] + red[1][k][c] + red[2][k][c] + red[3][k][c];
    }
}

// ------------------------------------------------------------------------------------ CLIP embed + ln_pre
template <int NV>
__global__ __launch_bounds__(256) void clip_embed_ln_kernel(const uint16_t* __restrict__ patches, const float* __restrict__ cls,
                                                            const float* __restrict__ pos, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float eps, void* __restrict__ x,
                                                            int x_f32, int BT, int HW, int D) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int L = HW + 1;
    if (row >= BT * L) return;
    const int bt = row / L, j = row % L;
    f32x4 v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        v[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (c < D) {
            if (j == 0) v[i] = *(const f32x4*)(cls + c);
            else {
                const u32x2 w = *(const u32x2*)(patches + ((size_t)bt * HW + (j - 1)) * D + c);
                v[i] = (f32x4){__uint_as_float(w[0] << 16), __uint_as_float(w[0] & 0xFFFF0000u),
                               __uint_as_float(w[1] << 16), __uint_as_float(w[1] & 0xFFFF0000u)};
            }
            const f32x4 pe = *(const f32x4*)(pos + (size_t)j * D + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[i][e] += pe[e];
        }
    }
    float mean, rstd;
    row_stats<NV>(v, D, lane, eps, mean, rstd);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < D) {
            const f32x4 g = *(const f32x4*)(gamma + c), b = *(const f32x4*)(beta + c);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean) * rstd * g[e] + b[e];
            store4(x, x_f32, (size_t)row * D + c, o);
        }
    }
}

template <int NV>
__global__ __launch_bounds__(256) void l2_normalize_kernel(float* __restrict__ x, int M, int D) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    f32x4 v[NV];
    load_row<NV>(x + (size_t)row * D, D, lane, v);
    float nn = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) nn += v[i][0] * v[i][0] + v[i][1] * v[i][1] + v[i][2] * v[i][2] + v[i][3] * v[i][3];
    const float inv = rsqrtf(wave_sum(nn));
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < D) *(f32x4*)(x + (size_t)row * D + c) = (f32x4){v[i][0] * inv, v[i][1] * inv, v[i][2] * inv, v[i][3] * inv};
    }
}

inline int nv_for(int D) { return (D + 255) / 256; }
inline bool dim_ok(int D) { return D > 0 && D <= 1024 && (D & 3) == 0; }

}  // namespace

#define DISPATCH_NV(D, CALL)                  \
    switch (nv_for(D)) {                      \
        case 1: { constexpr int NV = 1; CALL; } break; \
        case 2: { constexpr int NV = 2; CALL; } break; \
        case 3: { constexpr int NV = 3; CALL; } break; \
        default: { constexpr int NV = 4; CALL; } break; \
    }

extern "C" int unite_layernorm_fwd(const float* x, int32_t ldx, const int32_t* row_index, const float* gamma, const float* beta,
                                   float eps, const float* post_add, void* y, int32_t y_f32, float* mean, float* rstd,
                                   int32_t M, int32_t D, void* stream) {
    if (!x || !gamma || !beta || !y || M <= 0 || !dim_ok(D) || (ldx & 3)) return UNITE_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    DISPATCH_NV(D, hipLaunchKernelGGL((layernorm_fwd_kernel<NV, 0>), dim3((M + 3) / 4), dim3(256), 0, s, (const void*)x, ldx, row_index,
                                      gamma, beta, eps, post_add, y, y_f32, mean, rstd, M, D));
    UNITE_LAUNCH_CHECK();
    return UNITE_OK;
}

extern "C" int unite_layernorm_fwd_bf16in(const void* x, int32_t ldx, const int32_t* row_index, const float* gamma, const float* beta,
                                          float eps, const float* post_add, void* y, int32_t y_f32, float* mean, float* rstd,
                                          int32_t M, int32_t D, void* stream) {
    if (!x || !gamma || !beta || !y || M <= 0 || !dim_ok(D) || (ldx & 3) || (((uintptr_t)x) & 7)) return UNITE_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    DISPATCH_NV(D, hipLaunchKernelGGL((layernorm_fwd_kernel<NV, 1>), dim3((M + 3) / 4), dim3(256), 0, s, x, ldx, row_index, gamma,
                                      beta, eps, post_add, y, y_f32, mean, rstd, M, D));
    UNITE_LAUNCH_CHECK();
    return UNITE_OK;
}

extern "C" int unite_layernorm_fwd_f16in(const void* x, int32_t ldx, const int32_t* row_index, const float* gamma, const float* beta,
                                         float eps, const float* post_add, void* y, int32_t y_f32, float* mean, float* rstd,
                                         int32_t M, int32_t D, void* stream) {
    if (!x || !gamma || !beta || !y || M <= 0 || !dim_ok(D) || (ldx & 3) || (((uintptr_t)x) & 7)) return UNITE_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    DISPATCH_NV(D, hipLaunchKernelGGL((layernorm_fwd_kernel<NV, 2>), dim3((M + 3) / 4), dim3(256), 0, s, x, ldx, row_index, gamma,
                                      beta, eps, post_add, y, y_f32, mean, rstd, M, D));
    UNITE_LAUNCH_CHECK();
    return UNITE_OK;
}

extern "C" size_t unite_layernorm_bwd_workspace(int32_t M, int32_t D) {
    const size_t nb = (size_t)(M + bwd_rows_per_block(M) - 1) / bwd_rows_per_block(M);
    return nb * 3 * (size_t)D * sizeof(float);
}

extern "C" int unite_layernorm_bwd(const void* dy, int32_t dy_f32, const float* x, int32_t ldx, const float* mean, const float* rstd,
                                   const float* gamma, const float* dx_residual, float* dx_out, void* dx_bf16,
                                   const float* row_scale, int32_t rows_per_scale, float* dgamma, float* dbeta, float* dxsum,
                                   int32_t accumulate, void* workspace, int32_t M, int32_t D, void* stream) {
    if (!dy || !x || !mean || !rstd || !gamma || !workspace || M <= 0 || !dim_ok(D) || (ldx & 3)) return UNITE_EINVAL;
    if (row_scale && rows_per_scale <= 0) return UNITE_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int nb = (M + bwd_rows_per_block(M) - 1) / bwd_rows_per_block(M);
    if (dxsum && !dx_bf16) return UNITE_EINVAL;
    // the option combinations of the training steps as compile-time forms (ViT blocks: bf16 dy, residual gradient, f32 + bf16 results, with or
    // without a drop-path scale), anything else through the generic form
    const int flags = (dy_f32 ? 1 : 0) | (dx_residual ? 2 : 0) | (dx_out ? 4 : 0) | (dx_bf16 ? 8 : 0) | (row_scale ? 16 : 0);
#define LN_BWD_FORM(F) DISPATCH_NV(D, hipLaunchKernelGGL((layernorm_bwd_kernel<NV, F>), dim3(nb), dim3(256), 0, s, dy, dy_f32, x, ldx, mean, rstd, gamma, \
                                                         dx_residual, dx_out, dx_bf16, row_scale, rows_per_scale, (float*)workspace, M, D))
    if (flags == (2 | 4 | 8)) { LN_BWD_FORM(14); }
    else if (flags == (2 | 4 | 8 | 16)) { LN_BWD_FORM(30); }
    else if (flags == (2 | 4)) { LN_BWD_FORM(6); }
    else { LN_BWD_FORM(-1); }
#undef LN_BWD_FORM
    UNITE_LAUNCH_CHECK();
    if (dgamma || dbeta || dxsum) {
        hipLaunchKernelGGL(reduce_partials_kernel, dim3((D + 31) / 32, 3), dim3(256), 0, s, (const float*)workspace, nb, D, dgamma,
                           dbeta, dxsum, accumulate);
        UNITE_LAUNCH_CHECK();
    }
    return UNITE_OK;
}

extern "C" int unite_decoder_tail_fwd(const float* y, const float* gamma, const float* beta, float eps, const float* tgt, float* out,
                                      float* loss_sum, int32_t M, int32_t C, void* stream) {
    if (!y || !gamma || !beta || M <= 0 || !dim_ok(C)) return UNITE_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int blocks = (M + 3) / 4 < 1024 ? (M + 3) / 4 : 1024;
    DISPATCH_NV(C, hipLaunchKernelGGL((decoder_tail_fwd_kernel<NV>), dim3(blocks), dim3(256), 0, s, y, gamma, beta, eps, tgt, out,
                                      loss_sum, M, C));
    UNITE_LAUNCH_CHECK();
    return UNITE_OK;
}

extern "C" int unite_decoder_tail_bwd(const float* y, const float* gamma, const float* beta, float eps, const float* tgt,
                                      float loss_scale, const float* loss_scale_dev, const float* dout, void* dy_bf16, float* dgamma, float* dbeta,
                                      float* dysum, int32_t accumulate, void* workspace, int32_t M, int32_t C, void* stream) {
    if (!y || !gamma || !beta || !dy_bf16 || !workspace || (!tgt && !dout) || M <= 0 || !dim_ok(C)) return UNITE_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int nb = (M + bwd_rows_per_block(M) - 1) / bwd_rows_per_block(M);
    DISPATCH_NV(C, hipLaunchKernelGGL((decoder_tail_bwd_kernel<NV>), dim3(nb), dim3(256), 0, s, y, gamma, beta, eps, tgt, loss_scale,
                                      loss_scale_dev, dout, dy_bf16, (float*)workspace, M, C));
    UNITE_LAUNCH_CHECK();
    if (dgamma || dbeta || dysum) {
        hipLaunchKernelGGL(reduce_partials_kernel, dim3((C + 31) / 32, 3), dim3(256), 0, s, (const float*)workspace, nb, C, dgamma,
                           dbeta, dysum, accumulate);
        UNITE_LAUNCH_CHECK();
    }
    return UNITE_OK;
}

extern "C" int unite_clip_embed_ln(const void* patches, const float* class_embedding, const float* positional_embedding,
                                   const float* gamma, const float* beta, float eps, void* x, int32_t x_f32, int32_t BT, int32_t HW,
                                   int32_t D, void* stream) {
    if (!patches || !class_embedding || !positional_embedding || !gamma || !beta || !x || BT <= 0 || HW <= 0 || !dim_ok(D))
        return UNITE_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int rows = BT * (HW + 1);
    DISPATCH_NV(D, hipLaunchKernelGGL((clip_embed_ln_kernel<NV>), dim3((rows + 3) / 4), dim3(256), 0, s, (const uint16_t*)patches,
                                      class_embedding, positional_embedding, gamma, beta, eps, x, x_f32, BT, HW, D));
    UNITE_LAUNCH_CHECK();
    return UNITE_OK;
}

extern "C" int unite_l2_normalize_rows(float* x, int32_t M, int32_t D, void* stream) {
    if (!x || M <= 0 || !dim_ok(D)) return UNITE_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    DISPATCH_NV(D, hipLaunchKernelGGL((l2_normalize_kernel<NV>), dim3((M + 3) / 4), dim3(256), 0, s, x, M, D));
    UNITE_LAUNCH_CHECK();
    return UNITE_OK;
}
