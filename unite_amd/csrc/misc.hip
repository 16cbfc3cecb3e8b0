// Gather / index / small reduction kernels of the stage-1 step (all HBM- or latency-bound, no MFMA).
#include "common.h"
#include <string.h>

namespace {

// ------------------------------------------------------------------------------------ im2col of the listed tokens
// One wavefront per patch row of 3*P*P elements; each lane moves VEC pixels of one patch line (VEC = 4 for P % 4 == 0: a 64-B
// line of a 16-pixel patch is 4 lanes; VEC = 2 for P = 14).  Rows are ld_cols wide; columns [3*P*P, ld_cols) are written as zeros
// so the GEMM's K can be padded to a 16-byte multiple (588 -> 592 for CLIP-L/14).
template <int VEC>
__global__ __launch_bounds__(256) void im2col_gather_kernel(const float* __restrict__ video, const int32_t* __restrict__ token_index,
                                                            uint16_t* __restrict__ cols, int n_rows, int B, int T, int H, int W, int P,
                                                            int ld_cols) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const int GH = H / P, GW = W / P;
    const int tok = token_index ? token_index[row] : row;
    const int gw = tok % GW, gh = (tok / GW) % GH, t = (tok / (GW * GH)) % T, b = tok / (GW * GH * T);
    const int PP = P * P, KD = 3 * PP;
    const size_t plane = (size_t)H * W;
    uint16_t* dst = cols + (size_t)row * ld_cols;
    for (int e = lane * VEC; e < ld_cols; e += 64 * VEC) {
        if (e < KD) {
            const int c = e / PP, ph = (e % PP) / P, pw = e % P;
            const float* src = video + (((size_t)b * 3 + c) * T + t) * plane + (size_t)(gh * P + ph) * W + gw * P + pw;
            if (VEC == 4) {
                const f32x4 v = *(const f32x4*)src;
                *(u32x2*)(dst + e) = (u32x2){pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
            } else {
                const f32x2 v = *(const f32x2*)src;
                *(uint32_t*)(dst + e) = pack_bf16x2(v[0], v[1]);
            }
        } else {
            if (VEC == 4) *(u32x2*)(dst + e) = (u32x2){0u, 0u};
            else *(uint32_t*)(dst + e) = 0u;
        }
    }
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ table, const int32_t* __restrict__ index, int modulo,
                                                          float* __restrict__ out, int n_rows, int D) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    int src = index ? index[row] : row;
    if (modulo > 0) src %= modulo;
    for (int c = lane * 4; c < D; c += 256) *(f32x4*)(out + (size_t)row * D + c) = *(const f32x4*)(table + (size_t)src * D + c);
}

// ------------------------------------------------------------------------------------ column sums (bias gradients)
constexpr int CS_ROWS = 256;   // rows per workgroup
constexpr int CS_HEADER = 4096;  // arrival counters, one per 512-column block (zero outside a launch); N <= 512 * 1024
// block (bx, by): columns [bx*512, +512) as 64 lanes x 8, rows [by*256, +256) split over 4 waves -> partial[by][n]; the LAST row block of
// a column block to finish (arrive_last: nobody waits) adds the partial rows in order (bitwise reproducible) and writes the 512 sums
__global__ __launch_bounds__(256) void colsum_kernel(const uint16_t* __restrict__ x, int ldx, int M, int N, char* __restrict__ workspace,
                                                     float* __restrict__ out, int accumulate, int zero_lo, int zero_hi) {
    __shared__ float red[4][512];
    __shared__ uint32_t last_word;
    float* const partial = (float*)(workspace + CS_HEADER);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c0 = blockIdx.x * 512 + lane * 8;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (c0 < N) {
        const int r_end = min(M, (int)(blockIdx.y + 1) * CS_ROWS);
        for (int r = blockIdx.y * CS_ROWS + wave; r < r_end; r += 4) {
            const u32x4 w = *(const u32x4*)(x + (size_t)r * ldx + c0);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[2 * e] += __uint_as_float(w[e] << 16);
                acc[2 * e + 1] += __uint_as_float(w[e] & 0xFFFF0000u);
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[wave][lane * 8 + e] = acc[e];
    __syncthreads();
    for (int c = threadIdx.x; c < 512; c += 256) {
        const int gc = blockIdx.x * 512 + c;
        if (gc < N) partial[(size_t)blockIdx.y * N + gc] = red[0][c] + red[1][c] + red[2][c] + red[3][c];
    }
    const int nparts = (int)gridDim.y;
    if (!arrive_last((uint32_t*)workspace + blockIdx.x, (uint32_t)nparts, &last_word)) return;
    for (int c = threadIdx.x; c < 512; c += 256) {
        const int gc = blockIdx.x * 512 + c;
        if (gc >= N) continue;
        float a = 0.f;
        int i = 0;
        for (; i + 8 <= nparts; i += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = partial[(size_t)(i + u) * N + gc];
#pragma unroll
            for (int u = 0; u < 8; ++u) a += v[u];
        }
        for (; i < nparts; ++i) a += partial[(size_t)i * N + gc];
        if (gc >= zero_lo && gc < zero_hi) a = 0.f;
        out[gc] = accumulate ? out[gc] + a : a;
    }
}

// ------------------------------------------------------------------------------------ crop + Pillow-bilinear resize of uint8 frames
// The training transform of the reference crops every frame of a clip with ONE box and resizes the crop with Pillow's
// img.resize((w, h), Image.BILINEAR) (src/datasets/transforms.py:136-152, build.py:37).  Pillow's 8-bit resample
// (libImaging/Resample.c, pillow==10.0.1 in the reference's environment) is a separable triangle filter whose support grows with the
// down-scaling factor, weights normalised in double and rounded to 22-bit fixed point, a horizontal pass rounded to uint8, then a
// vertical pass rounded to uint8.  The same arithmetic here, bit for bit (oracle/pil_resize.py; tests compare with Pillow itself):
//   coefficient kernel: per clip and axis, the tap range and fixed-point weights of every output position (double arithmetic);
//   horizontal pass:    frames (B,T,H,W,3) -> tmp (B,T,H,OW,3) on the rows of the crop;   vertical pass: tmp -> out (B,T,OH,OW,3).
constexpr int CR_KMAX = 16;            // taps per output position: 2 ceil(scale) + 1 <= 16, i.e. crops up to 7.5 x the output side
constexpr int CR_MAXB = 64;            // clips per launch (the boxes travel as kernel arguments)
constexpr int CR_PREC = 32 - 8 - 2;
struct CropBoxes { int32_t x0[CR_MAXB], y0[CR_MAXB], w[CR_MAXB], h[CR_MAXB]; };

__device__ __forceinline__ double cr_tri(double x) {
    x = x < 0.0 ? -x : x;
    return x < 1.0 ? 1.0 - x : 0.0;
}

// grid (B, 2): axis 0 = x (in_size = box w, out OW), axis 1 = y.  bounds int32 [B][2][OMAX][2], kk int32 [B][2][OMAX][CR_KMAX]
__global__ __launch_bounds__(256) void crop_resize_coeffs_kernel(const CropBoxes boxes, int OH, int OW, int OMAX, int32_t* __restrict__ bounds,
                                                                 int32_t* __restrict__ kk) {
    const int b = blockIdx.x, axis = blockIdx.y;
    const int in_size = axis == 0 ? boxes.w[b] : boxes.h[b], out_size = axis == 0 ? OW : OH;
    const double scale = (double)in_size / (double)out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 1.0 * filterscale, ss = 1.0 / filterscale;
    for (int xx = threadIdx.x; xx < out_size; xx += blockDim.x) {
        const double center = (xx + 0.5) * scale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        double w[CR_KMAX];
        double ww = 0.0;
#pragma unroll
        for (int x = 0; x < CR_KMAX; ++x) {
            w[x] = x < xmax ? cr_tri((x + xmin - center + 0.5) * ss) : 0.0;
            if (x < xmax) ww += w[x];
        }
        int32_t* kp = kk + (((size_t)b * 2 + axis) * OMAX + xx) * CR_KMAX;
#pragma unroll
        for (int x = 0; x < CR_KMAX; ++x) {
            double v = (x < xmax && ww != 0.0) ? w[x] / ww : w[x];
            kp[x] = x < xmax ? (v < 0.0 ? (int32_t)(-0.5 + v * (double)(1 << CR_PREC)) : (int32_t)(0.5 + v * (double)(1 << CR_PREC))) : 0;
        }
        int32_t* bp = bounds + (((size_t)b * 2 + axis) * OMAX + xx) * 2;
        bp[0] = xmin;
        bp[1] = xmax;
    }
}

__device__ __forceinline__ uint8_t cr_clip8(int32_t acc) {
    const int32_t v = acc >> CR_PREC;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// one thread per output byte (ox, c) of a row: grid (ceil(OW*3 / 256), H rows of the crop at most, B*T)
__global__ __launch_bounds__(256) void crop_resize_h_kernel(const uint8_t* __restrict__ frames, const CropBoxes boxes, uint8_t* __restrict__ tmp,
                                                            const int32_t* __restrict__ bounds, const int32_t* __restrict__ kk, int T, int H,
                                                            int W, int OW, int OMAX) {
    const int bt = blockIdx.z, b = bt / T, row = blockIdx.y;
    if (row >= boxes.h[b]) return;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= OW * 3) return;
    const int ox = i / 3, c = i - ox * 3;
    const int32_t* bp = bounds + (((size_t)b * 2 + 0) * OMAX + ox) * 2;
    const int32_t* kp = kk + (((size_t)b * 2 + 0) * OMAX + ox) * CR_KMAX;
    const int xmin = bp[0], n = bp[1];
    const uint8_t* src = frames + (((size_t)bt * H + boxes.y0[b] + row) * W + boxes.x0[b] + xmin) * 3 + c;
    int32_t acc = 1 << (CR_PREC - 1);
    for (int x = 0; x < n; ++x) acc += (int32_t)src[x * 3] * kp[x];
    tmp[((size_t)bt * H + row) * OW * 3 + i] = boxes.w[b] == OW ? src[(ox - xmin) * 3] : cr_clip8(acc);      // equal size: Pillow copies
}

__global__ __launch_bounds__(256) void crop_resize_v_kernel(const uint8_t* __restrict__ tmp, const CropBoxes boxes, uint8_t* __restrict__ out,
                                                            const int32_t* __restrict__ bounds, const int32_t* __restrict__ kk, int T, int H,
                                                            int OH, int OW, int OMAX) {
    const int bt = blockIdx.z, b = bt / T, oy = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= OW * 3) return;
    const int32_t* bp = bounds + (((size_t)b * 2 + 1) * OMAX + oy) * 2;
    const int32_t* kp = kk + (((size_t)b * 2 + 1) * OMAX + oy) * CR_KMAX;
    const int ymin = bp[0], n = bp[1];
    const uint8_t* src = tmp + ((size_t)bt * H + ymin) * OW * 3 + i;
    int32_t acc = 1 << (CR_PREC - 1);
    for (int y = 0; y < n; ++y) acc += (int32_t)src[(size_t)y * OW * 3] * kp[y];
    out[((size_t)bt * OH + oy) * OW * 3 + i] = boxes.h[b] == OH ? src[(size_t)(oy - ymin) * OW * 3] : cr_clip8(acc);
}

// ------------------------------------------------------------------------------------ mask sampling
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// one workgroup per frame row, one thread per position: 256 threads for N <= 256 (224 @ 16, 196 @ 14), 1024 for N <= 1024 (CLIP-L/14 @ 336: 576
// patches).  keys in LDS; rank by counting; ascending compaction by scan.
constexpr int MASK_NMAX = 1024;
inline int mask_threads(int N) { return N <= 256 ? 256 : 1024; }
__global__ __launch_bounds__(1024) void mask_sample_kernel(const float* __restrict__ weights, uint64_t seed, const uint64_t* __restrict__ seed_dev,
                                                          const int64_t* __restrict__ importance,
                                                          const uint8_t* __restrict__ mask_in, uint8_t* __restrict__ mask,
                                                          int32_t* __restrict__ vis_tokens, int32_t* __restrict__ vis_rows_cls, int BT, int N,
                                                          int n_vis) {
    __shared__ float key[MASK_NMAX];
    __shared__ int flag[MASK_NMAX];
    const int bt = blockIdx.x, j = threadIdx.x;
    int visible = 0;
    if (mask_in) {
        visible = (j < N) ? (mask_in[(size_t)bt * N + j] == 0) : 0;
    } else if (importance) {
        flag[j] = 0;
        __syncthreads();
        if (j < n_vis) flag[(int)importance[(size_t)bt * N + j]] = 1;
        __syncthreads();
        visible = (j < N) ? flag[j] : 0;
    } else {
        float k = INFINITY;
        if (j < N) {
            const float w = weights[(size_t)bt * N + j];
            const uint64_t sd = seed_dev ? *seed_dev : seed;      // device-resident seed: the launch can be replayed from a HIP graph
            const uint64_t h = splitmix64(sd ^ splitmix64(((uint64_t)bt << 20) + (uint64_t)j + 1));
            const float u = ((float)(h >> 40) + 0.5f) * (1.0f / 16777216.0f);     // (0,1)
            k = (w > 0.f) ? -__logf(u) / w : INFINITY;                             // exponential race
        }
        key[j] = k;
        __syncthreads();
        if (j < N) {
            int rank = 0;
            for (int i = 0; i < N; ++i) {
                const float ki = key[i];
                rank += (ki < k) || (ki == k && i < j);
            }
            visible = rank < n_vis;
        }
    }
    if (mask && j < N) mask[(size_t)bt * N + j] = visible ? 0 : 1;
    // ascending compaction of the visible positions
    __syncthreads();
    flag[j] = visible;
    __syncthreads();
    if (visible) {
        int pos = 0;
        for (int i = 0; i < j; ++i) pos += flag[i];
        if (pos < n_vis) {
            vis_tokens[(size_t)bt * n_vis + pos] = bt * N + j;
            if (vis_rows_cls) vis_rows_cls[(size_t)bt * n_vis + pos] = bt * (N + 1) + 1 + j;
        }
    }
}

// ------------------------------------------------------------------------------------ stochastic depth keep-vectors
// timm drop_path as the blocks use it (reference modeling_finetune.py:42-53): per residual branch and sample the multiplier
// floor(keep + U[0,1)) / keep.  out[l * per_layer + i] for layer l with keep probability keep[l]; counter-based uniforms
// (splitmix64 of seed and element index), so no generator state lives on the device and nothing syncs with the host.
__global__ __launch_bounds__(256) void drop_path_scales_kernel(const float* __restrict__ keep, uint64_t seed, const uint64_t* __restrict__ seed_dev,
                                                               float* __restrict__ out, int per_layer, int total) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const float k = keep[i / per_layer];
    const uint64_t sd = seed_dev ? *seed_dev : seed;
    const uint64_t h = splitmix64(sd ^ splitmix64(0xD809A7ull + (uint64_t)i));
    const float u = (float)(h >> 40) * (1.0f / 16777216.0f);      // [0,1)
    out[i] = floorf(k + u) / k;
}

// ------------------------------------------------------------------------------------ stage 3: greedy committee masks
// utils.get_greedy_masks (reference src/utils.py:89-120): per frame sort the attention descending; committee member i keeps
// ranks i, i+k, i+2k, ... (its first n_vis of them).  One workgroup per frame, rank by counting (ties: lower index first).
__global__ __launch_bounds__(1024) void greedy_masks_kernel(const float* __restrict__ weights, int k, uint8_t* __restrict__ mask,
                                                           int32_t* __restrict__ vis_tokens, int32_t* __restrict__ vis_rows_cls, int BT, int N,
                                                           int n_vis) {
    __shared__ float key[MASK_NMAX];
    __shared__ int member[MASK_NMAX];
    const int bt = blockIdx.x, j = threadIdx.x;
    const float w = (j < N) ? weights[(size_t)bt * N + j] : -INFINITY;
    key[j] = w;
    __syncthreads();
    int mem = -1;
    if (j < N) {
        int rank = 0;
        for (int i = 0; i < N; ++i) {
            const float wi = key[i];
            rank += (wi > w) || (wi == w && i < j);
        }
        if (rank / k < n_vis) mem = rank % k;
    }
    member[j] = mem;
    __syncthreads();
    if (j < N) {
        for (int i = 0; i < k; ++i) mask[((size_t)i * BT + bt) * N + j] = (mem == i) ? 0 : 1;
        if (mem >= 0) {
            int pos = 0;
            for (int i = 0; i < j; ++i) pos += (member[i] == mem);
            const size_t o = ((size_t)mem * BT + bt) * n_vis + pos;
            vis_tokens[o] = bt * N + j;                       // token id inside this member's own copy of the B target clips
            if (vis_rows_cls) vis_rows_cls[o] = bt * (N + 1) + 1 + j;
        }
    }
}

// ------------------------------------------------------------------------------------ stage 3: pseudo-label selection
// run_stage3.py:488-613 on device, one thread per target clip.  weight[b] = sel_b * (conf_weighted ? msp_b : 1), label = student
// prediction on the full clip; the engine's target loss is then  ratio / B_t * sum_b weight_b * CE(masked logits_b, label_b)
// (= ratio * sel_ratio * mean over the selected clips, :599-613) with no host sync on the selection count.
enum { SEL_CONF = 0, SEL_CONS = 1, SEL_CONS_OR_CONF = 2, SEL_CONS_AND_CONF = 3, SEL_CLIP_ONLY = 4, SEL_CLIP_MATCH_OR_CONF = 5, SEL_ORACLE = 6 };
__global__ __launch_bounds__(256) void pseudo_label_kernel(const float* __restrict__ logits_full, const float* __restrict__ logits_masked, int k,
                                                           const float* __restrict__ clip_probs, const int64_t* __restrict__ labels_t,
                                                           int strategy, float threshold, float clip_threshold, int conf_weighted,
                                                           int64_t* __restrict__ pseudo, float* __restrict__ weight, uint8_t* __restrict__ sel_out,
                                                           float* __restrict__ msp_out, int B, int C) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    const float* lf = logits_full + (size_t)b * C;
    float m = -INFINITY;
    int pred = 0;
    for (int c = 0; c < C; ++c)
        if (lf[c] > m) { m = lf[c]; pred = c; }
    float ssum = 0.f;
    for (int c = 0; c < C; ++c) ssum += __expf(lf[c] - m);
    const float msp = 1.0f / ssum;                                 // max softmax probability (:489-490)
    int votes = 0;
    for (int i = 0; i < k; ++i) {
        const float* lm = logits_masked + ((size_t)i * B + b) * C;
        float mm = -INFINITY;
        int pm = 0;
        for (int c = 0; c < C; ++c)
            if (lm[c] > mm) { mm = lm[c]; pm = c; }
        votes += (pm == pred);
    }
    const bool cons = votes >= k;                                  // :510-520
    const bool conf = msp >= threshold;                            // :522-523 (global_threshold is overwritten to 0.5 by the caller)
    bool sel = false;
    if (strategy == SEL_CONF) sel = conf;
    else if (strategy == SEL_CONS) sel = cons;
    else if (strategy == SEL_CONS_OR_CONF) sel = cons || conf;
    else if (strategy == SEL_CONS_AND_CONF) sel = cons && conf;
    else if (strategy == SEL_ORACLE) sel = labels_t && labels_t[b] == pred;
    else {
        float cm = -INFINITY;
        int cp = 0;
        for (int c = 0; c < C; ++c) {
            const float v = clip_probs[(size_t)b * C + c];
            if (v > cm) { cm = v; cp = c; }
        }
        if (strategy == SEL_CLIP_ONLY) sel = cm >= threshold;      // :551-554
        else {                                                      // clip_matchORconf, :556-572
            const bool match = cp == pred;
            const bool sconf = msp >= clip_threshold, cconf = cm >= clip_threshold;
            sel = match || ((sconf != cconf) && !match);
        }
    }
    pseudo[b] = pred;                                              // :575-576: the student's prediction is the label
    weight[b] = sel ? (conf_weighted ? msp : 1.0f) : 0.0f;
    if (sel_out) sel_out[b] = sel ? 1 : 0;
    if (msp_out) msp_out[b] = msp;
}

// ------------------------------------------------------------------------------------ CLS-row attention probabilities
// one workgroup per frame (bt); thread j <-> key j; loops over heads.  qkv packed [B*N, 3*H*64].
constexpr int CLS_KPT = 4;      // keys per thread: N <= 1024 (257 = 224 @ patch 14 + CLS, 577 = 336 @ 14)
__global__ __launch_bounds__(256) void attn_cls_probs_kernel(const uint16_t* __restrict__ qkv, float* __restrict__ probs, int N, int H,
                                                             float scale) {
    __shared__ float q[64];
    __shared__ float red[4];
    const int bt = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ld = 3 * H * 64;
    float accp[CLS_KPT] = {0.f, 0.f, 0.f, 0.f};
    for (int h = 0; h < H; ++h) {
        __syncthreads();
        if (tid < 64) q[tid] = bf16_to_f32(qkv[(size_t)bt * N * ld + h * 64 + tid]) * scale;
        __syncthreads();
        float s[CLS_KPT];
        float m = -INFINITY;
#pragma unroll
        for (int i = 0; i < CLS_KPT; ++i) {
            const int j = tid + 256 * i;
            s[i] = -INFINITY;
            if (j < N) {
                const uint16_t* kp = qkv + ((size_t)bt * N + j) * ld + H * 64 + h * 64;
                float d = 0.f;
#pragma unroll
                for (int c = 0; c < 64; c += 8) {
                    const u32x4 w = *(const u32x4*)(kp + c);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        d += q[c + 2 * e] * __uint_as_float(w[e] << 16);
                        d += q[c + 2 * e + 1] * __uint_as_float(w[e] & 0xFFFF0000u);
                    }
                }
                s[i] = d;
            }
            m = fmaxf(m, s[i]);
        }
        m = wave_max(m);
        if (lane == 0) red[wave] = m;
        __syncthreads();
        m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        __syncthreads();
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < CLS_KPT; ++i) {
            s[i] = (tid + 256 * i < N) ? __expf(s[i] - m) : 0.f;
            t += s[i];
        }
        t = wave_sum(t);
        if (lane == 0) red[wave] = t;
        __syncthreads();
        t = red[0] + red[1] + red[2] + red[3];
#pragma unroll
        for (int i = 0; i < CLS_KPT; ++i) accp[i] += s[i] / t;
    }
#pragma unroll
    for (int i = 0; i < CLS_KPT; ++i) {
        const int j = tid + 256 * i;
        if (j >= 1 && j < N) probs[(size_t)bt * (N - 1) + (j - 1)] = accp[i] / (float)H;
    }
}

// ------------------------------------------------------------------------------------ bicubic plane resize (teacher input)
// torch.nn.functional.interpolate(mode='bicubic', align_corners=False) semantics: source x = (ox + 0.5) * W/OW - 0.5, the four
// taps floor(x)-1 .. floor(x)+2 clamped to the plane, Keys cubic with A = -0.75, f32 arithmetic.  One thread per output pixel;
// a wave covers 64 consecutive pixels of a row, so the four source rows are read as contiguous spans.
__device__ __forceinline__ void cubic_coeffs(float t, float c[4]) {
    const float A = -0.75f;
    const float x0 = t + 1.0f, x3 = 2.0f - t, x2 = 1.0f - t;
    c[0] = ((A * x0 - 5.0f * A) * x0 + 8.0f * A) * x0 - 4.0f * A;
    c[1] = ((A + 2.0f) * t - (A + 3.0f)) * t * t + 1.0f;
    c[2] = ((A + 2.0f) * x2 - (A + 3.0f)) * x2 * x2 + 1.0f;
    c[3] = ((A * x3 - 5.0f * A) * x3 + 8.0f * A) * x3 - 4.0f * A;
}

__global__ __launch_bounds__(256) void resize_bicubic_kernel(const float* __restrict__ src, float* __restrict__ dst, int H, int W, int OH,
                                                             int OW, float sh, float sw) {
    const int ox = blockIdx.x * 256 + threadIdx.x, oy = blockIdx.y;
    if (ox >= OW) return;
    const float* sp = src + (size_t)blockIdx.z * H * W;
    const float ry = sh * ((float)oy + 0.5f) - 0.5f, rx = sw * ((float)ox + 0.5f) - 0.5f;
    const float fy = floorf(ry), fx = floorf(rx);
    float cy[4], cx[4];
    cubic_coeffs(ry - fy, cy);
    cubic_coeffs(rx - fx, cx);
    const int iy = (int)fy, ix = (int)fx;
    float acc = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int y = min(max(iy - 1 + a, 0), H - 1);
        float r = 0.f;
#pragma unroll
        for (int b = 0; b < 4; ++b) r += cx[b] * sp[(size_t)y * W + min(max(ix - 1 + b, 0), W - 1)];
        acc += cy[a] * r;
    }
    dst[((size_t)blockIdx.z * OH + oy) * OW + ox] = acc;
}

// ------------------------------------------------------------------------------------ clip frames -> training tensor
// uint8 (B,T,H,W,3) decoded frames -> f32 (B,3,T,H,W): ((x / 255) - mean[c]) / std[c], optional left-right flip per clip.
// The arithmetic order (two correctly rounded f32 divisions, one subtraction) is the reference's ToTorchFormatTensor +
// GroupNormalize (src/datasets/transforms.py:93-94,245), so the result is bit-identical to its CPU pipeline.  One thread = 4
// pixels of a row: 12 contiguous input bytes (a wave reads 768 B), one 16-B store per channel plane.
__global__ __launch_bounds__(256) void clip_u8_to_f32_kernel(const uint8_t* __restrict__ frames, float* __restrict__ out,
                                                             const uint8_t* __restrict__ flip, float m0, float m1, float m2, float s0, float s1,
                                                             float s2, int T, int H, int W, size_t n_quads) {
    const int W4 = W >> 2;
    const size_t plane = (size_t)H * W;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_quads; i += (size_t)gridDim.x * 256) {
        const int xq = (int)(i % W4);
        const size_t row = i / W4;                      // ((b * T + t) * H + y)
        const int y = (int)(row % H);
        const size_t bt = row / H;
        const int t = (int)(bt % T), b = (int)(bt / T);
        const bool fl = flip && flip[b];
        const int x0 = fl ? W - 4 - 4 * xq : 4 * xq;    // source quad; its pixels are reversed below when flipped
        const uint8_t* src = frames + (row * W + x0) * 3;
        uint32_t w[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) w[k] = ((const uint32_t*)src)[k];        // 12 bytes; (row * W + x0) * 3 is a multiple of 4
        float px[4][3];
#pragma unroll
        for (int e = 0; e < 12; ++e) px[e / 3][e % 3] = (float)((w[e >> 2] >> (8 * (e & 3))) & 0xFFu);
        const float mean[3] = {m0, m1, m2}, sd[3] = {s0, s1, s2};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            f32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = (px[fl ? 3 - j : j][c] / 255.0f - mean[c]) / sd[c];
            *(f32x4*)(out + (((size_t)b * 3 + c) * T + t) * plane + (size_t)y * W + 4 * xq) = v;
        }
    }
}

// ------------------------------------------------------------------------------------ zero-shot similarities (utils.py:55-68)
// out[b, c] = mean_t softmax_c( scale * <img[b*T + t, :], text[c, :]> ).  One workgroup per clip; thread c owns class c (n_cls <= 256).
__global__ __launch_bounds__(256) void clip_similarity_kernel(const float* __restrict__ img, const float* __restrict__ text, float* __restrict__ out,
                                                              int T, int C, int n_cls, float scale) {
    __shared__ float red[4];
    const int b = blockIdx.x, c = threadIdx.x, lane = c & 63, wave = c >> 6;
    float acc = 0.f;
    for (int t = 0; t < T; ++t) {
        const float* f = img + (size_t)(b * T + t) * C;
        float logit = -INFINITY;
        if (c < n_cls) {
            const float* w = text + (size_t)c * C;
            float d = 0.f;
            for (int k = 0; k < C; k += 4) {
                const f32x4 a = *(const f32x4*)(f + k), bb = *(const f32x4*)(w + k);
                d += a[0] * bb[0] + a[1] * bb[1] + a[2] * bb[2] + a[3] * bb[3];
            }
            logit = scale * d;
        }
        float m = wave_max(logit);
        if (lane == 0) red[wave] = m;
        __syncthreads();
        m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        __syncthreads();
        const float e = c < n_cls ? __expf(logit - m) : 0.f;
        float ssum = wave_sum(e);
        if (lane == 0) red[wave] = ssum;
        __syncthreads();
        ssum = red[0] + red[1] + red[2] + red[3];
        __syncthreads();
        acc += e / ssum;
    }
    if (c < n_cls) out[(size_t)b * n_cls + c] = acc / (float)T;
}

// ------------------------------------------------------------------------------------ token mean
// one workgroup per (clip, 64 columns): thread (g, q) = (tid >> 4, tid & 15) sums tokens g, g + 16, ... of columns 4q .. 4q+3
// (16-byte loads, 256 B per 16 lanes), then the 16 token groups are added in a fixed order through LDS.  (The first version
// walked all N tokens in one thread per column: 1.1 ms for 16 x 3136 x 768.)
__global__ __launch_bounds__(256) void token_mean_fwd_kernel(const float* __restrict__ x, float* __restrict__ out, int N, int D) {
    __shared__ f32x4 red[16][16];
    const int b = blockIdx.y, q = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int c = blockIdx.x * 64 + 4 * q;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    if (c < D) {
        const float* p = x + (size_t)b * N * D + c;
        int n = g;
        for (; n + 48 < N; n += 64) {
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *(const f32x4*)(p + (size_t)(n + 16 * u) * D);
#pragma unroll
            for (int u = 0; u < 4; ++u) a = (f32x4){a[0] + v[u][0], a[1] + v[u][1], a[2] + v[u][2], a[3] + v[u][3]};
        }
        for (; n < N; n += 16) {
            const f32x4 v = *(const f32x4*)(p + (size_t)n * D);
            a = (f32x4){a[0] + v[0], a[1] + v[1], a[2] + v[2], a[3] + v[3]};
        }
    }
    red[g][q] = a;
    __syncthreads();
    if (g == 0 && c < D) {
        f32x4 t = red[0][q];
#pragma unroll
        for (int i = 1; i < 16; ++i) t = (f32x4){t[0] + red[i][q][0], t[1] + red[i][q][1], t[2] + red[i][q][2], t[3] + red[i][q][3]};
        const float inv = 1.0f / (float)N;
        *(f32x4*)(out + (size_t)b * D + c) = (f32x4){t[0] * inv, t[1] * inv, t[2] * inv, t[3] * inv};
    }
}
// ------------------------------------------------------------------------------------ small fp32 linear (classifier head)
// y[b,c] = <x[b,:], W[c,:]> + bias[c]   -- B <= a few dozen rows, C = number of classes: one workgroup per row
__global__ __launch_bounds__(256) void linear_f32_fwd_kernel(const float* __restrict__ x, const float* __restrict__ W, const float* __restrict__ bias,
                                                             float* __restrict__ y, int C, int D) {
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int c = wave; c < C; c += 4) {
        float a = 0.f;
        for (int d = lane * 4; d < D; d += 256) {
            const f32x4 xv = *(const f32x4*)(x + (size_t)b * D + d), wv = *(const f32x4*)(W + (size_t)c * D + d);
            a += xv[0] * wv[0] + xv[1] * wv[1] + xv[2] * wv[2] + xv[3] * wv[3];
        }
        a = wave_sum(a);
        if (lane == 0) y[(size_t)b * C + c] = a + (bias ? bias[c] : 0.f);
    }
}
// dx[b,d] = sum_c dy[b,c] W[c,d]
__global__ __launch_bounds__(256) void linear_f32_dx_kernel(const float* __restrict__ dy, const float* __restrict__ W, float* __restrict__ dx, int C,
                                                            int D) {
    const int b = blockIdx.x;
    for (int d = threadIdx.x; d < D; d += 256) {
        float a = 0.f;
        for (int c = 0; c < C; ++c) a += dy[(size_t)b * C + c] * W[(size_t)c * D + d];
        dx[(size_t)b * D + d] = a;
    }
}
// dW[c,d] (+)= sum_b dy[b,c] x[b,d];  db[c] (+)= sum_b dy[b,c]
__global__ __launch_bounds__(256) void linear_f32_dw_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dW,
                                                            float* __restrict__ db, int B, int C, int D, int accumulate) {
    const int c = blockIdx.x;
    for (int d = threadIdx.x; d < D; d += 256) {
        float a = 0.f;
        for (int b = 0; b < B; ++b) a += dy[(size_t)b * C + c] * x[(size_t)b * D + d];
        dW[(size_t)c * D + d] = accumulate ? dW[(size_t)c * D + d] + a : a;
    }
    if (db && threadIdx.x == 0) {
        float a = 0.f;
        for (int b = 0; b < B; ++b) a += dy[(size_t)b * C + c];
        db[c] = accumulate ? db[c] + a : a;
    }
}

__global__ __launch_bounds__(256) void token_mean_bwd_kernel(const float* __restrict__ dout, float* __restrict__ dx, int accumulate, int N, int D,
                                                             size_t total4) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total4) return;
    const size_t e = i * 4;
    const int c = (int)(e % D);
    const size_t b = e / ((size_t)N * D);
    f32x4 g = *(const f32x4*)(dout + b * D + c);
    const float inv = 1.0f / (float)N;
    g = (f32x4){g[0] * inv, g[1] * inv, g[2] * inv, g[3] * inv};
    if (accumulate) {
        const f32x4 o = *(const f32x4*)(dx + e);
        g = (f32x4){g[0] + o[0], g[1] + o[1], g[2] + o[2], g[3] + o[3]};
    }
    *(f32x4*)(dx + e) = g;
}

// ------------------------------------------------------------------------------------ softmax cross-entropy (one wave per row)
__global__ __launch_bounds__(256) void softmax_ce_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels,
                                                         const float* __restrict__ row_weight, float grad_scale, float* __restrict__ loss_sum,
                                                         float* __restrict__ dlogits, int M, int C) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int64_t lab = labels[row];
    const float w = row_weight ? row_weight[row] : 1.0f;
    float m = -INFINITY;
    for (int c = lane; c < C; c += 64) m = fmaxf(m, logits[(size_t)row * C + c]);
    m = wave_max(m);
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += __expf(logits[(size_t)row * C + c] - m);
    s = wave_sum(s);
    const bool valid = lab >= 0 && lab < C;
    if (dlogits) {
        for (int c = lane; c < C; c += 64) {
            const float p = __expf(logits[(size_t)row * C + c] - m) / s;
            dlogits[(size_t)row * C + c] = valid ? grad_scale * w * (p - (c == lab ? 1.f : 0.f)) : 0.f;
        }
    }
    if (loss_sum && valid && lane == 0) atomicAdd(loss_sum, w * (logf(s) + m - logits[(size_t)row * C + lab]));
}

// y[m,:] = bf16(row_scale[m / rows_per_scale] * x[m,:])
__global__ __launch_bounds__(256) void scale_cast_bf16_kernel(const float* __restrict__ x, const float* __restrict__ row_scale, int rows_per_scale,
                                                              uint16_t* __restrict__ y, int D, size_t total4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (size_t)gridDim.x * 256) {
        const size_t e = i * 4;
        const float sc = row_scale ? row_scale[(e / D) / rows_per_scale] : 1.0f;
        const f32x4 v = *(const f32x4*)(x + e);
        *(u32x2*)(y + e) = (u32x2){pack_bf16x2(v[0] * sc, v[1] * sc), pack_bf16x2(v[2] * sc, v[3] * sc)};
    }
}

__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * 256 * 8;
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8; i < n; i += stride) {
        if (i + 8 <= n) {
            const f32x4 a = *(const f32x4*)(src + i), b = *(const f32x4*)(src + i + 4);
            *(u32x4*)(dst + i) = (u32x4){pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3]), pack_bf16x2(b[0], b[1]), pack_bf16x2(b[2], b[3])};
        } else {
            for (int64_t k = i; k < n; ++k) dst[k] = f32_to_bf16(src[k]);
        }
    }
}

}  // namespace

extern "C" int unite_im2col_gather(const float* video, const int32_t* token_index, void* cols, int32_t ld_cols, int32_t n_rows, int32_t B,
                                   int32_t T, int32_t H, int32_t W, int32_t P, void* stream) {
    if (!video || !cols || n_rows <= 0 || P <= 0 || (P & 1) || H % P || W % P || (W & 1) || ld_cols < 3 * P * P || (ld_cols & 7))
        return UNITE_EINVAL;
    const dim3 grid((n_rows + 3) / 4), block(256);
    if ((P & 3) == 0 && (W & 3) == 0)
        hipLaunchKernelGGL(im2col_gather_kernel<4>, grid, block, 0, (hipStream_t)stream, video, token_index, (uint16_t*)cols, n_rows, B, T, H,
                           W, P, ld_cols);
    else
        hipLaunchKernelGGL(im2col_gather_kernel<2>, grid, block, 0, (hipStream_t)stream, video, token_index, (uint16_t*)cols, n_rows, B, T, H,
                           W, P, ld_cols);
    UNITE_LAUNCH_CHECK();
    return UNITE_OK;
}

extern "C" int unite_gather_rows_bf16(const void* table, const int32_t* index, void* out, int32_t n_rows, int32_t D, void* stream) {
    if (!table || !index || !out || n_rows <= 0 || D <= 0 || (D & 7)) return UNITE_EINVAL;
    // a bf16 row of D elements is a row of D/2 dwords: the f32 kernel moves 16 bytes per lane either way
    hipLaunchKernelGGL(gather_rows_kernel, dim3((n_rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const float*)table, index, 0, (float*)out,
                       n_rows, D / 2);
    UNITE_LAUNCH_CHECK();
    return UNITE_OK;
}

extern "C" int unite_gather_rows_f32(const float* table, const int32_t* index, int32_t modulo, float* out, int32_t n_rows, int32_t D,
                                     void* stream) {
    if (!table || !out || n_rows <= 0 || D <= 0 || (D & 3)) return UNITE_EINVAL;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((n_rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, table, index, modulo, out, n_rows, D);
    UNITE_LAUNCH_CHECK();
    return UNITE_OK;
}

extern "C" size_t unite_colsum_workspace(int32_t M, int32_t N) {
    return CS_HEADER + (size_t)((M + CS_ROWS - 1) / CS_ROWS) * (size_t)N * sizeof(float);
}

extern "C" int unite_colsum_bf16(const void* x, int32_t ldx, int32_t M, int32_t N, float* out, int32_t accumulate, int32_t zero_lo,
                                 int32_t zero_hi, void* workspace, void* stream) {
    if (!x || !out || !workspace || M <= 0 || N <= 0 || (N & 7) || (ldx & 7) || (N + 511) / 512 > CS_HEADER / 4) return UNITE_EINVAL;
    const int nparts = (M + CS_ROWS - 1) / CS_ROWS;
    if (nparts > 65535) return UNITE_ENOSUP;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(colsum_kernel, dim3((N + 511) / 512, nparts), dim3(256), 0, s, (const uint16_t*)x, ldx, M, N, (char*)workspace, out,
                       accumulate, zero_lo, zero_hi);
    UNITE_LAUNCH_CHECK();
    return UNITE_OK;
}

static inline size_t cr_align(size_t v) { return (v + 255) & ~(size_t)255; }
extern "C" size_t unite_crop_resize_workspace(int32_t B, int32_t T, int32_t H, int32_t OH, int32_t OW) {
    const size_t omax = (size_t)(OH > OW ? OH : OW), nb = (size_t)(B < CR_MAXB ? B : CR_MAXB);
    return cr_align(nb * 2 * omax * 2 * 4) + cr_align(nb * 2 * omax * CR_KMAX * 4) + cr_align(nb * (size_t)T * H * OW * 3);
}

extern "C" int unite_crop_resize_u8(const uint8_t* frames, const int32_t* boxes_host, uint8_t* out, int32_t B, int32_t T, int32_t H, int32_t W,
                                    int32_t OH, int32_t OW, void* workspace, void* stream) {
    if (!frames || !boxes_host || !out || !workspace || B <= 0 || T <= 0 || H <= 0 || W <= 0 || OH <= 0 || OW <= 0) return UNITE_EINVAL;
    if ((size_t)B * T > 65535 * 64 || H > 65535 || OH > 65535) return UNITE_ENOSUP;
    const int omax = OH > OW ? OH : OW;
    hipStream_t s = (hipStream_t)stream;
    for (int b0 = 0; b0 < B; b0 += CR_MAXB) {
        const int nb = B - b0 < CR_MAXB ? B - b0 : CR_MAXB;
        CropBoxes bx;
        memset(&bx, 0, sizeof(bx));
        int hmax = 0;
        for (int i = 0; i < nb; ++i) {
            const int32_t* q = boxes_host + (size_t)(b0 + i) * 4;
            if (q[0] < 0 || q[1] < 0 || q[2] <= 0 || q[3] <= 0 || q[0] + q[2] > W || q[1] + q[3] > H) return UNITE_EINVAL;
            // taps per output position: 2 ceil(max(in / out, 1)) + 1 must fit CR_KMAX
            if (2 * ((q[2] + OW - 1) / OW > 1 ? (q[2] + OW - 1) / OW : 1) + 1 > CR_KMAX || 2 * ((q[3] + OH - 1) / OH > 1 ? (q[3] + OH - 1) / OH : 1) + 1 > CR_KMAX)
                return UNITE_ENOSUP;
            bx.x0[i] = q[0]; bx.y0[i] = q[1]; bx.w[i] = q[2]; bx.h[i] = q[3];
            hmax = q[3] > hmax ? q[3] : hmax;
        }
        const size_t nbw = (size_t)(B < CR_MAXB ? B : CR_MAXB);       // the layout unite_crop_resize_workspace sized
        char* ws = (char*)workspace;
        int32_t* bounds = (int32_t*)ws;
        int32_t* kk = (int32_t*)(ws + cr_align(nbw * 2 * omax * 2 * 4));
        uint8_t* tmp = (uint8_t*)((char*)kk + cr_align(nbw * 2 * omax * CR_KMAX * 4));
        const uint8_t* fr = frames + (size_t)b0 * T * H * W * 3;
        uint8_t* op = out + (size_t)b0 * T * OH * OW * 3;
        hipLaunchKernelGGL(crop_resize_coeffs_kernel, dim3(nb, 2), dim3(256), 0, s, bx, OH, OW, omax, bounds, kk);
        UNITE_LAUNCH_CHECK();
        const unsigned gx = (unsigned)((OW * 3 + 255) / 256);
        hipLaunchKernelGGL(crop_resize_h_kernel, dim3(gx, hmax, nb * T), dim3(256), 0, s, fr, bx, tmp, (const int32_t*)bounds, (const int32_t*)kk, T, H, W,
                           OW, omax);
        UNITE_LAUNCH_CHECK();
        hipLaunchKernelGGL(crop_resize_v_kernel, dim3(gx, OH, nb * T), dim3(256), 0, s, (const uint8_t*)tmp, bx, op, (const int32_t*)bounds,
                           (const int32_t*)kk, T, H, OH, OW, omax);
        UNITE_LAUNCH_CHECK();
    }
    return UNITE_OK;
}

extern "C" int unite_mask_sample(const float* weights, uint64_t seed, uint8_t* mask, int32_t* vis_tokens, int32_t* vis_rows_cls,
                                 int32_t BT, int32_t N, int32_t n_vis, void* stream) {
    if (!weights || !mask || !vis_tokens || BT <= 0 || N <= 0 || N > MASK_NMAX || n_vis <= 0 || n_vis > N) return UNITE_EINVAL;
    hipLaunchKernelGGL(mask_sample_kernel, dim3(BT), dim3(mask_threads(N)), 0, (hipStream_t)stream, weights, seed, (const uint64_t*)nullptr,
                       (const int64_t*)nullptr, (const uint8_t*)nullptr, mask, vis_tokens, vis_rows_cls, BT, N, n_vis);
    UNITE_LAUNCH_CHECK();
    return UNITE_OK;
}

extern "C" int unite_mask_sample_dev(const float* weights, const uint64_t* seed_dev, uint8_t* mask, int32_t* vis_tokens, int32_t* vis_rows_cls,
                                     int32_t BT, int32_t N, int32_t n_vis, void* stream) {
    if (!weights || !seed_dev || !mask || !vis_tokens || BT <= 0 || N <= 0 || N > MASK_NMAX || n_vis <= 0 || n_vis > N) return UNITE_EINVAL;
    hipLaunchKernelGGL(mask_sample_kernel, dim3(BT), dim3(mask_threads(N)), 0, (hipStream_t)stream, weights, 0ull, seed_dev, (const int64_t*)nullptr,
                       (const uint8_t*)nullptr, mask, vis_tokens, vis_rows_cls, BT, N, n_vis);
    UNITE_LAUNCH_CHECK();
    return UNITE_OK;
}

extern "C" int unite_drop_path_scales(const float* keep, uint64_t seed, float* out, int32_t layers, int32_t per_layer, void* stream) {
    if (!keep || !out || layers <= 0 || per_layer <= 0) return UNITE_EINVAL;
    const int total = layers * per_layer;
    hipLaunchKernelGGL(drop_path_scales_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, keep, seed, (const uint64_t*)nullptr, out,
                       per_layer, total);
    UNITE_LAUNCH_CHECK();
    return UNITE_OK;
}

extern "C" int unite_drop_path_scales_dev(const float* keep, const uint64_t* seed_dev, float* out, int32_t layers, int32_t per_layer, void* stream) {
    if (!keep || !seed_dev || !out || layers <= 0 || per_layer <= 0) return UNITE_EINVAL;
    const int total = layers * per_layer;
    hipLaunchKernelGGL(drop_path_scales_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, keep, 0ull, seed_dev, out, per_layer, total);
    UNITE_LAUNCH_CHECK();
    return UNITE_OK;
}

extern "C" int unite_mask_from_importance(const int64_t* importance, uint8_t* mask, int32_t* vis_tokens, int32_t* vis_rows_cls,
                                          int32_t BT, int32_t N, int32_t n_vis, void* stream) {
    if (!importance || !mask || !vis_tokens || BT <= 0 || N <= 0 || N > MASK_NMAX || n_vis <= 0 || n_vis > N) return UNITE_EINVAL;
    hipLaunchKernelGGL(mask_sample_kernel, dim3(BT), dim3(mask_threads(N)), 0, (hipStream_t)stream, (const float*)nullptr, 0ull, (const uint64_t*)nullptr, importance,
                       (const uint8_t*)nullptr, mask, vis_tokens, vis_rows_cls, BT, N, n_vis);
    UNITE_LAUNCH_CHECK();
    return UNITE_OK;
}

extern "C" int unite_mask_to_tokens(const uint8_t* mask, int32_t* vis_tokens, int32_t* vis_rows_cls, int32_t BT, int32_t N,
                                    int32_t n_vis, void* stream) {
    if (!mask || !vis_tokens || BT <= 0 || N <= 0 || N > MASK_NMAX || n_vis <= 0 || n_vis > N) return UNITE_EINVAL;
    hipLaunchKernelGGL(mask_sample_kernel, dim3(BT), dim3(mask_threads(N)), 0, (hipStream_t)stream, (const float*)nullptr, 0ull, (const uint64_t*)nullptr,
                       (const int64_t*)nullptr, mask, (uint8_t*)nullptr, vis_tokens, vis_rows_cls, BT, N, n_vis);
    UNITE_LAUNCH_CHECK();
    return UNITE_OK;
}

extern "C" int unite_greedy_masks(const float* weights, int32_t k, uint8_t* mask, int32_t* vis_tokens, int32_t* vis_rows_cls, int32_t BT,
                                  int32_t N, int32_t n_vis, void* stream) {
    if (!weights || !mask || !vis_tokens || k <= 0 || BT <= 0 || N <= 0 || N > MASK_NMAX || n_vis <= 0 || (int64_t)n_vis * k > N) return UNITE_EINVAL;
    hipLaunchKernelGGL(greedy_masks_kernel, dim3(BT), dim3(mask_threads(N)), 0, (hipStream_t)stream, weights, k, mask, vis_tokens, vis_rows_cls, BT, N, n_vis);
    UNITE_LAUNCH_CHECK();
    return UNITE_OK;
}

extern "C" int unite_pseudo_label_select(const float* logits_full, const float* logits_masked, int32_t k, const float* clip_probs,
                                         const int64_t* labels_t, int32_t strategy, float threshold, float clip_threshold,
                                         int32_t conf_weighted, int64_t* pseudo, float* weight, uint8_t* sel, float* msp, int32_t B, int32_t C,
                                         void* stream) {
    if (!logits_full || !logits_masked || !pseudo || !weight || B <= 0 || C <= 0 || k <= 0 || strategy < 0 || strategy > 6) return UNITE_EINVAL;
    if ((strategy == 4 || strategy == 5) && !clip_probs) return UNITE_EINVAL;
    hipLaunchKernelGGL(pseudo_label_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, logits_full, logits_masked, k, clip_probs,
                       labels_t, strategy, threshold, clip_threshold, conf_weighted, pseudo, weight, sel, msp, B, C);
    UNITE_LAUNCH_CHECK();
    return UNITE_OK;
}

extern "C" int unite_attn_cls_probs(const void* qkv, float* probs, int32_t B, int32_t N, int32_t H, float scale, void* stream) {
    if (!qkv || !probs || B <= 0 || N <= 1 || N > 256 * CLS_KPT || H <= 0) return UNITE_EINVAL;
    hipLaunchKernelGGL(attn_cls_probs_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)qkv, probs, N, H, scale);
    UNITE_LAUNCH_CHECK();
    return UNITE_OK;
}

extern "C" int unite_clip_similarity(const float* img, const float* text, float* out, int32_t B, int32_t T, int32_t C, int32_t n_cls,
                                     float scale, void* stream) {
    if (!img || !text || !out || B <= 0 || T <= 0 || C <= 0 || (C & 3) || n_cls <= 0 || n_cls > 256) return UNITE_EINVAL;
    hipLaunchKernelGGL(clip_similarity_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, img, text, out, T, C, n_cls, scale);
    UNITE_LAUNCH_CHECK();
    return UNITE_OK;
}

extern "C" int unite_clip_u8_to_f32(const uint8_t* frames, float* out, const uint8_t* flip, const float* mean3, const float* std3, int32_t B,
                                    int32_t T, int32_t H, int32_t W, void* stream) {
    if (!frames || !out || !mean3 || !std3 || B <= 0 || T <= 0 || H <= 0 || W <= 0 || (W & 3) || (((uintptr_t)frames) & 3) ||
        (((uintptr_t)out) & 15))
        return UNITE_EINVAL;
    const size_t n_quads = (size_t)B * T * H * (W / 4);
    const size_t blocks = (n_quads + 255) / 256;
    hipLaunchKernelGGL(clip_u8_to_f32_kernel, dim3((unsigned)(blocks > 16384 ? 16384 : blocks)), dim3(256), 0, (hipStream_t)stream, frames, out, flip,
                       mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], T, H, W, n_quads);
    UNITE_LAUNCH_CHECK();
    return UNITE_OK;
}

extern "C" int unite_resize_bicubic(const float* src, float* dst, int32_t planes, int32_t H, int32_t W, int32_t OH, int32_t OW,
                                    void* stream) {
    if (!src || !dst || planes <= 0 || planes > 65535 || H <= 0 || W <= 0 || OH <= 0 || OH > 65535 || OW <= 0) return UNITE_EINVAL;
    hipLaunchKernelGGL(resize_bicubic_kernel, dim3((OW + 255) / 256, OH, planes), dim3(256), 0, (hipStream_t)stream, src, dst, H, W, OH, OW,
                       (float)H / (float)OH, (float)W / (float)OW);
    UNITE_LAUNCH_CHECK();
    return UNITE_OK;
}

extern "C" int unite_token_mean_fwd(const float* x, float* out, int32_t B, int32_t N, int32_t D, void* stream) {
    if (!x || !out || B <= 0 || N <= 0 || D <= 0 || (D & 3)) return UNITE_EINVAL;
    hipLaunchKernelGGL(token_mean_fwd_kernel, dim3((D + 63) / 64, B), dim3(256), 0, (hipStream_t)stream, x, out, N, D);
    UNITE_LAUNCH_CHECK();
    return UNITE_OK;
}

extern "C" int unite_token_mean_bwd(const float* dout, float* dx, int32_t accumulate, int32_t B, int32_t N, int32_t D, void* stream) {
    if (!dout || !dx || B <= 0 || N <= 0 || D <= 0 || (D & 3)) return UNITE_EINVAL;
    const size_t total4 = (size_t)B * N * D / 4;
    hipLaunchKernelGGL(token_mean_bwd_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dout, dx, accumulate,
                       N, D, total4);
    UNITE_LAUNCH_CHECK();
    return UNITE_OK;
}

extern "C" int unite_linear_f32_fwd(const float* x, const float* W, const float* bias, float* y, int32_t B, int32_t C, int32_t D, void* stream) {
    if (!x || !W || !y || B <= 0 || C <= 0 || D <= 0 || (D & 3)) return UNITE_EINVAL;
    hipLaunchKernelGGL(linear_f32_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, x, W, bias, y, C, D);
    UNITE_LAUNCH_CHECK();
    return UNITE_OK;
}

extern "C" int unite_linear_f32_bwd(const float* x, const float* W, const float* dy, float* dx, float* dW, float* db, int32_t B, int32_t C,
                                    int32_t D, int32_t accumulate, void* stream) {
    if (!x || !W || !dy || B <= 0 || C <= 0 || D <= 0) return UNITE_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (dx) {
        hipLaunchKernelGGL(linear_f32_dx_kernel, dim3(B), dim3(256), 0, s, dy, W, dx, C, D);
        UNITE_LAUNCH_CHECK();
    }
    if (dW) {
        hipLaunchKernelGGL(linear_f32_dw_kernel, dim3(C), dim3(256), 0, s, dy, x, dW, db, B, C, D, accumulate);
        UNITE_LAUNCH_CHECK();
    }
    return UNITE_OK;
}

// ------------------------------------------------------------------------------------ element-wise regression losses (clip_loss_type)
// loss_sum += sum_i f(o_i - t_i);  grad_i = grad_scale * f'(o_i - t_i)   with f = d^2 (0: mse), |d| (1: l1), Huber beta = 1 (2: smooth_l1)
__global__ __launch_bounds__(256) void pointwise_loss_kernel(const float* __restrict__ o, const float* __restrict__ t, int kind, float grad_scale,
                                                             float* __restrict__ loss_sum, float* __restrict__ grad, size_t n4) {
    __shared__ float wl[4];
    float l = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const f32x4 a = ((const f32x4*)o)[i], b = ((const f32x4*)t)[i];
        f32x4 g;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float d = a[e] - b[e], ad = fabsf(d), sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
            if (kind == 0) { l += d * d; g[e] = 2.f * d; }
            else if (kind == 1) { l += ad; g[e] = sg; }
            else { l += ad < 1.f ? 0.5f * d * d : ad - 0.5f; g[e] = ad < 1.f ? d : sg; }
            g[e] *= grad_scale;
        }
        if (grad) ((f32x4*)grad)[i] = g;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) l += __shfl_xor(l, m);
    if ((threadIdx.x & 63) == 0) wl[threadIdx.x >> 6] = l;
    __syncthreads();
    if (threadIdx.x == 0 && loss_sum) atomicAdd(loss_sum, wl[0] + wl[1] + wl[2] + wl[3]);
}

extern "C" int unite_pointwise_loss(const float* out, const float* target, int32_t kind, float grad_scale, float* loss_sum, float* grad,
                                    int64_t n, void* stream) {
    if (!out || !target || kind < 0 || kind > 2 || n <= 0 || (n & 3) || (((uintptr_t)out | (uintptr_t)target | (uintptr_t)grad) & 15))
        return UNITE_EINVAL;
    const size_t n4 = (size_t)n / 4;
    const unsigned grid = (unsigned)((n4 + 255) / 256 < 1024 ? (n4 + 255) / 256 : 1024);
    hipLaunchKernelGGL(pointwise_loss_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, out, target, kind, grad_scale, loss_sum, grad, n4);
    UNITE_LAUNCH_CHECK();
    return UNITE_OK;
}

extern "C" int unite_softmax_ce(const float* logits, const int64_t* labels, const float* row_weight, float grad_scale, float* loss_sum,
                                float* dlogits, int32_t M, int32_t C, void* stream) {
    if (!logits || !labels || M <= 0 || C <= 0 || C > 1024) return UNITE_EINVAL;
    hipLaunchKernelGGL(softmax_ce_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, logits, labels, row_weight, grad_scale,
                       loss_sum, dlogits, M, C);
    UNITE_LAUNCH_CHECK();
    return UNITE_OK;
}

extern "C" int unite_scale_cast_bf16(const float* x, const float* row_scale, int32_t rows_per_scale, void* y, int32_t M, int32_t D,
                                     void* stream) {
    if (!x || !y || M <= 0 || D <= 0 || (D & 3) || (row_scale && rows_per_scale <= 0)) return UNITE_EINVAL;
    const size_t total4 = (size_t)M * D / 4;
    const size_t blocks = (total4 + 255) / 256;
    hipLaunchKernelGGL(scale_cast_bf16_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, (hipStream_t)stream, x, row_scale,
                       rows_per_scale, (uint16_t*)y, D, total4);
    UNITE_LAUNCH_CHECK();
    return UNITE_OK;
}

extern "C" int unite_cast_f32_bf16(const float* src, void* dst, int64_t n, void* stream) {
    if (!src || !dst || n <= 0) return UNITE_EINVAL;
    const int64_t blocks = (n / 8 + 255) / 256;
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3((unsigned)(blocks > 2048 ? 2048 : (blocks < 1 ? 1 : blocks))), dim3(256), 0,
                       (hipStream_t)stream, src, (uint16_t*)dst, n);
    UNITE_LAUNCH_CHECK();
    return UNITE_OK;
}


// ---- stage-2 / stage-3 input path (src/datasets/kinetics_sparse.py) -----------------------------------------------------------------------
// (1) validation / test: Resize(short side, 'bilinear') of the decoded uint8 frames = cv2.resize(..., INTER_LINEAR) (functional_umt.py:58-66).
// OpenCV's 8-bit linear resize in fixed point, as published (imgproc/resize.cpp): per output column fx = (float)((dx + 0.5) * scale_x - 0.5),
// sx = floor(fx), the fraction in 11 bits (saturate_cast<short>(f * 2048), round half to even), clamped at the borders (fraction 0); rows alike;
// horizontal pass in int (S[sx] * a0 + S[sx + 1] * a1), vertical pass  (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2.
// cv2 is not in the build image: this is held bit for bit to oracle/cv2_resize.py (the same published algorithm in numpy), "parity unpinned".
__device__ __forceinline__ void cv_linear_coef(int d, double scale, int n_src, int& s0, int& s1, int& a0, int& a1, bool horizontal) {
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (horizontal) {                       // columns: the fraction is dropped at both borders, the second tap is never read past the row
        if (s < 0) { f = 0.f; s = 0; }
        if (s >= n_src - 1) { f = 0.f; s = n_src - 1; }
        s0 = s;
        s1 = min(s + 1, n_src - 1);
    } else {                                // rows: the two row indices are clipped, the weights are not
        s0 = min(max(s, 0), n_src - 1);
        s1 = min(max(s + 1, 0), n_src - 1);
    }
    a0 = (int)(short)__float2int_rn((1.f - f) * 2048.f);
    a1 = (int)(short)__float2int_rn(f * 2048.f);
}

__global__ __launch_bounds__(256) void resize_u8_linear_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int T, int H, int W, int OH,
                                                               int OW, double scale_x, double scale_y) {
    const size_t total = (size_t)T * OH * OW;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int dx = (int)(i % OW), dy = (int)((i / OW) % OH), t = (int)(i / ((size_t)OW * OH));
        int x0, x1, a0, a1, y0, y1, b0, b1;
        cv_linear_coef(dx, scale_x, W, x0, x1, a0, a1, true);
        cv_linear_coef(dy, scale_y, H, y0, y1, b0, b1, false);
        const uint8_t* r0 = src + ((size_t)t * H + y0) * W * 3;
        const uint8_t* r1 = src + ((size_t)t * H + y1) * W * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int h0 = (int)r0[x0 * 3 + c] * a0 + (int)r0[x1 * 3 + c] * a1;
            const int h1 = (int)r1[x0 * 3 + c] * a0 + (int)r1[x1 * 3 + c] * a1;
            dst[i * 3 + c] = (uint8_t)((((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2);
        }
    }
}

// (2) training: everything behind RandAugment in _aug_frame (kinetics_sparse.py:232-262) -- ToTensor (/ 255), tensor_normalize, random_resized_crop
// (crop box, then torch.nn.functional.interpolate(bilinear, align_corners=False) to S x S, video_transforms.py:560-592), horizontal flip -- in ONE
// pass over the uint8 frames.  ATen's arithmetic: source index max(scale * (d + 0.5) - 0.5, 0) with scale = in / out in f32, second tap one further
// unless at the border, out = wy0 (wx0 p00 + wx1 p01) + wy1 (wx0 p10 + wx1 p11) on the NORMALISED pixels (the reference normalises before it
// crops).  The flip mirrors the output column (it comes after the interpolation in the reference).
__global__ __launch_bounds__(256) void train_clip_kernel(const uint8_t* __restrict__ frames, float* __restrict__ out, int T, int H, int W, int S,
                                                         int ci, int cj, int ch, int cw, int flip, float m0, float m1, float m2, float s0, float s1,
                                                         float s2) {
    const size_t total = (size_t)T * S * S;
    const float sh = (float)ch / (float)S, sw = (float)cw / (float)S;
    const float mean[3] = {m0, m1, m2}, sd[3] = {s0, s1, s2};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int x = (int)(i % S), y = (int)((i / S) % S), t = (int)(i / ((size_t)S * S));
        const int xs = flip ? S - 1 - x : x;
        const float fy = fmaxf(sh * ((float)y + 0.5f) - 0.5f, 0.f), fx = fmaxf(sw * ((float)xs + 0.5f) - 0.5f, 0.f);
        const int y0 = min((int)floorf(fy), ch - 1), x0 = min((int)floorf(fx), cw - 1);
        const float ly = fminf(fmaxf(fy - (float)y0, 0.f), 1.f), lx = fminf(fmaxf(fx - (float)x0, 0.f), 1.f);
        const int y1 = y0 + (y0 < ch - 1 ? 1 : 0), x1 = x0 + (x0 < cw - 1 ? 1 : 0);
        const uint8_t* base = frames + (size_t)t * H * W * 3;
        const uint8_t* p00 = base + ((size_t)(ci + y0) * W + cj + x0) * 3;
        const uint8_t* p01 = base + ((size_t)(ci + y0) * W + cj + x1) * 3;
        const uint8_t* p10 = base + ((size_t)(ci + y1) * W + cj + x0) * 3;
        const uint8_t* p11 = base + ((size_t)(ci + y1) * W + cj + x1) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v00 = ((float)p00[c] / 255.0f - mean[c]) / sd[c], v01 = ((float)p01[c] / 255.0f - mean[c]) / sd[c];
            const float v10 = ((float)p10[c] / 255.0f - mean[c]) / sd[c], v11 = ((float)p11[c] / 255.0f - mean[c]) / sd[c];
            const float top = (1.f - lx) * v00 + lx * v01, bot = (1.f - lx) * v10 + lx * v11;
            out[(((size_t)c * T + t) * S + y) * S + x] = (1.f - ly) * top + ly * bot;
        }
    }
}

// ---- diagnostic: shader clock under load (include/unite_hip.h: unite_clock_probe) ---------------------------------------------------------
namespace {
__global__ __launch_bounds__(64) void clock_probe_kernel(uint64_t* __restrict__ out, int samples, uint64_t interval_ticks) {
    if (threadIdx.x != 0) return;
    uint64_t next = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < samples; ++i) {
        // bounded wait for the next sample time: the reference counter always advances, so every iteration ends
        while ((int64_t)(__builtin_amdgcn_s_memrealtime() - next) < 0) __builtin_amdgcn_s_sleep(32);
        const uint64_t c = __builtin_amdgcn_s_memtime(), r = __builtin_amdgcn_s_memrealtime();
        out[2 * i] = c;
        out[2 * i + 1] = r;
        next += interval_ticks;
    }
}
// one {shader clock, reference clock} pair per XCD, taken by whichever workgroups land there (the counters are per XCD: only pairs of the
// SAME XCD may be subtracted); an ordinary short kernel in the caller's stream, unlike the resident probe wave above
__global__ __launch_bounds__(64) void clock_stamp_kernel(uint64_t* __restrict__ out) {
    if (threadIdx.x != 0) return;
    uint32_t xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 7u;
    out[2 * xcc] = __builtin_amdgcn_s_memtime();
    out[2 * xcc + 1] = __builtin_amdgcn_s_memrealtime();
}
}  // namespace

extern "C" int unite_clock_stamp(uint64_t* out, void* stream) {
    if (!out) return UNITE_EINVAL;
    hipLaunchKernelGGL(clock_stamp_kernel, dim3(64), dim3(64), 0, (hipStream_t)stream, out);
    UNITE_LAUNCH_CHECK();
    return UNITE_OK;
}

extern "C" int unite_clock_probe(uint64_t* samples_out, int32_t samples, int32_t interval_us, void* stream) {
    if (!samples_out || samples <= 0 || samples > 65536 || interval_us < 1 || interval_us > 100000) return UNITE_EINVAL;
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, samples_out, samples, (uint64_t)interval_us * 100u);
    UNITE_LAUNCH_CHECK();
    return UNITE_OK;
}


extern "C" int unite_resize_u8_linear(const uint8_t* frames, uint8_t* out, int32_t T, int32_t H, int32_t W, int32_t OH, int32_t OW, void* stream) {
    if (!frames || !out || T <= 0 || H <= 0 || W <= 0 || OH <= 0 || OW <= 0) return UNITE_EINVAL;
    const size_t total = (size_t)T * OH * OW, blocks = (total + 255) / 256;
    // OpenCV: inv_scale = dsize / ssize in double, scale = 1 / inv_scale
    const double scale_x = 1.0 / ((double)OW / (double)W), scale_y = 1.0 / ((double)OH / (double)H);
    hipLaunchKernelGGL(resize_u8_linear_kernel, dim3((unsigned)(blocks > 16384 ? 16384 : blocks)), dim3(256), 0, (hipStream_t)stream, frames, out, T, H, W,
                       OH, OW, scale_x, scale_y);
    UNITE_LAUNCH_CHECK();
    return UNITE_OK;
}

extern "C" int unite_train_clip_u8(const uint8_t* frames, float* out, int32_t T, int32_t H, int32_t W, int32_t S, int32_t crop_i, int32_t crop_j,
                                   int32_t crop_h, int32_t crop_w, int32_t flip, const float* mean3, const float* std3, void* stream) {
    if (!frames || !out || !mean3 || !std3 || T <= 0 || H <= 0 || W <= 0 || S <= 0 || crop_h <= 0 || crop_w <= 0 || crop_i < 0 || crop_j < 0 ||
        crop_i + crop_h > H || crop_j + crop_w > W)
        return UNITE_EINVAL;
    const size_t total = (size_t)T * S * S, blocks = (total + 255) / 256;
    hipLaunchKernelGGL(train_clip_kernel, dim3((unsigned)(blocks > 16384 ? 16384 : blocks)), dim3(256), 0, (hipStream_t)stream, frames, out, T, H, W, S,
                       crop_i, crop_j, crop_h, crop_w, flip ? 1 : 0, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2]);
    UNITE_LAUNCH_CHECK();
    return UNITE_OK;
}
