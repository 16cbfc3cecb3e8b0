// Fused AdamW + global gradient norm over ONE flat fp32 parameter buffer (HBM-bound streaming kernels).
// Algorithmic traffic of the AdamW step: read p, g, m, v (16 B) + write p, m, v (12 B) + bf16 shadow (2 B) = 30 B / element.
#include "common.h"
#include <string.h>

namespace {

constexpr int MAX_GROUPS = 64;
struct GroupTable {
    float lr[MAX_GROUPS];
    float wd[MAX_GROUPS];
};

__global__ __launch_bounds__(256) void adamw_flat_kernel(float* __restrict__ param, const float* __restrict__ grad, float* __restrict__ m,
                                                         float* __restrict__ v, uint16_t* __restrict__ pbf, const uint8_t* __restrict__ chunk_group,
                                                         int64_t n, GroupTable tab, float beta1, float beta2, float eps, float inv_bc1,
                                                         float inv_sqrt_bc2, const float* __restrict__ grad_scale_dev,
                                                         const int32_t* __restrict__ found_inf_dev, const float* __restrict__ hp_dev) {
    if (found_inf_dev && *found_inf_dev) return;
    if (hp_dev) {      // device-resident hyper-parameters (lr[64], wd[64], 1 / bc1, 1 / sqrt(bc2)): the launch is replayable from a HIP graph
        inv_bc1 = hp_dev[2 * MAX_GROUPS];
        inv_sqrt_bc2 = hp_dev[2 * MAX_GROUPS + 1];
    }
    const float gs = grad_scale_dev ? *grad_scale_dev : 1.0f;
    const int64_t nchunks = (n + 1023) >> 10;
    for (int64_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        const int gidx = chunk_group ? chunk_group[chunk] : 0;
        const float lr = hp_dev ? hp_dev[gidx] : tab.lr[gidx], wd = hp_dev ? hp_dev[MAX_GROUPS + gidx] : tab.wd[gidx];
        if (lr < 0.f) continue;              // group without gradients this step: torch.optim.AdamW's `if p.grad is None: continue`
        const float decay = 1.0f - lr * wd, step = lr * inv_bc1;
        const int64_t i = (chunk << 10) + threadIdx.x * 4;
        if (i + 4 <= n) {
            f32x4 p = *(f32x4*)(param + i), g = *(const f32x4*)(grad + i), mm = *(f32x4*)(m + i), vv = *(f32x4*)(v + i);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float ge = g[e] * gs;
                p[e] *= decay;
                mm[e] = beta1 * mm[e] + (1.0f - beta1) * ge;
                vv[e] = beta2 * vv[e] + (1.0f - beta2) * ge * ge;
                p[e] -= step * mm[e] / (sqrtf(vv[e]) * inv_sqrt_bc2 + eps);
            }
            *(f32x4*)(param + i) = p;
            *(f32x4*)(m + i) = mm;
            *(f32x4*)(v + i) = vv;
            if (pbf) *(u32x2*)(pbf + i) = (u32x2){pack_bf16x2(p[0], p[1]), pack_bf16x2(p[2], p[3])};
        } else {
            for (int64_t k = i; k < n; ++k) {
                const float ge = grad[k] * gs;
                float pe = param[k] * decay;
                const float me = beta1 * m[k] + (1.0f - beta1) * ge;
                const float ve = beta2 * v[k] + (1.0f - beta2) * ge * ge;
                pe -= step * me / (sqrtf(ve) * inv_sqrt_bc2 + eps);
                param[k] = pe; m[k] = me; v[k] = ve;
                if (pbf) pbf[k] = f32_to_bf16(pe);
            }
        }
    }
}

constexpr int GN_BLOCKS = 1024;
// chunk_group / skip_group: 1024-element chunks of that optimizer group are left out (parameters without a gradient this step --
// frozen, or in layers that were not executed: torch's get_grad_norm_ only sees p.grad is not None, utils.py:631-643)
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ partial,
                                                            const uint8_t* __restrict__ chunk_group, int skip_group) {
    __shared__ float red[4];
    float s = 0.f;
    const int64_t stride = (int64_t)gridDim.x * 1024;
    for (int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x * 4; i < n; i += stride) {
        if (chunk_group && chunk_group[i >> 10] == skip_group) continue;
        if (i + 4 <= n) {
            const f32x4 a = *(const f32x4*)(g + i);
            s += a[0] * a[0] + a[1] * a[1] + a[2] * a[2] + a[3] * a[3];
        } else {
            for (int64_t k = i; k < n; ++k) s += g[k] * g[k];
        }
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ __launch_bounds__(256) void sumsq_final_kernel(const float* __restrict__ partial, int nparts, float max_norm, float* __restrict__ norm_out,
                                                          float* __restrict__ clip_coef_out) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < nparts; i += 256) s += partial[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float nrm = sqrtf(red[0] + red[1] + red[2] + red[3]);
        norm_out[0] = nrm;
        if (clip_coef_out) clip_coef_out[0] = (max_norm > 0.f) ? fminf(1.0f, max_norm / (nrm + 1e-6f)) : 1.0f;
    }
}

}  // namespace

extern "C" int unite_adamw_flat(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, void* param_bf16,
                                const uint8_t* chunk_group, int64_t n_elems, const float* lr, const float* weight_decay,
                                int32_t n_groups, float beta1, float beta2, float eps, int32_t step, const float* grad_scale_dev,
                                const int32_t* found_inf_dev, void* stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || !lr || !weight_decay || n_elems <= 0 || n_groups <= 0 || n_groups > MAX_GROUPS ||
        step <= 0)
        return UNITE_EINVAL;
    if ((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) || (((uintptr_t)param_bf16) & 7))
        return UNITE_EINVAL;
    GroupTable tab;
    for (int i = 0; i < MAX_GROUPS; ++i) {
        tab.lr[i] = i < n_groups ? lr[i] : 0.f;
        tab.wd[i] = i < n_groups ? weight_decay[i] : 0.f;
    }
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    const int64_t nchunks = (n_elems + 1023) >> 10;
    const unsigned grid = (unsigned)(nchunks < 256 * 16 ? nchunks : 256 * 16);
    hipLaunchKernelGGL(adamw_flat_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq,
                       (uint16_t*)param_bf16, chunk_group, n_elems, tab, beta1, beta2, eps, (float)(1.0 / bc1), (float)(1.0 / sqrt(bc2)),
                       grad_scale_dev, found_inf_dev, (const float*)nullptr);
    UNITE_LAUNCH_CHECK();
    return UNITE_OK;
}

extern "C" int unite_adamw_flat_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, void* param_bf16,
                                    const uint8_t* chunk_group, int64_t n_elems, const float* hp_dev, float beta1, float beta2, float eps,
                                    const float* grad_scale_dev, const int32_t* found_inf_dev, void* stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || !hp_dev || n_elems <= 0) return UNITE_EINVAL;
    if ((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) || (((uintptr_t)param_bf16) & 7))
        return UNITE_EINVAL;
    GroupTable tab;
    memset(&tab, 0, sizeof(tab));
    const int64_t nchunks = (n_elems + 1023) >> 10;
    const unsigned grid = (unsigned)(nchunks < 256 * 16 ? nchunks : 256 * 16);
    hipLaunchKernelGGL(adamw_flat_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq,
                       (uint16_t*)param_bf16, chunk_group, n_elems, tab, beta1, beta2, eps, 1.0f, 1.0f, grad_scale_dev, found_inf_dev, hp_dev);
    UNITE_LAUNCH_CHECK();
    return UNITE_OK;
}

extern "C" size_t unite_grad_norm_workspace(int64_t n) {
    (void)n;
    return GN_BLOCKS * sizeof(float);
}

static int grad_norm_impl(const float* grad, int64_t n, const uint8_t* chunk_group, int skip_group, float max_norm, float* norm_out,
                          float* clip_coef_out, void* workspace, void* stream);

extern "C" int unite_grad_norm_flat(const float* grad, int64_t n, float max_norm, float* norm_out, float* clip_coef_out, void* workspace,
                                    void* stream) {
    return grad_norm_impl(grad, n, nullptr, -1, max_norm, norm_out, clip_coef_out, workspace, stream);
}

extern "C" int unite_grad_norm_flat_masked(const float* grad, int64_t n, const uint8_t* chunk_group, int32_t skip_group, float max_norm,
                                           float* norm_out, float* clip_coef_out, void* workspace, void* stream) {
    if (!chunk_group || skip_group < 0 || skip_group > 255) return UNITE_EINVAL;
    return grad_norm_impl(grad, n, chunk_group, skip_group, max_norm, norm_out, clip_coef_out, workspace, stream);
}

static int grad_norm_impl(const float* grad, int64_t n, const uint8_t* chunk_group, int skip_group, float max_norm, float* norm_out,
                          float* clip_coef_out, void* workspace, void* stream) {
    if (!grad || !norm_out || !workspace || n <= 0 || (((uintptr_t)grad) & 15)) return UNITE_EINVAL;
    const int64_t want = (n + 1023) / 1024;
    const int nparts = (int)(want < GN_BLOCKS ? want : GN_BLOCKS);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(nparts), dim3(256), 0, s, grad, n, (float*)workspace, chunk_group, skip_group);
    UNITE_LAUNCH_CHECK();
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, s, (const float*)workspace, nparts, max_norm, norm_out, clip_coef_out);
    UNITE_LAUNCH_CHECK();
    return UNITE_OK;
}

extern "C" int unite_abi_version(void) { return UNITE_ABI_VERSION; }
extern "C" const char* unite_target_arch(void) { return "gfx950"; }
