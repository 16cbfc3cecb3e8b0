/*
 * unite_hip.h -- C ABI of libunite_hip.so: the MI355X (gfx950) kernels under UNITE's
 * data-parallel training hot path (reddyav1/unite: run_stage1.py:294-505 and the model
 * code it calls).  The reference has no FFI of its own: every op below replaces an
 * ATen/cuDNN call the reference reaches through torch.nn (file:line cited per entry).
 * INTEGRATION.md shows the ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer borrowed from the caller (PyTorch owns the memory);
 *     the library never allocates, frees or synchronises; workspace is passed in.
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*).
 *   - return value: 0 on success, a negative UNITE_E* code for bad arguments, or a
 *     positive hipError_t from the launch.  Nothing throws across the boundary.
 *   - matrices are row-major; "bf16" is the upper 16 bits of an IEEE f32 (uint16_t storage).
 *   - all leading dimensions and column counts of bf16 matrices must be multiples of 8
 *     (16-byte rows), base pointers 16-byte aligned, and every buffer < 2 GiB.
 */
#ifndef UNITE_HIP_H
#define UNITE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UNITE_OK 0
#define UNITE_EINVAL (-1)     /* bad shape / alignment / null pointer */
#define UNITE_ENOSUP (-2)     /* shape outside what the kernels were built for */

#define UNITE_ABI_VERSION 2
int unite_abi_version(void);
/* name of the code-object target the library was built for ("gfx950") */
const char* unite_target_arch(void);

/* ------------------------------------------------------------------------------------
 * GEMM  out[M,N] = epilogue( op(A)[M,K] * op(B)[K,N] ),  bf16 operands, f32 accumulate (MFMA).
 *   trans_a = 0 : A stored [M,K] (lda >= K)        trans_a = 1 : A stored [K,M] (lda >= M)
 *   trans_b = 0 : B stored [N,K] (ldb >= K), i.e. an nn.Linear weight [out,in]
 *   trans_b = 1 : B stored [K,N] (ldb >= N)
 * Replaces F.linear / nn.Linear / Conv3d-as-GEMM / x @ proj and their autograd backward:
 *   modeling_finetune.py:106,117 (qkv, proj), :67,71 (fc1, fc2), :165-174 (patch embed),
 *   modeling_adaptation.py:204 (decoder head), clip.py:38-44,123-128,170.
 * Epilogue, applied in this order on the f32 accumulator v:
 *   v += bias[n]                                   (bias != NULL)
 *   act == UNITE_ACT_GELU      : if aux_out: aux_out[m,n] = bf16(v); v = gelu_erf(v)
 *   act == UNITE_ACT_QUICKGELU : v = v * sigmoid(1.702 v)                       (clip.py:29-31)
 *   act == UNITE_ACT_DGELU     : v *= gelu_erf'(aux_in[m,n])                    (backward of fc1 from the saved pre-activation)
 *   act == UNITE_ACT_GELU_DSAVE: aux_out[m,n] = q16(gelu_erf'(v)); v = gelu_erf(v)    (aux_out required: the derivative is saved, not the input)
 *   act == UNITE_ACT_MULAUX    : v *= unq16(aux_in[m,n])                        (backward of fc1 from the saved derivative)
 *       q16(d) = round(65535 (d + 0.25) / 2) as an unsigned 16-bit number (the derivative lies in [-0.13, 1.13]); unq16(q) = 2 q / 65535 - 0.25.
 *       The buffer has the size and leading dimension of a bf16 one; its contents are opaque to the caller.
 *   v *= row_scale[m / rows_per_scale]             (row_scale != NULL; stochastic depth)
 *   v += residual[m,n]                             (residual != NULL; f32, bf16 if residual_bf16 == 1, IEEE half if == 2; ld = ldr)
 *   v += out[m,n]                                  (accumulate != 0, f32 out only)
 *   out[m,n] = v   as f32 (out_f32 != 0) or bf16;  out_bf16_copy[m,n] = bf16(v) if given;  colsum_out (+)= column sums of out.
 * ------------------------------------------------------------------------------------ */
enum { UNITE_ACT_NONE = 0, UNITE_ACT_GELU = 1, UNITE_ACT_QUICKGELU = 2, UNITE_ACT_DGELU = 3, UNITE_ACT_GELU_DSAVE = 4, UNITE_ACT_MULAUX = 5 };

typedef struct unite_gemm_args {
    int32_t M, N, K;
    int32_t trans_a, trans_b;
    const void* A; int32_t lda;
    const void* B; int32_t ldb;
    const float* bias;
    int32_t act;
    const void* aux_in;  int32_t ld_aux_in;     /* bf16 [M,N] */
    void* aux_out;       int32_t ld_aux_out;    /* bf16 [M,N] */
    const float* row_scale; int32_t rows_per_scale;
    const void* residual;   int32_t ldr;        /* f32 [M,N] (bf16 / half [M,N] if residual_bf16 == 1 / 2, see below) */
    void* out; int32_t ldc; int32_t out_f32; int32_t accumulate;
    void* out_bf16_copy; int32_t ld_copy;
    void* workspace; int64_t workspace_bytes;   /* optional scratch (16-byte aligned): lets short-and-wide products with a plain
                                                   f32 output (weight gradients) run split-K through f32 slabs [S][M][N],
                                                   summed in a fixed order (bitwise reproducible) by a second launch on the same
                                                   stream (default) or, with UNITE_SPLITK_SEPARATE=0, by the LAST slice of a tile to
                                                   finish inside the same launch; NULL = never split.  Layout: a header of
                                                   UNITE_WS_HEADER_BYTES (arrival counters) that must be ZERO before the first
                                                   call and is left zero by every call, then the slabs.  One workspace per
                                                   stream: two launches in flight must not share one. */
    float* colsum_out; int32_t colsum_accumulate;
                                                /* optional f32 [N]: colsum_out[n] (+)= sum over m of the STORED out[m,n] (after rounding
                                                   for a bf16 output) -- the bias gradient of the Linear whose input gradient this
                                                   product is (db = colsum(dY), modeling_finetune.py:67 under autograd), so that dY is
                                                   not read again for it.  Per-tile partial sums go through `workspace`
                                                   (>= unite_gemm_colsum_workspace(M, N) bytes), summed in a fixed order. */
    /* ---- ABI 2: everything below may be left zero (a zeroed struct asks for the process-wide defaults) ---- */
    int32_t plan_flags;                         /* bit 0: plan_persistent is a per-call hint, bit 1: plan_sharing is, bit 2: bit 3 is the
                                                   main-loop schedule of the tile kernels for THIS launch (0: fragment reads at the head
                                                   of each phase, 1: software-pipelined reads between the MFMAs; default UNITE_GEMM_SCHED,
                                                   1) -- same products bit for bit, an A/B switch; bit 4: bit 5 is the 256 x 256 kernel's epilogue form where
                                                   both apply (bf16 output, no residual / saved pre-activation / column sums / split-K): 0 f32 LDS
                                                   image in two passes, 1 transposed accumulators + one bf16 image (default UNITE_GEMM_EPI) --
                                                   same bits out */
    int32_t plan_persistent;                    /* as unite_gemm_set_policy, for THIS launch only */
    float   plan_sharing;                       /* as unite_gemm_set_sharing, for THIS launch only */
    int32_t residual_bf16;                      /* 1: `residual` points at bf16 [M,N] (the frozen teacher's bf16 residual stream).
                                                   2: `residual` points at IEEE-half [M,N] AND the 16-bit `out` is written as IEEE half, saturating at +-65504
                                                   (the f16 residual stream OpenAI's CLIP runs with; out_f32, out_bf16_copy, colsum_out and
                                                   act must be 0 / NULL then) */
    float* rowsum_a_out;                        /* optional f32 [M]: rowsum_a_out[m] (+)= sum_k op(A)[m,k] -- for a weight gradient
                                                   dW = dY^T X (trans_a = 1, A = dY stored [tokens, out]) this is the bias gradient
                                                   db = colsum(dY) (modeling_finetune.py:67,106 under autograd), taken from the A tiles
                                                   the product already holds in LDS.  Deep tile kernels only; with split-K the
                                                   per-slice partial sums go through `workspace` like the slabs (fixed order). */
    int32_t rowsum_accumulate;
    int32_t rowsum_zero_lo, rowsum_zero_hi;     /* rows in [lo, hi) are written as 0 instead (the k third of the packed (q,0,v) bias) */
} unite_gemm_args;

#define UNITE_WS_HEADER_BYTES 32768             /* 4096 tile counters + 4096 spare words */

size_t unite_gemm_colsum_workspace(int32_t M, int32_t N);

int unite_gemm_bf16(const unite_gemm_args* args, void* stream);

/* Kernel selection knob (diagnostics, tests, A/B runs inside one process): which products take the persistent 256 x 128
 * kernel (gemm_pp.hip: the epilogue of a tile runs under the main loop of the workgroup's next tile; needs A k-contiguous,
 * K % 64 == 0, K >= 768) instead of the tile kernels.  0 never, 1 the shapes it measured faster on (default), 2 whenever it
 * supports the problem, -1 back to the UNITE_GEMM_PP environment variable.  Results are the same either way (f32 accumulation
 * in K order inside a K-tile; integer-valued products are bit-exact on both). */
/* The setters below change PROCESS-WIDE DEFAULTS only; a caller that needs a different choice for some launches passes it in
 * unite_gemm_args.plan_* (no global is written on that path, so host threads do not see each other's hints). */
int unite_gemm_set_policy(int32_t persistent);
/* the value last set (-1 if the environment variable decides): lets a caller change the policy for a group of launches and put it back */
int unite_gemm_get_policy(void);

/* How much of the GPU the caller's launches share with independent work on other streams (stage 1: the frozen teacher runs one batch
 * ahead of the student, run_stage1.py:360-397 vs :410-456).  The planner picks tile size and split-K by
 *     (1 - w) * latency of the launch alone on the GPU  +  w * CU time it occupies (workgroups x time each / resident slots + reduction pass)
 * w = 0 (default): the launch is alone, a partially filled last round is wasted, split-K and small tiles pay; w -> 1: whatever it leaves
 * idle is used by the other stream, so fewer, larger workgroups win (256 x 256 tiles, fewer split-K slices).  0 <= w <= 1; results do
 * not depend on it beyond the f32 summation order of split-K.  UNITE_GEMM_PLAN_WORK in the environment pins the value. */
int unite_gemm_set_sharing(float work_weight);
float unite_gemm_get_sharing(void);
/* The planner's choice for a product with a plain f32 output that may split K through `slab_bytes` of workspace (beyond the header and the
 * row-sum area): *kind = 1 (128 x 128 tiles), 2 (256 x 256) or 3 (128 x 256), *splitk = number of K slices.  `sharing` as plan_sharing,
 * `rowsum` != 0 if rowsum_a_out will be set (tile kernels with the row sums only).  Host arithmetic only: callable without a GPU. */
int unite_gemm_plan(int32_t M, int32_t N, int32_t K, int32_t trans_a, int32_t trans_b, float sharing, int64_t slab_bytes, int32_t rowsum,
                    int32_t* kind, int32_t* splitk);

/* `count` (1..4) independent problems with the same trans_a / trans_b in ONE launch (no split-K, workspace ignored):
 * the four weight gradients of a transformer block (dW = dY^T X with K = tokens: modeling_finetune.py:67-71,106,117
 * under autograd) fill the chip together instead of each needing split-K slabs.  Results are identical to `count`
 * single calls without a workspace (same tile kernel, same accumulation order).  The fused bias sums are NOT available here:
 * a problem with rowsum_a_out or colsum_out set is refused with UNITE_EINVAL (they need the single-problem launch). */
int unite_gemm_bf16_grouped(const unite_gemm_args* args, int32_t count, void* stream);

/* Diagnostics for bench.py's roofline leg: when enabled, every unite_gemm_bf16 launch is bracketed by two HIP events
 * recorded on the launch stream (pool of max_launches pairs; launches beyond the pool are not timed).
 * unite_prof_summary synchronises on the recorded events and returns the summed durations (ms), the number of timed
 * launches, their algorithmic FLOPs (2 M N K each) and their algorithmic HBM bytes (every operand, output, residual and saved
 * pre-activation matrix once).  Not for use under stream capture. */
int unite_prof_enable(int32_t on, int32_t max_launches);
int unite_prof_summary(double* total_ms, int64_t* launches, double* total_flops, double* total_bytes);
/* Diagnostic: the shader clock the chip actually holds while other kernels run.  ONE wave (one workgroup of 64 threads, no LDS, a handful of
 * registers) takes `samples` pairs {shader-clock counter (s_memtime), 100-MHz reference counter (s_memrealtime)}, `interval_us` apart, sleeping
 * in between, and exits: samples_out is uint64 [samples][2] in device memory.  Launched on a stream of its own beside the work to observe; the
 * clock over an interval is (d s_memtime / d s_memrealtime) x 100 MHz (MI355X_MICROARCH.md, DVFS give-back item 6).  samples <= 65536,
 * 1 <= interval_us <= 100000: the wave always exits after samples x interval_us. */
int unite_clock_probe(uint64_t* samples_out, int32_t samples, int32_t interval_us, void* stream);
/* The same pair taken ONCE per XCD by a short kernel in the caller's stream: out is uint64 [8][2] (XCD, {s_memtime, s_memrealtime}; an XCD no
 * workgroup of the 64 landed on keeps its old contents).  Two stamps around a span of work give the average clock over it, per XCD -- usable
 * inside a multi-stream step, where a resident probe wave disturbs the schedule (measured: the stage-1 step takes 42 instead of 20 ms beside it). */
int unite_clock_stamp(uint64_t* out, void* stream);

/* ------------------------------------------------------------------------------------
 * LayerNorm over the last dim (D % 4 == 0, D <= 1024), one wavefront per row, fp32 statistics.
 *   y[i,:] = LN(x[src(i),:]) * gamma + beta (+ post_add[i,:]),  src(i) = row_index ? row_index[i] : i
 * Replaces nn.LayerNorm: modeling_finetune.py:127,133 / modeling_adaptation.py:168 (eps 1e-6),
 * clip.py:20-26,152,168 (eps 1e-5, fp32-forced).  mean/rstd (f32 [M]) are saved for backward when given.
 * ------------------------------------------------------------------------------------ */
int unite_layernorm_fwd(const float* x, int32_t ldx, const int32_t* row_index,
                        const float* gamma, const float* beta, float eps,
                        const float* post_add,            /* f32 [M,D] or NULL */
                        void* y, int32_t y_f32,           /* [M,D] bf16 or f32 */
                        float* mean, float* rstd,         /* [M] or NULL */
                        int32_t M, int32_t D, void* stream);
/* The same with a bf16 input matrix (x: bf16 [*, ldx]): the frozen teacher's residual stream when it is kept in bf16
 * (clip.py:60-64 under no_grad; unite_amd.clip UNITE_TEACHER_RES16). */
int unite_layernorm_fwd_bf16in(const void* x, int32_t ldx, const int32_t* row_index,
                               const float* gamma, const float* beta, float eps, const float* post_add,
                               void* y, int32_t y_f32, float* mean, float* rstd, int32_t M, int32_t D, void* stream);
/* ... and with an IEEE-half input matrix (the teacher's f16 residual stream, unite_gemm_args.residual_bf16 == 2). */
int unite_layernorm_fwd_f16in(const void* x, int32_t ldx, const int32_t* row_index,
                              const float* gamma, const float* beta, float eps, const float* post_add,
                              void* y, int32_t y_f32, float* mean, float* rstd, int32_t M, int32_t D, void* stream);

/* Backward of y = LN(x)*gamma+beta.  dy is bf16 (dy_f32 == 0) or f32 [M,D].
 *   dx_out[i,:] = (dx_residual ? dx_residual[i,:] : 0) + dLN/dx
 *   dx_bf16[i,:] = bf16(row_scale[i / rows_per_scale] * dx_out[i,:])      (optional)
 *   dgamma/dbeta (f32 [D]) are OVERWRITTEN or added to (accumulate bit 0), dxsum likewise (accumulate bit 1);
 *   partial sums go through `workspace` (>= unite_layernorm_bwd_workspace(M, D) bytes) -> deterministic. */
size_t unite_layernorm_bwd_workspace(int32_t M, int32_t D);
int unite_layernorm_bwd(const void* dy, int32_t dy_f32, const float* x, int32_t ldx,
                        const float* mean, const float* rstd, const float* gamma,
                        const float* dx_residual, float* dx_out,
                        void* dx_bf16, const float* row_scale, int32_t rows_per_scale,
                        float* dgamma, float* dbeta,
                        float* dxsum,      /* optional f32 [D]: column sums of dx_bf16 = bias gradient of the Linear that consumes it */
                        int32_t accumulate,
                        void* workspace, int32_t M, int32_t D, void* stream);

/* Column sums of a bf16 matrix (bias gradients): out[n] (+)= sum_m x[m,n]; columns in [zero_lo, zero_hi) are written
 * as 0 instead (the k third of the packed (q,0,v) attention bias).  workspace >= unite_colsum_workspace(M, N) bytes; ONE launch:
 * the row block of a 512-column block that finishes last adds the partial rows in order.  The first 4096 bytes of the workspace are
 * arrival counters: ZERO before the first call, left zero by every call. */
size_t unite_colsum_workspace(int32_t M, int32_t N);
int unite_colsum_bf16(const void* x, int32_t ldx, int32_t M, int32_t N, float* out, int32_t accumulate,
                      int32_t zero_lo, int32_t zero_hi, void* workspace, void* stream);

/* ------------------------------------------------------------------------------------
 * Fused multi-head self-attention on a packed qkv matrix [B*N, 3*H*64] (bf16, row = token,
 * columns = [q | k | v] x [head] x [64]); softmax(q k^T * scale) v, head_dim = 64, N <= 320.
 * Replaces modeling_finetune.py:107-116 and nn.MultiheadAttention (clip.py:38,48-53).
 *   out  : bf16 [B*N, H*64];   lse : f32 [B, H, N] (log-sum-exp of the scaled scores, for backward)
 * ------------------------------------------------------------------------------------ */
int unite_attn_fwd(const void* qkv, void* out, float* lse, int32_t B, int32_t N, int32_t H, float scale,
                   void* stream);
/* Backward: dqkv (bf16 [B*N, 3*H*64]) from dout (bf16 [B*N, H*64]); recomputes the probabilities
 * from qkv and lse.  delta (f32 [B,H,N]) is workspace. */
int unite_attn_bwd(const void* qkv, const void* out, const void* dout, const float* lse, float* delta,
                   void* dqkv, int32_t B, int32_t N, int32_t H, float scale, void* stream);
/* Frozen-teacher block front half in one kernel (clip.py:38,48-53 under no_grad): for every (frame, head)
 *   [q | k | v] = h_frame [L x D] . in_proj_weight[head rows]^T + in_proj_bias  (bf16 operands, f32 accumulation, bf16 results)
 *   out_frame[:, head] = softmax(q k^T * scale) v
 * without writing qkv to memory.  h: bf16 [BT*L, D] (the LayerNorm output), w_in: bf16 [3D, D], b_in: f32 [3D], out: bf16 [BT*L, D].
 * 192 < L <= 224 (197 = 14 x 14 patches + CLS), D = 64 H.  Same arithmetic as unite_gemm_bf16 (bias) + unite_attn_fwd. */
int unite_teacher_qkv_attn(const void* h, const void* w_in, const float* b_in, void* out, int32_t BT, int32_t L, int32_t H,
                           int32_t D, float scale, void* stream);

/* Head-averaged softmax probabilities of query row 0 (CLS) over key columns 1..N-1:
 * probs f32 [B, N-1].  Replaces need_weights=True + attn[:,0,1:] (clip.py:51,95-96,183). */
int unite_attn_cls_probs(const void* qkv, float* probs, int32_t B, int32_t N, int32_t H, float scale,
                         void* stream);

/* ------------------------------------------------------------------------------------
 * Patch rows ("im2col") of a (B,3,T,H,W) f32 clip for a Conv3d with kernel = stride = (1,P,P):
 *   cols[i, c*P*P + ph*P + pw] = bf16(video[b, c, t, gh*P+ph, gw*P+pw]),  token = token_index ?
 *   token_index[i] : i,  token = ((b*T + t)*GH + gh)*GW + gw.   Only the listed tokens are read
 * (the reference embeds all 1568 and drops 80 % two lines later: modeling_finetune.py:174,
 * modeling_adaptation.py:153).  P even (16; 14 for CLIP-L/14, clip.py:263-290).  cols rows are ld_cols
 * elements apart (ld_cols >= 3*P*P, % 8 == 0); columns [3*P*P, ld_cols) are written as zeros, so
 * the patch-embed GEMM can run with K padded to a 16-byte multiple (588 -> 592).
 * ------------------------------------------------------------------------------------ */
int unite_im2col_gather(const float* video, const int32_t* token_index, void* cols, int32_t ld_cols,
                        int32_t n_rows, int32_t B, int32_t T, int32_t H, int32_t W, int32_t P, void* stream);

/* Zero-shot CLIP similarities of utils.clip_infer (src/utils.py:55-68): out[b,c] = mean over the T frames of clip b of
 * softmax_c(scale * <img[b*T+t,:], text[c,:]>), both operands L2-normalised f32, scale = 100, n_cls <= 256, C % 4 == 0. */
int unite_clip_similarity(const float* img, const float* text, float* out, int32_t B, int32_t T, int32_t C, int32_t n_cls,
                          float scale, void* stream);

/* Decoded uint8 frames (B,T,H,W,3) -> the f32 (B,3,T,H,W) clip tensor the engines take: ((x / 255) - mean[c]) / std[c] with an
 * optional left-right flip per clip (flip: device uint8[B] or NULL).  Replaces the CPU-side GroupRandomHorizontalFlip + Stack +
 * ToTorchFormatTensor + GroupNormalize + view/transpose of src/datasets/transforms.py:68-96,209-245 and mae.py:218-219 (the
 * reference notes the transpose alone is "80% of the loading time"); same operation order, bit-identical.  mean3 / std3: HOST
 * arrays of 3 floats.  W % 4 == 0. */
int unite_clip_u8_to_f32(const uint8_t* frames, float* out, const uint8_t* flip, const float* mean3, const float* std3, int32_t B,
                         int32_t T, int32_t H, int32_t W, void* stream);

/* Crop + resize of decoded uint8 frames on the device: frames (B,T,H,W,3) -> out (B,T,OH,OW,3), every frame of clip b cropped with
 * the clip's box (x0, y0, w, h) and resized with the arithmetic of Pillow's ``img.resize((OW, OH), Image.BILINEAR)`` on 8-bit images,
 * bit for bit (triangle filter with support max(w / OW, 1), 22-bit fixed-point weights, horizontal then vertical pass, each rounded
 * to uint8): the reference's GroupMultiScaleCrop (src/datasets/transforms.py:136-152; build.py:37) without the PIL workers.
 * boxes_host: HOST int32 [B][4] (the crop is drawn on the host: transforms.py:154-177); boxes up to 7.5 x the output side.
 * workspace >= unite_crop_resize_workspace(B, T, H, OH, OW) bytes. */
size_t unite_crop_resize_workspace(int32_t B, int32_t T, int32_t H, int32_t OH, int32_t OW);
int unite_crop_resize_u8(const uint8_t* frames, const int32_t* boxes_host, uint8_t* out, int32_t B, int32_t T, int32_t H, int32_t W,
                         int32_t OH, int32_t OW, void* workspace, void* stream);

/* ---- stage-2 / stage-3 input path (src/datasets/kinetics_sparse.py:132-281)
 * Validation / test views: Resize(short side, 'bilinear') on decoded uint8 frames is cv2.resize(INTER_LINEAR) in the reference
 * (functional_umt.py:44-66).  frames uint8 [T,H,W,3] -> out uint8 [T,OH,OW,3] by OpenCV's published 8-bit fixed-point algorithm (11-bit
 * weights, the two-shift vertical pass).  cv2 is not in the build image: checked bit for bit against oracle/cv2_resize.py only. */
int unite_resize_u8_linear(const uint8_t* frames, uint8_t* out, int32_t T, int32_t H, int32_t W, int32_t OH, int32_t OW, void* stream);
/* Training clips, everything behind RandAugment in _aug_frame (:232-262): ToTensor, tensor_normalize, random_resized_crop (crop box
 * (i, j, h, w) drawn on the host, then ATen's bilinear interpolate, align_corners = False, to S x S; video_transforms.py:560-592), horizontal flip.
 * frames uint8 [T,H,W,3] (after RandAugment) -> out f32 [3,T,S,S]. */
int unite_train_clip_u8(const uint8_t* frames, float* out, int32_t T, int32_t H, int32_t W, int32_t S, int32_t crop_i, int32_t crop_j,
                        int32_t crop_h, int32_t crop_w, int32_t flip, const float* mean3, const float* std3, void* stream);

/* Bicubic resize of `planes` f32 images H x W -> OH x OW with the semantics of
 * torch.nn.functional.interpolate(mode='bicubic', align_corners=False) (A = -0.75, clamped taps):
 * the teacher-input resize of run_stage1.py:362-368 / run_stage3.py:438-445 (224 -> 196 for CLIP-L/14). */
int unite_resize_bicubic(const float* src, float* dst, int32_t planes, int32_t H, int32_t W,
                         int32_t OH, int32_t OW, void* stream);

/* out[i,:] = table[(index[i] % modulo), :]   (f32, D % 4 == 0) -- sinusoid position rows of the
 * visible tokens (modeling_adaptation.py:144,153,318-319). */
int unite_gather_rows_f32(const float* table, const int32_t* index, int32_t modulo, float* out,
                          int32_t n_rows, int32_t D, void* stream);

/* out[i,:] = table[index[i], :] for bf16 rows (D % 8 == 0): the visible rows of the teacher's last attention output, so
 * that the rest of its last block runs on the 20 % of the tokens whose features are used (clip.py:100-104,168-173). */
int unite_gather_rows_bf16(const void* table, const int32_t* index, void* out, int32_t n_rows, int32_t D, void* stream);

/* CLIP token assembly + ln_pre (clip.py:148-152): patches bf16 [BT*HW, D] ->
 * x [BT*(HW+1), D] = LN([class_embedding ; patches] + positional_embedding), f32 (x_f32 == 1), bf16 (0) or IEEE half (2). */
int unite_clip_embed_ln(const void* patches, const float* class_embedding, const float* positional_embedding,
                        const float* gamma, const float* beta, float eps, void* x, int32_t x_f32,
                        int32_t BT, int32_t HW, int32_t D, void* stream);

/* rows /= ||row||_2  (f32 [M,D], in place; clip.py:172-173). */
int unite_l2_normalize_rows(float* x, int32_t M, int32_t D, void* stream);

/* ------------------------------------------------------------------------------------
 * Attention-guided mask sampling (run_stage1.py:379-387): per frame row of `weights`
 * (f32 [BT, N], N <= 1024) draw a weighted permutation without replacement and keep its first
 * n_vis entries visible.  Implemented as an exponential race (key = -log(u)/w, keep the n_vis
 * smallest): the same distribution as torch.multinomial(w, N)[:, :n_vis] as a SET.
 *   mask    : uint8 [BT*N], 1 = masked          vis_tokens : int32 [BT*n_vis] global token ids
 *   (bt*N + j), ascending -- the row order of x[~mask] (modeling_adaptation.py:153).
 *   vis_rows_cls (optional) : int32 [BT*n_vis] = bt*(N+1) + 1 + j, the same tokens as rows of the
 *   teacher's [BT, 1+N, D] activations (class token first, clip.py:150).
 * ------------------------------------------------------------------------------------ */
int unite_mask_sample(const float* weights, uint64_t seed, uint8_t* mask, int32_t* vis_tokens, int32_t* vis_rows_cls,
                      int32_t BT, int32_t N, int32_t n_vis, void* stream);
/* Stochastic-depth multipliers (timm drop_path, called at modeling_finetune.py:50,143-146; rates from
 * modeling_adaptation.py:93): out[l*per_layer + i] = floor(keep[l] + u) / keep[l], u ~ U[0,1) from a counter-based
 * generator keyed by (seed, element) -- the same two-valued distribution {0, 1/keep} as the reference's
 * x.div(keep) * floor(keep + rand).  keep: device f32 [layers]; out: device f32 [layers*per_layer]. */
int unite_drop_path_scales(const float* keep, uint64_t seed, float* out, int32_t layers, int32_t per_layer, void* stream);
/* Graph-replayable forms: the seed is read from device memory at run time (`seed_dev`, one uint64 the host rewrites between
 * replays), so a captured launch draws fresh masks / keep-vectors every time it is replayed. */
int unite_mask_sample_dev(const float* weights, const uint64_t* seed_dev, uint8_t* mask, int32_t* vis_tokens, int32_t* vis_rows_cls,
                          int32_t BT, int32_t N, int32_t n_vis, void* stream);
int unite_drop_path_scales_dev(const float* keep, const uint64_t* seed_dev, float* out, int32_t layers, int32_t per_layer, void* stream);
/* Same outputs from an explicit permutation (int64 [BT,N], the reference's `importance`). */
int unite_mask_from_importance(const int64_t* importance, uint8_t* mask, int32_t* vis_tokens, int32_t* vis_rows_cls,
                               int32_t BT, int32_t N, int32_t n_vis, void* stream);
/* Same outputs from a caller-supplied mask (uint8 [BT*N], 1 = masked, exactly n_vis zeros per row of N):
 * the index form of x[~mask] (modeling_adaptation.py:153) for masks that did not come from the sampler. */
int unite_mask_to_tokens(const uint8_t* mask, int32_t* vis_tokens, int32_t* vis_rows_cls,
                         int32_t BT, int32_t N, int32_t n_vis, void* stream);

/* Stage-3 committee masks (src/utils.py:89-120, run_stage3.py:497-500): per frame, member i of k keeps the attention ranks
 * i, i+k, ... (n_vis of them; the members are disjoint).  mask uint8 [k, BT, N] (1 = masked); vis_tokens int32 [k][BT*n_vis]:
 * per member, ascending token ids (bt*N + j) inside its own copy of the B target clips (the reference repeats the clips k
 * times, 'k (B T) N -> (k B) (T N)', and runs one batched pass; the members are independent); vis_rows_cls as above. */
int unite_greedy_masks(const float* weights, int32_t k, uint8_t* mask, int32_t* vis_tokens, int32_t* vis_rows_cls,
                       int32_t BT, int32_t N, int32_t n_vis, void* stream);

/* Stage-3 pseudo-label selection (run_stage3.py:488-613) for B target clips with C classes.  strategy: 0 conf, 1 cons,
 * 2 consORconf, 3 consANDconf, 4 clip_only, 5 clip_matchORconf, 6 oracle.  Outputs: pseudo (int64 [B], the student's
 * prediction on the full clip), weight (f32 [B] = selected ? (conf_weighted ? max-softmax-prob : 1) : 0), sel / msp optional. */
int unite_pseudo_label_select(const float* logits_full, const float* logits_masked, int32_t k, const float* clip_probs,
                              const int64_t* labels_t, int32_t strategy, float threshold, float clip_threshold,
                              int32_t conf_weighted, int64_t* pseudo, float* weight, uint8_t* sel, float* msp,
                              int32_t B, int32_t C, void* stream);

/* ------------------------------------------------------------------------------------
 * Decoder tail + UMT loss (modeling_adaptation.py:204-207, run_stage1.py:431):
 *   u = LN_eps(y) * gamma + beta ;  o = u / ||u||_2 ;  loss_sum += sum_rows (2 - 2 <o, tgt>)
 * y f32 [M,C] (C % 4 == 0, C <= 1024); out (f32 [M,C], optional) receives o; loss_sum is ONE f32
 * accumulator the caller zeroes (the caller divides by the row count of all taps).
 * Backward: d loss / d y for loss = loss_scale * (loss_scale_dev ? *loss_scale_dev : 1) *
 * sum_rows(2 - 2<o,tgt>)  (dout == NULL; loss_scale_dev is a device scalar, e.g. autograd's upstream gradient) or for an
 * explicit upstream gradient dout (f32 [M,C]);  dy bf16 [M,C]; dgamma/dbeta as in layernorm_bwd.
 * ------------------------------------------------------------------------------------ */
int unite_decoder_tail_fwd(const float* y, const float* gamma, const float* beta, float eps,
                           const float* tgt, float* out, float* loss_sum, int32_t M, int32_t C, void* stream);
int unite_decoder_tail_bwd(const float* y, const float* gamma, const float* beta, float eps,
                           const float* tgt, float loss_scale, const float* loss_scale_dev, const float* dout,
                           void* dy_bf16, float* dgamma, float* dbeta,
                           float* dysum,   /* optional f32 [C]: column sums of dy_bf16 = bias gradient of the decoder head */
                           int32_t accumulate,
                           void* workspace, int32_t M, int32_t C, void* stream);

/* ------------------------------------------------------------------------------------
 * Fused AdamW over ONE flat f32 parameter buffer (torch.optim.AdamW semantics,
 * optim_factory.py:162-163), with per-chunk hyper-parameter groups (optim_factory.py:76-118):
 *   p *= 1 - lr_g*wd_g ; m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ;
 *   p -= lr_g/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps),   g = grad * grad_scale
 * chunk_group[c] (uint8) names the group of elements [c*1024, (c+1)*1024); lr/wd arrays are HOST
 * arrays of n_groups (<= 64) values, passed by value into the launch.  param_bf16 (optional) gets the
 * bf16 shadow copy the GEMMs read.  grad_scale_dev (optional, device f32[1]) multiplies the gradient
 * (1/loss-scale or the clip coefficient); skip the step entirely if *found_inf_dev != 0 (optional).
 * A group with lr < 0 is left untouched (parameters without a gradient: torch's `p.grad is None`).
 * ------------------------------------------------------------------------------------ */
int unite_adamw_flat(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, void* param_bf16,
                     const uint8_t* chunk_group, int64_t n_elems,
                     const float* lr, const float* weight_decay, int32_t n_groups,
                     float beta1, float beta2, float eps, int32_t step,
                     const float* grad_scale_dev, const int32_t* found_inf_dev, void* stream);
/* Graph-replayable form: learning rates, weight decays and the two bias-correction factors are read from device memory at run
 * time -- hp_dev: f32 [130] = lr[64] | weight_decay[64] | 1 / (1 - beta1^t) | 1 / sqrt(1 - beta2^t), rewritten by the host between
 * replays (run_stage1.py:326-338 writes the schedule into param_groups every step). */
int unite_adamw_flat_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, void* param_bf16,
                         const uint8_t* chunk_group, int64_t n_elems, const float* hp_dev, float beta1, float beta2, float eps,
                         const float* grad_scale_dev, const int32_t* found_inf_dev, void* stream);

/* y[m,:] = bf16(row_scale[m / rows_per_scale] * x[m,:])  (f32 [M,D] -> bf16; the GEMM-operand copy of a gradient that did not
 * come out of a LayerNorm backward, e.g. the token-mean gradient of stage 2). */
int unite_scale_cast_bf16(const float* x, const float* row_scale, int32_t rows_per_scale, void* y, int32_t M, int32_t D, void* stream);

/* f32 -> bf16 cast of a flat buffer (initial shadow copy of the weights). */
int unite_cast_f32_bf16(const float* src, void* dst, int64_t n, void* stream);

/* Global L2 norm of a flat f32 buffer (utils.py:631-643): norm_out[0] = ||g||_2; if max_norm > 0
 * also writes clip_coef_out[0] = min(1, max_norm/(norm+1e-6)) (torch clip_grad_norm_).
 * workspace >= unite_grad_norm_workspace(n) bytes. */
size_t unite_grad_norm_workspace(int64_t n);
int unite_grad_norm_flat(const float* grad, int64_t n, float max_norm, float* norm_out, float* clip_coef_out,
                         void* workspace, void* stream);
/* The same norm over the chunks whose optimizer group is not `skip_group` (chunk_group: uint8 per 1024 elements, the table
 * unite_adamw_flat takes): parameters that have no gradient in this step -- frozen ones, layers that were not executed -- do not
 * enter the norm or the clip coefficient, as torch's `p.grad is not None` filter (utils.py:631-643). */
int unite_grad_norm_flat_masked(const float* grad, int64_t n, const uint8_t* chunk_group, int32_t skip_group, float max_norm,
                                float* norm_out, float* clip_coef_out, void* workspace, void* stream);

/* Token mean over the sequence (stage 2/3 pooling, modeling_finetune.py:376, run_stage3.py:333-338):
 * out[b,:] = mean_n x[b,n,:]  (x f32 [B,N,D], out f32 [B,D]).  Backward: dx[b,n,:] (+)= dout[b,:]/N. */
int unite_token_mean_fwd(const float* x, float* out, int32_t B, int32_t N, int32_t D, void* stream);
int unite_token_mean_bwd(const float* dout, float* dx, int32_t accumulate, int32_t B, int32_t N, int32_t D, void* stream);

/* Small fp32 linear layer (classifier head, modeling_finetune.py:313,382; run_stage3.py src_classifier):
 *   y[b,c] = <x[b,:], W[c,:]> + bias[c];  backward: dx = dy W (optional), dW (+)= dy^T x, db (+)= colsum(dy). */
int unite_linear_f32_fwd(const float* x, const float* W, const float* bias, float* y, int32_t B, int32_t C, int32_t D, void* stream);
int unite_linear_f32_bwd(const float* x, const float* W, const float* dy, float* dx, float* dW, float* db,
                         int32_t B, int32_t C, int32_t D, int32_t accumulate, void* stream);

/* Softmax cross-entropy with optional per-row weights (run_stage2.py:681, run_stage3.py:486,606-612):
 * loss_sum += sum_i w_i * CE(logits[i,:], label[i]) over rows with label >= 0;  dlogits = scale * w_i *
 * (softmax - onehot).  logits f32 [M,C], C <= 1024. */
int unite_softmax_ce(const float* logits, const int64_t* labels, const float* row_weight, float grad_scale,
                     float* loss_sum, float* dlogits, int32_t M, int32_t C, void* stream);

/* The other three clip_loss_type options of stage 1 (run_stage1.py:403-408,431-434: nn.MSELoss / nn.L1Loss / nn.SmoothL1Loss on the
 * normalised decoder outputs; every shipped config uses 'l2', which is fused into unite_decoder_tail_*):
 *   loss_sum += sum_i f(out_i - target_i),  grad_i = grad_scale * f'(out_i - target_i)   (the caller passes grad_scale = 1 / n for the mean)
 * kind 0 = mse (d^2), 1 = l1 (|d|), 2 = smooth_l1 (Huber, beta = 1).  n % 4 == 0, 16-byte aligned pointers; grad may be NULL. */
int unite_pointwise_loss(const float* out, const float* target, int32_t kind, float grad_scale, float* loss_sum, float* grad,
                         int64_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* UNITE_HIP_H */
