"""Tensor-level wrappers over the C ABI (include/unite_hip.h).  PyTorch supplies device memory and the
stream; every device op below is a hand-written gfx950 kernel in libunite_hip.so -- there is no eager /
CPU fallback, a CPU tensor raises."""
from __future__ import annotations

import contextlib
import ctypes as C
import threading
from typing import Optional, Sequence

import torch

from . import _lib

ACT_NONE, ACT_GELU, ACT_QUICKGELU, ACT_DGELU, ACT_GELU_DSAVE, ACT_MULAUX = 0, 1, 2, 3, 4, 5
BF16 = torch.bfloat16
F32 = torch.float32
F16 = torch.float16


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream() -> int:
    """the current HIP stream of the current device as an integer handle (torch.cuda.current_stream() costs ~8 us of Python per call,
    600 calls per step; the raw accessor ~0.3 us)"""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    if not t.is_cuda:
        raise _lib.UniteHipError("unite_amd device ops need CUDA(HIP) tensors; there is no CPU fallback")
    return t.data_ptr()


def _req(t: torch.Tensor, dtype, name: str):
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if t.stride(-1) != 1:
        raise ValueError(f"{name}: last dim must be contiguous")


def _acc_bits(accumulate) -> int:
    """bool -> both outputs groups; int -> bit 0: gamma/beta gradients, bit 1: bias column sums"""
    if isinstance(accumulate, bool):
        return 3 if accumulate else 0
    return int(accumulate)


def _ld(t: torch.Tensor) -> int:
    return t.stride(0) if t.dim() == 2 else t.shape[-1]


class _PlanHints(threading.local):
    """per-thread GEMM planner hints that travel INSIDE each unite_gemm_args (plan_flags / plan_persistent / plan_sharing): no process
    global is written, so two host threads -- or a non-Python binder beside this one -- never see each other's choice"""
    persistent: Optional[int] = None
    sharing: Optional[float] = None
    sched: Optional[int] = None
    epi: Optional[int] = None


_hints = _PlanHints()


@contextlib.contextmanager
def plan(persistent: Optional[int] = None, sharing: Optional[float] = None, sched: Optional[int] = None, epi: Optional[int] = None):
    """GEMM launches enqueued inside the block carry these hints (None: leave as is): ``persistent`` as unite_gemm_set_policy
    (0 never / 1 measured shapes / 2 whenever supported), ``sharing`` as unite_gemm_set_sharing (0 .. 1: how much the launch's CU time
    counts against its latency -- the launches share the GPU with another stream), ``sched`` the tile kernels' main-loop schedule
    (plan_flags bits 2 / 3: 0 reads at the head of each phase, 1 software-pipelined; an A/B switch, the products are bit-identical), ``epi`` the
    256 x 256 kernel's epilogue form where both apply (bits 4 / 5: 0 f32 image in two passes, 1 transposed accumulators + bf16 image; same bits out)."""
    before = (_hints.persistent, _hints.sharing, _hints.sched, _hints.epi)
    if persistent is not None:
        _hints.persistent = int(persistent)
    if sharing is not None:
        _hints.sharing = float(sharing)
    if sched is not None:
        _hints.sched = int(sched)
    if epi is not None:
        _hints.epi = int(epi)
    try:
        yield
    finally:
        _hints.persistent, _hints.sharing, _hints.sched, _hints.epi = before


def keep_plan(ctx) -> None:
    """autograd.Function.forward: remember the hints in force (the backward runs on the autograd engine's own thread, where this
    thread's hints are not visible)"""
    ctx._unite_plan = (_hints.persistent, _hints.sharing)


@contextlib.contextmanager
def kept_plan(ctx):
    """autograd.Function.backward: the launches of the backward carry the hints its forward ran under"""
    before = (_hints.persistent, _hints.sharing)
    _hints.persistent, _hints.sharing = getattr(ctx, "_unite_plan", before)
    try:
        yield
    finally:
        _hints.persistent, _hints.sharing = before


def prepare_workspace(workspace: torch.Tensor) -> torch.Tensor:
    """GEMM workspaces start with UNITE_WS_HEADER_BYTES of arrival counters that must be zero before the first launch (every launch
    leaves them zero): zeroed once per tensor object."""
    if not getattr(workspace, "_unite_ws_ready", False):
        workspace.view(torch.uint8).view(-1)[:min(_lib.WS_HEADER_BYTES, workspace.numel() * workspace.element_size())].zero_()
        workspace._unite_ws_ready = True
    return workspace


def gemm(a: torch.Tensor, b: torch.Tensor, out: torch.Tensor, *, trans_a: bool = False, trans_b: bool = False,
         bias: Optional[torch.Tensor] = None, act: int = ACT_NONE, aux_in: Optional[torch.Tensor] = None,
         aux_out: Optional[torch.Tensor] = None, row_scale: Optional[torch.Tensor] = None, rows_per_scale: int = 1,
         residual: Optional[torch.Tensor] = None, accumulate: bool = False,
         out_bf16_copy: Optional[torch.Tensor] = None, workspace: Optional[torch.Tensor] = None,
         colsum_out: Optional[torch.Tensor] = None, colsum_accumulate: bool = False,
         rowsum_out: Optional[torch.Tensor] = None, rowsum_accumulate: bool = False, rowsum_zero_range=(0, 0)) -> torch.Tensor:
    """out[M,N] = epilogue(op(a) @ op(b)); a: [M,K] (or [K,M] if trans_a), b: [N,K] (or [K,N] if trans_b); 2-D views.
    residual: f32 or bf16 [M,N].  colsum_out (f32 [N]) (+)= column sums of the stored out (needs workspace >= gemm_colsum_workspace(M, N)
    bytes).  rowsum_out (f32 [M]) (+)= row sums of op(a) (the bias gradient when the product is a weight gradient dY^T X); rows in
    rowsum_zero_range are written as zeros."""
    lib = _lib.load()
    g = _gemm_args(a, b, out, trans_a, trans_b, bias, act, aux_in, aux_out, row_scale, rows_per_scale, residual, accumulate,
                   out_bf16_copy, workspace)
    if rowsum_out is not None:
        _req(rowsum_out, F32, "rowsum_out")
        if rowsum_out.numel() != out.shape[0]:
            raise ValueError("rowsum_out must have M elements")
        g.rowsum_a_out, g.rowsum_accumulate = _ptr(rowsum_out), int(rowsum_accumulate)
        g.rowsum_zero_lo, g.rowsum_zero_hi = int(rowsum_zero_range[0]), int(rowsum_zero_range[1])
    if colsum_out is not None:
        _req(colsum_out, F32, "colsum_out")
        if colsum_out.numel() != out.shape[1]:
            raise ValueError("colsum_out must have N elements")
        g.colsum_out, g.colsum_accumulate = _ptr(colsum_out), int(colsum_accumulate)
    _lib.check(lib.unite_gemm_bf16(C.byref(g), _stream()), "unite_gemm_bf16")
    return out


def gemm_colsum_workspace(M: int, N: int) -> int:
    return int(_lib.load().unite_gemm_colsum_workspace(M, N))


def gemm_grouped(problems):
    """problems: up to 4 tuples (a, b, out, kwargs-of-gemm) with equal trans_a / trans_b -> one launch (unite_gemm_bf16_grouped)."""
    lib = _lib.load()
    arr = (_lib.GemmArgs * len(problems))()
    for i, (a, b, out, kw) in enumerate(problems):
        kw = dict(kw)
        arr[i] = _gemm_args(a, b, out, kw.pop("trans_a", False), kw.pop("trans_b", False), kw.pop("bias", None), kw.pop("act", ACT_NONE),
                            kw.pop("aux_in", None), kw.pop("aux_out", None), kw.pop("row_scale", None), kw.pop("rows_per_scale", 1),
                            kw.pop("residual", None), kw.pop("accumulate", False), kw.pop("out_bf16_copy", None), None)
        if kw:
            raise TypeError(f"unknown gemm arguments {sorted(kw)}")
    _lib.check(lib.unite_gemm_bf16_grouped(arr, len(problems), _stream()), "unite_gemm_bf16_grouped")


def _gemm_args(a, b, out, trans_a, trans_b, bias, act, aux_in, aux_out, row_scale, rows_per_scale, residual, accumulate,
               out_bf16_copy, workspace):
    _req(a, BF16, "a"); _req(b, BF16, "b")
    M, N = out.shape
    K = a.shape[0] if trans_a else a.shape[1]
    am = a.shape[1] if trans_a else a.shape[0]
    bk = b.shape[0] if trans_b else b.shape[1]
    bn = b.shape[1] if trans_b else b.shape[0]
    if am != M or bn != N or bk != K:
        raise ValueError(f"gemm shape mismatch: a{tuple(a.shape)} b{tuple(b.shape)} out{tuple(out.shape)} ta={trans_a} tb={trans_b}")
    g = _lib.GemmArgs()
    g.M, g.N, g.K = M, N, K
    g.trans_a, g.trans_b = int(trans_a), int(trans_b)
    g.A, g.lda = _ptr(a), a.stride(0)
    g.B, g.ldb = _ptr(b), b.stride(0)
    if bias is not None:
        _req(bias, F32, "bias")
    g.bias = _ptr(bias)
    g.act = act
    g.aux_in, g.ld_aux_in = _ptr(aux_in), (aux_in.stride(0) if aux_in is not None else 0)
    g.aux_out, g.ld_aux_out = _ptr(aux_out), (aux_out.stride(0) if aux_out is not None else 0)
    g.row_scale, g.rows_per_scale = _ptr(row_scale), rows_per_scale
    if residual is not None and residual.dtype not in (F32, BF16, F16):
        raise TypeError("residual must be f32, bf16 or f16")
    g.residual, g.ldr = _ptr(residual), (residual.stride(0) if residual is not None else 0)
    g.residual_bf16 = 0 if residual is None else 2 if residual.dtype == F16 else int(residual.dtype == BF16)
    if (out.dtype == F16) != (g.residual_bf16 == 2):
        raise ValueError("an f16 output goes with an f16 residual (the teacher's f16 residual stream) and nothing else")
    if out.dtype not in (BF16, F32, F16):
        raise TypeError("out must be bf16, f32 or (with an f16 residual) f16")
    g.out, g.ldc, g.out_f32, g.accumulate = _ptr(out), out.stride(0), int(out.dtype == F32), int(accumulate)
    g.out_bf16_copy, g.ld_copy = _ptr(out_bf16_copy), (out_bf16_copy.stride(0) if out_bf16_copy is not None else 0)
    if workspace is not None:
        prepare_workspace(workspace)
    g.workspace, g.workspace_bytes = _ptr(workspace), (workspace.numel() * workspace.element_size() if workspace is not None else 0)
    if _hints.persistent is not None:
        g.plan_flags |= 1
        g.plan_persistent = _hints.persistent
    if _hints.sharing is not None:
        g.plan_flags |= 2
        g.plan_sharing = _hints.sharing
    if _hints.sched is not None:
        g.plan_flags |= 4 | (8 if _hints.sched else 0)
    if _hints.epi is not None:
        g.plan_flags |= 16 | (32 if _hints.epi else 0)
    return g


def layernorm_fwd(x: torch.Tensor, gamma, beta, eps: float, y: torch.Tensor, *, row_index=None, post_add=None,
                  mean=None, rstd=None) -> torch.Tensor:
    lib = _lib.load()
    M, D = y.shape
    if x.dtype == F16:           # the teacher's f16 residual stream
        _lib.check(lib.unite_layernorm_fwd_f16in(_ptr(x), x.stride(0), _ptr(row_index), _ptr(gamma), _ptr(beta), eps, _ptr(post_add),
                                                 _ptr(y), int(y.dtype == F32), _ptr(mean), _ptr(rstd), M, D, _stream()),
                   "unite_layernorm_fwd_f16in")
        return y
    if x.dtype == BF16:          # the teacher's bf16 residual stream
        _req(x, BF16, "x")
        _lib.check(lib.unite_layernorm_fwd_bf16in(_ptr(x), x.stride(0), _ptr(row_index), _ptr(gamma), _ptr(beta), eps, _ptr(post_add),
                                                  _ptr(y), int(y.dtype == F32), _ptr(mean), _ptr(rstd), M, D, _stream()),
                   "unite_layernorm_fwd_bf16in")
        return y
    _req(x, F32, "x")
    _lib.check(lib.unite_layernorm_fwd(_ptr(x), x.stride(0), _ptr(row_index), _ptr(gamma), _ptr(beta), eps, _ptr(post_add),
                                       _ptr(y), int(y.dtype == F32), _ptr(mean), _ptr(rstd), M, D, _stream()),
               "unite_layernorm_fwd")
    return y


def layernorm_bwd_workspace(M: int, D: int) -> int:
    return int(_lib.load().unite_layernorm_bwd_workspace(M, D))


def layernorm_bwd(dy, x, mean, rstd, gamma, *, dx_residual=None, dx_out=None, dx_bf16=None, row_scale=None,
                  rows_per_scale: int = 1, dgamma=None, dbeta=None, dxsum=None, accumulate: bool = False,
                  workspace: torch.Tensor = None):
    lib = _lib.load()
    M, D = x.shape
    _lib.check(lib.unite_layernorm_bwd(_ptr(dy), int(dy.dtype == F32), _ptr(x), x.stride(0), _ptr(mean), _ptr(rstd), _ptr(gamma),
                                       _ptr(dx_residual), _ptr(dx_out), _ptr(dx_bf16), _ptr(row_scale), rows_per_scale,
                                       _ptr(dgamma), _ptr(dbeta), _ptr(dxsum), _acc_bits(accumulate), _ptr(workspace), M, D, _stream()),
               "unite_layernorm_bwd")


def colsum_workspace(M: int, N: int) -> int:
    return int(_lib.load().unite_colsum_workspace(M, N))


def colsum(x: torch.Tensor, out: torch.Tensor, workspace: torch.Tensor, accumulate: bool = False, zero_range=(0, 0)):
    lib = _lib.load()
    _req(x, BF16, "x")
    M, N = x.shape
    prepare_workspace(workspace)
    _lib.check(lib.unite_colsum_bf16(_ptr(x), x.stride(0), M, N, _ptr(out), int(accumulate), zero_range[0], zero_range[1],
                                     _ptr(workspace), _stream()),
               "unite_colsum_bf16")
    return out


def attn_fwd(qkv, out, lse, B: int, N: int, H: int, scale: float):
    lib = _lib.load()
    _req(qkv, BF16, "qkv")
    assert qkv.is_contiguous() and out.is_contiguous() and qkv.shape[-1] == 3 * H * 64
    _lib.check(lib.unite_attn_fwd(_ptr(qkv), _ptr(out), _ptr(lse), B, N, H, scale, _stream()), "unite_attn_fwd")
    return out


def attn_bwd(qkv, out, dout, lse, delta, dqkv, B: int, N: int, H: int, scale: float):
    lib = _lib.load()
    assert qkv.is_contiguous() and out.is_contiguous() and dout.is_contiguous() and dqkv.is_contiguous()
    _lib.check(lib.unite_attn_bwd(_ptr(qkv), _ptr(out), _ptr(dout), _ptr(lse), _ptr(delta), _ptr(dqkv), B, N, H, scale, _stream()),
               "unite_attn_bwd")
    return dqkv


def attn_cls_probs(qkv, probs, B: int, N: int, H: int, scale: float):
    lib = _lib.load()
    _lib.check(lib.unite_attn_cls_probs(_ptr(qkv), _ptr(probs), B, N, H, scale, _stream()), "unite_attn_cls_probs")
    return probs


def im2col_gather(video, token_index, cols, P: int):
    lib = _lib.load()
    _req(video, F32, "video")
    assert video.is_contiguous()
    B, Cc, T, H, W = video.shape
    assert Cc == 3
    _lib.check(lib.unite_im2col_gather(_ptr(video), _ptr(token_index), _ptr(cols), cols.stride(0), cols.shape[0], B, T, H, W, P, _stream()),
               "unite_im2col_gather")
    return cols


def teacher_qkv_attn(h, w_in, b_in, out, BT: int, L: int, H: int, scale: float):
    """fused QKV projection + attention of the frozen teacher: h bf16 [BT*L, D] -> out bf16 [BT*L, D] (no qkv round trip)"""
    lib = _lib.load()
    _req(h, BF16, "h"); _req(w_in, BF16, "w_in"); _req(b_in, F32, "b_in"); _req(out, BF16, "out")
    D = h.shape[1]
    assert h.is_contiguous() and out.is_contiguous() and w_in.is_contiguous() and tuple(w_in.shape) == (3 * D, D) and h.shape[0] == BT * L
    _lib.check(lib.unite_teacher_qkv_attn(_ptr(h), _ptr(w_in), _ptr(b_in), _ptr(out), BT, L, H, D, scale, _stream()), "unite_teacher_qkv_attn")
    return out


def clip_similarity(img, text, out, T: int, scale: float = 100.0):
    lib = _lib.load()
    _req(img, F32, "img"); _req(text, F32, "text"); _req(out, F32, "out")
    B, n_cls = out.shape
    assert img.shape[0] == B * T and img.shape[1] == text.shape[1] and text.shape[0] == n_cls and img.is_contiguous() and text.is_contiguous()
    _lib.check(lib.unite_clip_similarity(_ptr(img), _ptr(text), _ptr(out), B, T, img.shape[1], n_cls, scale, _stream()), "unite_clip_similarity")
    return out


def clip_u8_to_f32(frames, out, mean, std, flip=None):
    """frames uint8 (B,T,H,W,3) -> out f32 (B,3,T,H,W), normalised (and flipped where flip[b] != 0)."""
    lib = _lib.load()
    _req(frames, torch.uint8, "frames")
    _req(out, F32, "out")
    B, T, H, W, Cc = frames.shape
    assert Cc == 3 and frames.is_contiguous() and out.is_contiguous() and tuple(out.shape) == (B, 3, T, H, W)
    if flip is not None:
        _req(flip, torch.uint8, "flip")
    m3, s3 = (C.c_float * 3)(*[float(v) for v in mean]), (C.c_float * 3)(*[float(v) for v in std])
    _lib.check(lib.unite_clip_u8_to_f32(_ptr(frames), _ptr(out), _ptr(flip), m3, s3, B, T, H, W, _stream()), "unite_clip_u8_to_f32")
    return out


def resize_u8_linear(frames, out):
    """frames uint8 (T,H,W,3) -> out uint8 (T,OH,OW,3): OpenCV's 8-bit INTER_LINEAR resize (the reference's validation / test Resize)"""
    _req(frames, torch.uint8, "frames")
    _req(out, torch.uint8, "out")
    T, H, W, Cc = frames.shape
    assert Cc == 3 and frames.is_contiguous() and out.is_contiguous() and out.shape[0] == T and out.shape[3] == 3
    _lib.check(_lib.load().unite_resize_u8_linear(_ptr(frames), _ptr(out), T, H, W, out.shape[1], out.shape[2], _stream()), "unite_resize_u8_linear")
    return out


def train_clip_u8(frames, out, box, flip, mean, std):
    """frames uint8 (T,H,W,3) -> out f32 (3,T,S,S): / 255, normalise, crop box (i, j, h, w), bilinear resize (ATen, align_corners False), flip"""
    _req(frames, torch.uint8, "frames")
    _req(out, F32, "out")
    T, H, W, Cc = frames.shape
    S = out.shape[-1]
    assert Cc == 3 and frames.is_contiguous() and out.is_contiguous() and tuple(out.shape) == (3, T, S, S)
    m3, s3 = (C.c_float * 3)(*[float(v) for v in mean]), (C.c_float * 3)(*[float(v) for v in std])
    i, j, h, w = (int(v) for v in box)
    _lib.check(_lib.load().unite_train_clip_u8(_ptr(frames), _ptr(out), T, H, W, S, i, j, h, w, int(bool(flip)), m3, s3, _stream()),
               "unite_train_clip_u8")
    return out


def crop_resize_workspace(B: int, T: int, H: int, OH: int, OW: int) -> int:
    return int(_lib.load().unite_crop_resize_workspace(B, T, H, OH, OW))


def crop_resize_u8(frames, boxes, out, workspace):
    """frames uint8 (B,T,H,W,3) on the device, boxes: B host tuples (x0, y0, w, h), out uint8 (B,T,OH,OW,3): per-clip crop +
    Pillow-bilinear resize (unite_crop_resize_u8)"""
    lib = _lib.load()
    _req(frames, torch.uint8, "frames"); _req(out, torch.uint8, "out")
    B, T, H, W, Cc = frames.shape
    OH, OW = out.shape[2], out.shape[3]
    assert Cc == 3 and frames.is_contiguous() and out.is_contiguous() and tuple(out.shape) == (B, T, OH, OW, 3) and len(boxes) == B
    flat = [int(v) for box in boxes for v in box]
    arr = (C.c_int32 * (4 * B))(*flat)
    _lib.check(lib.unite_crop_resize_u8(_ptr(frames), arr, _ptr(out), B, T, H, W, OH, OW, _ptr(workspace), _stream()), "unite_crop_resize_u8")
    return out


def resize_bicubic(video, out):
    """(B,C,T,H,W) f32 -> out (B,C,T,OH,OW): per-plane bicubic resize, align_corners=False."""
    lib = _lib.load()
    _req(video, F32, "video")
    _req(out, F32, "out")
    assert video.is_contiguous() and out.is_contiguous() and video.shape[:-2] == out.shape[:-2]
    H, W = video.shape[-2:]
    OH, OW = out.shape[-2:]
    planes = video.numel() // (H * W)
    for lo in range(0, planes, 32768):                          # grid.z limit
        n = min(32768, planes - lo)
        _lib.check(lib.unite_resize_bicubic(video.data_ptr() + lo * H * W * 4, out.data_ptr() + lo * OH * OW * 4, n, H, W, OH, OW, _stream()),
                   "unite_resize_bicubic")
    return out


def gather_rows(table, index, out, modulo: int = 0):
    lib = _lib.load()
    if table.dtype in (BF16, F16):          # a copy of 16-bit rows either way
        _req(out, table.dtype, "out")
        assert modulo == 0 and index is not None
        _lib.check(lib.unite_gather_rows_bf16(_ptr(table), _ptr(index), _ptr(out), out.shape[0], out.shape[1], _stream()), "unite_gather_rows_bf16")
        return out
    _lib.check(lib.unite_gather_rows_f32(_ptr(table), _ptr(index), modulo, _ptr(out), out.shape[0], out.shape[1], _stream()),
               "unite_gather_rows_f32")
    return out


def clip_embed_ln(patches, cls, pos, gamma, beta, eps: float, x, BT: int, HW: int, D: int):
    lib = _lib.load()
    _lib.check(lib.unite_clip_embed_ln(_ptr(patches), _ptr(cls), _ptr(pos), _ptr(gamma), _ptr(beta), eps, _ptr(x), 1 if x.dtype == F32 else 2 if x.dtype == F16 else 0, BT, HW, D,
                                       _stream()),
               "unite_clip_embed_ln")
    return x


def l2_normalize_rows(x):
    lib = _lib.load()
    _lib.check(lib.unite_l2_normalize_rows(_ptr(x), x.shape[0], x.shape[1], _stream()), "unite_l2_normalize_rows")
    return x


def mask_sample(weights, seed: int, mask, vis_tokens, n_vis: int, vis_rows_cls=None, seed_dev=None):
    """seed_dev: optional device int64 [1] holding the seed (graph-replayable form); `seed` is ignored then"""
    lib = _lib.load()
    BT, N = weights.shape
    if seed_dev is not None:
        _lib.check(lib.unite_mask_sample_dev(_ptr(weights), _ptr(seed_dev), _ptr(mask), _ptr(vis_tokens), _ptr(vis_rows_cls), BT, N, n_vis,
                                             _stream()), "unite_mask_sample_dev")
        return
    _lib.check(lib.unite_mask_sample(_ptr(weights), seed & 0xFFFFFFFFFFFFFFFF, _ptr(mask), _ptr(vis_tokens), _ptr(vis_rows_cls),
                                     BT, N, n_vis, _stream()), "unite_mask_sample")


def drop_path_scales(keep, seed: int, out, seed_dev=None):
    """out (layers, ...) f32 <- floor(keep[l] + u) / keep[l] (timm drop_path multipliers); keep: device f32 [layers].
    seed_dev: optional device int64 [1] holding the seed (graph-replayable form)."""
    lib = _lib.load()
    _req(keep, F32, "keep"); _req(out, F32, "out")
    layers = keep.numel()
    assert out.is_contiguous() and out.numel() % layers == 0
    if seed_dev is not None:
        _lib.check(lib.unite_drop_path_scales_dev(_ptr(keep), _ptr(seed_dev), _ptr(out), layers, out.numel() // layers, _stream()),
                   "unite_drop_path_scales_dev")
        return out
    _lib.check(lib.unite_drop_path_scales(_ptr(keep), seed & 0xFFFFFFFFFFFFFFFF, _ptr(out), layers, out.numel() // layers, _stream()),
               "unite_drop_path_scales")
    return out


def mask_from_importance(importance, mask, vis_tokens, n_vis: int, vis_rows_cls=None):
    lib = _lib.load()
    BT, N = importance.shape
    assert importance.dtype == torch.int64 and importance.is_contiguous()
    _lib.check(lib.unite_mask_from_importance(_ptr(importance), _ptr(mask), _ptr(vis_tokens), _ptr(vis_rows_cls), BT, N, n_vis,
                                              _stream()), "unite_mask_from_importance")


def mask_to_tokens(mask_u8, vis_tokens, n_vis: int, BT: int, N: int, vis_rows_cls=None):
    lib = _lib.load()
    assert mask_u8.dtype == torch.uint8 and mask_u8.is_contiguous()
    _lib.check(lib.unite_mask_to_tokens(_ptr(mask_u8), _ptr(vis_tokens), _ptr(vis_rows_cls), BT, N, n_vis, _stream()),
               "unite_mask_to_tokens")


def decoder_tail_fwd(y, gamma, beta, eps: float, tgt, out, loss_sum):
    lib = _lib.load()
    M, Cd = y.shape
    _lib.check(lib.unite_decoder_tail_fwd(_ptr(y), _ptr(gamma), _ptr(beta), eps, _ptr(tgt), _ptr(out), _ptr(loss_sum), M, Cd, _stream()),
               "unite_decoder_tail_fwd")


def decoder_tail_bwd(y, gamma, beta, eps: float, tgt, loss_scale: float, dout, dy_bf16, dgamma, dbeta, workspace,
                     accumulate: bool = False, loss_scale_dev=None, dysum=None):
    lib = _lib.load()
    M, Cd = y.shape
    _lib.check(lib.unite_decoder_tail_bwd(_ptr(y), _ptr(gamma), _ptr(beta), eps, _ptr(tgt), loss_scale, _ptr(loss_scale_dev), _ptr(dout), _ptr(dy_bf16),
                                          _ptr(dgamma), _ptr(dbeta), _ptr(dysum), _acc_bits(accumulate), _ptr(workspace), M, Cd, _stream()),
               "unite_decoder_tail_bwd")


def adamw_flat(param, grad, exp_avg, exp_avg_sq, param_bf16, chunk_group, lrs: Sequence[float], wds: Sequence[float],
               beta1: float, beta2: float, eps: float, step: int, grad_scale=None, found_inf=None):
    lib = _lib.load()
    n = len(lrs)
    lr_arr = (C.c_float * n)(*lrs)
    wd_arr = (C.c_float * n)(*wds)
    _lib.check(lib.unite_adamw_flat(_ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq), _ptr(param_bf16), _ptr(chunk_group),
                                    param.numel(), lr_arr, wd_arr, n, beta1, beta2, eps, step, _ptr(grad_scale), _ptr(found_inf),
                                    _stream()), "unite_adamw_flat")


def adamw_flat_dev(param, grad, exp_avg, exp_avg_sq, param_bf16, chunk_group, hp_dev, beta1: float, beta2: float, eps: float,
                   grad_scale=None, found_inf=None):
    """AdamW with lr / weight decay / bias corrections read from device memory (hp_dev f32 [130]): graph-replayable"""
    lib = _lib.load()
    _req(hp_dev, F32, "hp_dev")
    assert hp_dev.numel() >= 130
    _lib.check(lib.unite_adamw_flat_dev(_ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq), _ptr(param_bf16), _ptr(chunk_group),
                                        param.numel(), _ptr(hp_dev), beta1, beta2, eps, _ptr(grad_scale), _ptr(found_inf), _stream()),
               "unite_adamw_flat_dev")


def cast_f32_bf16(src, dst):
    lib = _lib.load()
    _lib.check(lib.unite_cast_f32_bf16(_ptr(src), _ptr(dst), src.numel(), _stream()), "unite_cast_f32_bf16")
    return dst


def grad_norm_workspace(n: int) -> int:
    return int(_lib.load().unite_grad_norm_workspace(n))


def grad_norm_flat(grad, norm_out, workspace, max_norm: float = 0.0, clip_coef_out=None, chunk_group=None, skip_group: int = -1):
    """chunk_group / skip_group: leave out the 1024-element chunks of that optimizer group (parameters without a gradient this step)"""
    lib = _lib.load()
    if chunk_group is not None and skip_group >= 0:
        _lib.check(lib.unite_grad_norm_flat_masked(_ptr(grad), grad.numel(), _ptr(chunk_group), skip_group, max_norm, _ptr(norm_out),
                                                   _ptr(clip_coef_out), _ptr(workspace), _stream()), "unite_grad_norm_flat_masked")
        return norm_out
    _lib.check(lib.unite_grad_norm_flat(_ptr(grad), grad.numel(), max_norm, _ptr(norm_out), _ptr(clip_coef_out), _ptr(workspace), _stream()),
               "unite_grad_norm_flat")
    return norm_out


def token_mean_fwd(x, out):
    lib = _lib.load()
    B, N, D = x.shape
    _lib.check(lib.unite_token_mean_fwd(_ptr(x), _ptr(out), B, N, D, _stream()), "unite_token_mean_fwd")
    return out


def token_mean_bwd(dout, dx, accumulate: bool = False):
    lib = _lib.load()
    B, N, D = dx.shape
    _lib.check(lib.unite_token_mean_bwd(_ptr(dout), _ptr(dx), int(accumulate), B, N, D, _stream()), "unite_token_mean_bwd")
    return dx


def softmax_ce(logits, labels, loss_sum, dlogits=None, row_weight=None, grad_scale: float = 1.0):
    lib = _lib.load()
    M, Cc = logits.shape
    _lib.check(lib.unite_softmax_ce(_ptr(logits), _ptr(labels), _ptr(row_weight), grad_scale, _ptr(loss_sum), _ptr(dlogits), M, Cc, _stream()),
               "unite_softmax_ce")


POINTWISE_LOSS = {"mse": 0, "l1": 1, "smooth_l1": 2}


def pointwise_loss(out, target, kind: str, loss_sum, grad=None, grad_scale: float = 1.0):
    """loss_sum += sum f(out - target), grad = grad_scale * f'(out - target); kind in mse / l1 / smooth_l1 (run_stage1.py:403-408)."""
    lib = _lib.load()
    _req(out, F32, "out"); _req(target, F32, "target")
    if out.numel() != target.numel() or not out.is_contiguous() or not target.is_contiguous():
        raise ValueError("pointwise_loss needs contiguous tensors of equal size")
    _lib.check(lib.unite_pointwise_loss(_ptr(out), _ptr(target), POINTWISE_LOSS[kind], grad_scale, _ptr(loss_sum), _ptr(grad), out.numel(),
                                        _stream()), "unite_pointwise_loss")


def linear_f32_fwd(x, W, bias, y):
    lib = _lib.load()
    B, D = x.shape
    _lib.check(lib.unite_linear_f32_fwd(_ptr(x), _ptr(W), _ptr(bias), _ptr(y), B, W.shape[0], D, _stream()), "unite_linear_f32_fwd")
    return y


def linear_f32_bwd(x, W, dy, dx=None, dW=None, db=None, accumulate: bool = False):
    lib = _lib.load()
    B, D = x.shape
    _lib.check(lib.unite_linear_f32_bwd(_ptr(x), _ptr(W), _ptr(dy), _ptr(dx), _ptr(dW), _ptr(db), B, W.shape[0], D, int(accumulate), _stream()),
               "unite_linear_f32_bwd")


def scale_cast_colsum(x, y_bf16, colsum_out, workspace, row_scale=None, rows_per_scale: int = 1, accumulate: bool = False):
    """y = bf16(row_scale * x) and colsum_out (+)= column sums of y (the bias gradient of the Linear that consumes y)."""
    lib = _lib.load()
    M, D = x.shape
    _lib.check(lib.unite_scale_cast_bf16(_ptr(x), _ptr(row_scale), rows_per_scale, _ptr(y_bf16), M, D, _stream()), "unite_scale_cast_bf16")
    if colsum_out is not None:
        colsum(y_bf16, colsum_out, workspace, accumulate=accumulate)
    return y_bf16


def greedy_masks(weights, k: int, mask, vis_tokens, n_vis: int, vis_rows_cls=None):
    lib = _lib.load()
    BT, N = weights.shape
    _lib.check(lib.unite_greedy_masks(_ptr(weights), k, _ptr(mask), _ptr(vis_tokens), _ptr(vis_rows_cls), BT, N, n_vis, _stream()),
               "unite_greedy_masks")


SELECTION = {"conf": 0, "cons": 1, "consORconf": 2, "consANDconf": 3, "clip_only": 4, "clip_matchORconf": 5, "oracle": 6}


def pseudo_label_select(logits_full, logits_masked, strategy: str, threshold: float, clip_threshold: float, conf_weighted: bool,
                        pseudo, weight, clip_probs=None, labels_t=None, sel=None, msp=None):
    lib = _lib.load()
    k, B, Cc = logits_masked.shape
    _lib.check(lib.unite_pseudo_label_select(_ptr(logits_full), _ptr(logits_masked), k, _ptr(clip_probs), _ptr(labels_t), SELECTION[strategy],
                                             threshold, clip_threshold, int(conf_weighted), _ptr(pseudo), _ptr(weight), _ptr(sel), _ptr(msp),
                                             B, Cc, _stream()), "unite_pseudo_label_select")
