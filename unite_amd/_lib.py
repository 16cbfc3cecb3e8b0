"""ctypes binding of libunite_hip.so (the C ABI declared in include/unite_hip.h).

The library is built in-tree by ``make -C unite_amd/csrc`` (or ``__graft_entry__.build()``) into
``unite_amd/lib/libunite_hip.so``.  There is NO fallback: if the library is missing the import of any
op raises, so a GPU box can never silently run something else.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libunite_hip.so")
LIB_PATH = os.environ.get("UNITE_HIP_LIB", LIB_PATH)        # A/B runs of two builds of the library in one gpurun call

ABI_VERSION = 2               # UNITE_ABI_VERSION of include/unite_hip.h
WS_HEADER_BYTES = 32768       # UNITE_WS_HEADER_BYTES: arrival counters at the head of a GEMM workspace (zero outside a launch)

c_p = C.c_void_p
c_i = C.c_int32
c_f = C.c_float
c_sz = C.c_size_t
c_i64 = C.c_int64
c_u64 = C.c_uint64


class GemmArgs(C.Structure):
    """struct unite_gemm_args (include/unite_hip.h)."""
    _fields_ = [
        ("M", c_i), ("N", c_i), ("K", c_i),
        ("trans_a", c_i), ("trans_b", c_i),
        ("A", c_p), ("lda", c_i),
        ("B", c_p), ("ldb", c_i),
        ("bias", c_p),
        ("act", c_i),
        ("aux_in", c_p), ("ld_aux_in", c_i),
        ("aux_out", c_p), ("ld_aux_out", c_i),
        ("row_scale", c_p), ("rows_per_scale", c_i),
        ("residual", c_p), ("ldr", c_i),
        ("out", c_p), ("ldc", c_i), ("out_f32", c_i), ("accumulate", c_i),
        ("out_bf16_copy", c_p), ("ld_copy", c_i),
        ("workspace", c_p), ("workspace_bytes", c_i64),
        ("colsum_out", c_p), ("colsum_accumulate", c_i),
        # ABI 2 (zero = process-wide defaults / feature off)
        ("plan_flags", c_i), ("plan_persistent", c_i), ("plan_sharing", c_f),
        ("residual_bf16", c_i),
        ("rowsum_a_out", c_p), ("rowsum_accumulate", c_i), ("rowsum_zero_lo", c_i), ("rowsum_zero_hi", c_i),
    ]


# name -> (restype, argtypes); mirrors include/unite_hip.h one to one
SIGNATURES = {
    "unite_abi_version": (c_i, []),
    "unite_target_arch": (C.c_char_p, []),
    "unite_gemm_bf16": (c_i, [C.POINTER(GemmArgs), c_p]),
    "unite_gemm_set_policy": (c_i, [c_i]),
    "unite_gemm_get_policy": (c_i, []),
    "unite_gemm_set_sharing": (c_i, [C.c_float]),
    "unite_gemm_get_sharing": (C.c_float, []),
    "unite_gemm_plan": (c_i, [c_i, c_i, c_i, c_i, c_i, C.c_float, c_i64, c_i, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "unite_gemm_colsum_workspace": (c_sz, [c_i, c_i]),
    "unite_gemm_bf16_grouped": (c_i, [C.POINTER(GemmArgs), c_i, c_p]),
    "unite_prof_enable": (c_i, [c_i, c_i]),
    "unite_prof_summary": (c_i, [C.POINTER(C.c_double), C.POINTER(c_i64), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "unite_clock_probe": (c_i, [c_p, c_i, c_i, c_p]),
    "unite_clock_stamp": (c_i, [c_p, c_p]),
    "unite_layernorm_fwd": (c_i, [c_p, c_i, c_p, c_p, c_p, c_f, c_p, c_p, c_i, c_p, c_p, c_i, c_i, c_p]),
    "unite_layernorm_fwd_bf16in": (c_i, [c_p, c_i, c_p, c_p, c_p, c_f, c_p, c_p, c_i, c_p, c_p, c_i, c_i, c_p]),
    "unite_layernorm_fwd_f16in": (c_i, [c_p, c_i, c_p, c_p, c_p, c_f, c_p, c_p, c_i, c_p, c_p, c_i, c_i, c_p]),
    "unite_layernorm_bwd_workspace": (c_sz, [c_i, c_i]),
    "unite_layernorm_bwd": (c_i, [c_p, c_i, c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_p, c_p, c_i, c_p, c_i, c_i, c_p]),
    "unite_colsum_workspace": (c_sz, [c_i, c_i]),
    "unite_colsum_bf16": (c_i, [c_p, c_i, c_i, c_i, c_p, c_i, c_i, c_i, c_p, c_p]),
    "unite_attn_fwd": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_f, c_p]),
    "unite_attn_bwd": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_f, c_p]),
    "unite_attn_cls_probs": (c_i, [c_p, c_p, c_i, c_i, c_i, c_f, c_p]),
    "unite_teacher_qkv_attn": (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_p]),
    "unite_clip_similarity": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_p]),
    "unite_clip_u8_to_f32": (c_i, [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p]),
    "unite_crop_resize_workspace": (c_sz, [c_i, c_i, c_i, c_i, c_i]),
    "unite_crop_resize_u8": (c_i, [c_p, C.POINTER(c_i), c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p]),
    "unite_resize_u8_linear": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p]),
    "unite_train_clip_u8": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, C.POINTER(c_f), C.POINTER(c_f), c_p]),
    "unite_resize_bicubic": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p]),
    "unite_im2col_gather": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "unite_gather_rows_bf16": (c_i, [c_p, c_p, c_p, c_i, c_i, c_p]),
    "unite_gather_rows_f32": (c_i, [c_p, c_p, c_i, c_p, c_i, c_i, c_p]),
    "unite_clip_embed_ln": (c_i, [c_p, c_p, c_p, c_p, c_p, c_f, c_p, c_i, c_i, c_i, c_i, c_p]),
    "unite_l2_normalize_rows": (c_i, [c_p, c_i, c_i, c_p]),
    "unite_mask_sample": (c_i, [c_p, c_u64, c_p, c_p, c_p, c_i, c_i, c_i, c_p]),
    "unite_drop_path_scales": (c_i, [c_p, c_u64, c_p, c_i, c_i, c_p]),
    "unite_mask_sample_dev": (c_i, [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_p]),
    "unite_drop_path_scales_dev": (c_i, [c_p, c_p, c_p, c_i, c_i, c_p]),
    "unite_adamw_flat_dev": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_p, c_f, c_f, c_f, c_p, c_p, c_p]),
    "unite_mask_from_importance": (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_p]),
    "unite_mask_to_tokens": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_p]),
    "unite_greedy_masks": (c_i, [c_p, c_i, c_p, c_p, c_p, c_i, c_i, c_i, c_p]),
    "unite_pseudo_label_select": (c_i, [c_p, c_p, c_i, c_p, c_p, c_i, c_f, c_f, c_i, c_p, c_p, c_p, c_p, c_i, c_i, c_p]),
    "unite_decoder_tail_fwd": (c_i, [c_p, c_p, c_p, c_f, c_p, c_p, c_p, c_i, c_i, c_p]),
    "unite_decoder_tail_bwd": (c_i, [c_p, c_p, c_p, c_f, c_p, c_f, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_i, c_i, c_p]),
    "unite_adamw_flat": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_i64, C.POINTER(c_f), C.POINTER(c_f), c_i, c_f, c_f, c_f, c_i, c_p, c_p, c_p]),
    "unite_scale_cast_bf16": (c_i, [c_p, c_p, c_i, c_p, c_i, c_i, c_p]),
    "unite_cast_f32_bf16": (c_i, [c_p, c_p, c_i64, c_p]),
    "unite_grad_norm_workspace": (c_sz, [c_i64]),
    "unite_grad_norm_flat": (c_i, [c_p, c_i64, c_f, c_p, c_p, c_p, c_p]),
    "unite_grad_norm_flat_masked": (c_i, [c_p, c_i64, c_p, c_i, c_f, c_p, c_p, c_p, c_p]),
    "unite_token_mean_fwd": (c_i, [c_p, c_p, c_i, c_i, c_i, c_p]),
    "unite_token_mean_bwd": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_p]),
    "unite_linear_f32_fwd": (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_p]),
    "unite_linear_f32_bwd": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p]),
    "unite_softmax_ce": (c_i, [c_p, c_p, c_p, c_f, c_p, c_p, c_i, c_i, c_p]),
    "unite_pointwise_loss": (c_i, [c_p, c_p, c_i, C.c_float, c_p, c_p, c_i64, c_p]),
}

_lib: Optional[C.CDLL] = None


class UniteHipError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load libunite_hip.so and bind every symbol of the header; raises if the library or a symbol is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise UniteHipError(
            f"{LIB_PATH} not found: build it with `make -C unite_amd/csrc` (or __graft_entry__.build()). "
            "unite_amd has no CPU / eager fallback for its device ops.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    if lib.unite_abi_version() != ABI_VERSION:
        raise UniteHipError("libunite_hip.so ABI version mismatch")
    _lib = lib
    return lib


# ---- the collective side: include/unite_comm.h -> unite_amd/lib/libunite_comm.so (links RCCL; loaded on demand only)
COMM_LIB_PATH = os.path.join(os.path.dirname(LIB_PATH), "libunite_comm.so")
COMM_ID_BYTES = 128
COMM_SIGNATURES = {
    "unite_comm_bind": (c_i, [C.c_char_p]),
    "unite_comm_library": (c_i, [C.c_char_p, c_sz]),
    "unite_comm_unique_id": (c_i, [c_p, c_sz]),
    "unite_comm_init": (c_i, [c_i, c_i, c_p, c_sz]),
    "unite_comm_allreduce_bucket": (c_i, [c_p, c_i64, c_i, c_i, c_p]),
    "unite_comm_broadcast": (c_i, [c_p, c_i64, c_i, c_p]),
    "unite_comm_world": (c_i, []),
    "unite_comm_rank": (c_i, []),
    "unite_comm_destroy": (c_i, []),
}
_comm = None


def load_comm() -> C.CDLL:
    """Load libunite_comm.so (RCCL behind the C ABI of include/unite_comm.h); raises if it is missing."""
    global _comm
    if _comm is not None:
        return _comm
    if not os.path.exists(COMM_LIB_PATH):
        raise UniteHipError(f"{COMM_LIB_PATH} not found: build it with `make -C unite_amd/csrc`")
    lib = C.CDLL(COMM_LIB_PATH)
    for name, (res, args) in COMM_SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    # ONE RCCL per process: bind the copy PyTorch has mapped (or will map: the wheel's lib/librccl.so) instead of the system's
    import torch
    wheel = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    rc = lib.unite_comm_bind(wheel.encode() if os.path.exists(wheel) else None)
    if rc != 0:
        raise UniteHipError("libunite_comm.so found no usable librccl.so (looked at the process's mapped libraries, "
                            f"{wheel}, UNITE_RCCL_LIB and the system search path)")
    _comm = lib
    return lib


def comm_library() -> str:
    """the file the bound RCCL comes from (diagnostic: it must be the one torch.distributed's nccl backend uses)"""
    buf = C.create_string_buffer(1024)
    check_comm(load_comm().unite_comm_library(buf, 1024), "unite_comm_library")
    return buf.value.decode()


def check_comm(code: int, what: str) -> None:
    if code != 0:
        raise UniteHipError(f"{what} failed: " + ("bad argument / communicator state" if code < 0 else f"ncclResult_t {code - 1000}"))


def check(code: int, what: str) -> None:
    if code != 0:
        kind = {-1: "UNITE_EINVAL (bad shape/alignment/null pointer)", -2: "UNITE_ENOSUP (unsupported shape)"}.get(
            code, f"hipError_t {code}")
        raise UniteHipError(f"{what} failed: {kind}")
