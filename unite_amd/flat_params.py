"""Flat parameter / gradient storage in HBM.

All trainable tensors of a model live in ONE fp32 buffer (each tensor padded to a multiple of 1024 elements,
the AdamW kernel's group granule), with a parallel fp32 gradient buffer and a bf16 shadow the MFMA GEMMs read.
``nn.Parameter.data`` / ``.grad`` become views, so ``state_dict()`` / ``load_state_dict()`` keep the reference's
keys and shapes while
  * the optimizer step is one launch over the flat buffer (unite_adamw_flat),
  * the gradient norm is one reduction (unite_grad_norm_flat),
  * data-parallel all-reduce runs on contiguous slices of the flat gradient buffer with no packing copies.
Tensors are laid out in ``named_parameters()`` order = forward order, so backward finishes the buffer from its
end towards its start and buckets are contiguous ranges.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch
import torch.nn as nn

CHUNK = 1024


class FlatParams:
    def __init__(self, module: nn.Module, device: torch.device, with_grad: bool = True):
        named = [(n, p) for n, p in module.named_parameters()]
        self.names: List[str] = [n for n, _ in named]
        self.params: List[nn.Parameter] = [p for _, p in named]
        self.offsets: Dict[str, Tuple[int, int]] = {}
        off = 0
        self.qkv_bias: Dict[str, Tuple[int, int]] = {}      # "<attn prefix>" -> (offset, 3*D) of the packed (q, 0, v) bias
        i = 0
        while i < len(named):
            n, p = named[i]
            if n.endswith("attn.q_bias") and i + 1 < len(named) and named[i + 1][0] == n[:-len("q_bias")] + "v_bias" \
                    and named[i + 1][1].numel() == p.numel():
                # F.linear(x, qkv.weight, cat(q_bias, zeros, v_bias)) (modeling_finetune.py:104-106): keep the three
                # thirds contiguous so the GEMM epilogue reads ONE bias vector; the middle third is never a parameter
                # and stays zero (zero gradient -> AdamW leaves it at zero).
                d = p.numel()
                self.offsets[n] = (off, d)
                self.offsets[named[i + 1][0]] = (off + 2 * d, d)
                self.qkv_bias[n[:-len("q_bias")]] = (off, 3 * d)
                off += (3 * d + CHUNK - 1) // CHUNK * CHUNK
                i += 2
                continue
            self.offsets[n] = (off, p.numel())
            off += (p.numel() + CHUNK - 1) // CHUNK * CHUNK
            i += 1
        self.total = off
        self.device = device
        self.param = torch.zeros(self.total, dtype=torch.float32, device=device)
        self.grad = torch.zeros(self.total if with_grad else 0, dtype=torch.float32, device=device)
        self.shadow = torch.zeros(self.total, dtype=torch.bfloat16, device=device)
        self._bf16: Dict[str, torch.Tensor] = {}
        self._gview: Dict[str, torch.Tensor] = {}
        with torch.no_grad():
            for n, p in named:
                o, k = self.offsets[n]
                view = self.param[o:o + k].view(p.shape)
                view.copy_(p.data)
                p.data = view
                if with_grad:
                    g = self.grad[o:o + k].view(p.shape)
                    p.grad = g
                    self._gview[n] = g
                self._bf16[n] = self.shadow[o:o + k].view(p.shape)
        # True: the next backward ADDS to the gradient buffer (PyTorch semantics after a backward without
        # zero_grad); False: it overwrites (the state after zero_grad).  Avoids a 352 MB memset per step.
        self.accumulate = False
        self._versions = None
        self.sync_shadow()

    # ------------------------------------------------------------------
    def w16(self, name: str) -> torch.Tensor:
        """bf16 shadow of a parameter (GEMM operand)."""
        return self._bf16[name]

    def packed_qkv_bias(self, attn_prefix: str) -> torch.Tensor:
        """f32 [3*D] view (q_bias, zeros, v_bias) for the attention whose parameters start with attn_prefix."""
        o, k = self.qkv_bias[attn_prefix]
        return self.param[o:o + k]

    def packed_qkv_bias_grad(self, attn_prefix: str) -> torch.Tensor:
        """f32 [3*D] gradient view matching packed_qkv_bias (the middle third must stay zero)."""
        o, k = self.qkv_bias[attn_prefix]
        return self.grad[o:o + k]

    def g(self, name: str) -> torch.Tensor:
        """fp32 gradient view of a parameter."""
        return self._gview[name]

    def _version_sum(self) -> int:
        return sum(p._version for p in self.params)

    def sync_shadow(self) -> None:
        from . import ops
        ops.cast_f32_bf16(self.param, self.shadow)
        self._versions = self._version_sum()

    def refresh_if_stale(self) -> None:
        """Re-cast the bf16 shadow if a parameter was modified through PyTorch (load_state_dict, manual edits).
        The fused optimizer updates the shadow itself and does not bump tensor versions."""
        if self._version_sum() != self._versions:
            self.sync_shadow()

    def ensure_grad_views(self) -> None:
        """optimizer.zero_grad(set_to_none=True) drops p.grad; point it back at the flat buffer."""
        for n, p in zip(self.names, self.params):
            if p.grad is not self._gview[n]:
                p.grad = self._gview[n]

    def chunk_groups(self, group_of: Dict[str, int]) -> torch.Tensor:
        """uint8 [total/1024]: optimizer group id of every 1024-element chunk."""
        t = torch.zeros(self.total // CHUNK, dtype=torch.uint8)
        for n in self.names:
            o, k = self.offsets[n]
            t[o // CHUNK:(o + k + CHUNK - 1) // CHUNK] = group_of[n]
        for pre, (o, k) in self.qkv_bias.items():            # q_bias and v_bias always share a group (both 1-D, same layer)
            assert group_of[pre + "q_bias"] == group_of[pre + "v_bias"]
            t[o // CHUNK:(o + k + CHUNK - 1) // CHUNK] = group_of[pre + "q_bias"]
        return t.to(self.device)

    def layer_ranges(self, prefixes: List[str]) -> List[Tuple[int, int]]:
        """[start, end) element range covered by the parameters whose name starts with each prefix."""
        out = []
        for pre in prefixes:
            offs = [self.offsets[n] for n in self.names if n.startswith(pre)]
            out.append((min(o for o, _ in offs) // CHUNK * CHUNK, max((o + k + CHUNK - 1) // CHUNK * CHUNK for o, k in offs)))
        return out
