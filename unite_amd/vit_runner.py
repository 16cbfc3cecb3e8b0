"""Hand-scheduled forward / backward of the video ViT on the C-ABI kernels (no autograd tape, no ATen compute).

One ``ViTRunner`` drives the encoder of both student-style models:
  * stage 1/3 student: AdaptationVisionTransformer  (reference src/models/modeling_adaptation.py:131-179,304-334)
  * stage 2 classifier: VisionTransformer            (reference src/models/modeling_finetune.py:356-383)
Per block (reference Block.forward, modeling_finetune.py:143-146) the launches are
  LN1 -> GEMM qkv(+q/v bias) -> fused attention -> GEMM proj(+bias, drop-path scale, +residual)
  LN2 -> GEMM fc1(+bias, GELU, keeps pre-activation) -> GEMM fc2(+bias, drop-path scale, +residual)
and the backward mirrors them with dgrad (NN) / wgrad (TN) GEMMs writing fp32 gradients straight into the flat
gradient buffer.  The residual stream and its gradient stay fp32; GEMM operands are bf16.
Activations needed by the backward are kept in HBM (about 283 MB per block at B=32: 288 GB makes
recomputation pointless).
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Sequence

import torch

from . import ops
from .flat_params import FlatParams

BF16, F32 = torch.bfloat16, torch.float32
# workspace of the weight-gradient GEMMs: arrival counters + per-slice bias row sums + up to 16 split-K slabs of a 3072 x 1024 f32 output
SPLITK_WS_BYTES = 32768 + 16 * 4096 * 4 + 16 * 3072 * 1024 * 4


class Workspace:
    """Named device buffers reused across steps (shapes are static in training)."""

    def __init__(self, device):
        self.device = device
        self.bufs: Dict[str, torch.Tensor] = {}
        self.prefix = ""          # slot prefix: several forward passes (stage 3) keep separate activations / scratch

    def peek(self, name: str) -> torch.Tensor:
        return self.bufs[self.prefix + name]

    def get(self, name: str, shape, dtype) -> torch.Tensor:
        name = self.prefix + name
        t = self.bufs.get(name)
        shape = tuple(int(s) for s in shape)
        if t is None or tuple(t.shape) != shape or t.dtype != dtype:
            t = torch.empty(shape, dtype=dtype, device=self.device)
            self.bufs[name] = t
        return t

    def bytes_(self, name: str, nbytes: int) -> torch.Tensor:
        name = self.prefix + name
        t = self.bufs.get(name)
        if t is None or t.numel() < nbytes:
            t = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=self.device)
            self.bufs[name] = t
        return t


class ViTRunner:
    def __init__(self, fp: FlatParams, prefix: str, embed_dim: int, depth: int, num_heads: int, mlp_hidden: int,
                 ln_eps: float, patch_size: int, num_patches: int, pos_table: torch.Tensor, drop_path_rates: Sequence[float]):
        if embed_dim != num_heads * 64:
            raise NotImplementedError("the gfx950 attention kernels are built for head_dim 64")
        self.fp, self.pre = fp, prefix
        self.D, self.depth, self.H, self.Hd = embed_dim, depth, num_heads, mlp_hidden
        self.eps, self.P, self.num_patches = ln_eps, patch_size, num_patches
        self.pos_table = pos_table              # f32 [num_patches, D] on device (not a parameter, modeling_adaptation.py:91)
        self.dp_rates = list(drop_path_rates)
        self.ws = Workspace(fp.device)
        self.saved: List[dict] = []
        self._fw: dict = {}
        self._slots: Dict[str, tuple] = {}
        self._slot = ""
        self.scale = 64 ** -0.5                 # head_dim ** -0.5 (modeling_finetune.py:86)
        self.wgrad_stream = os.environ.get("UNITE_WGRAD_STREAM", "1") != "0"
        # how many streams the four weight-gradient GEMMs of a block are spread over (1: one after the other)
        self.wgrad_streams = min(4, max(1, int(os.environ.get("UNITE_WGRAD_STREAMS", "1"))))
        # student decoders on a side stream beside the encoder blocks (modeling_adaptation): shortens the student's serial chain by ~0.6 ms
        # at B = 32, but with the teacher one batch ahead the GPU is full either way: 20.71 vs 20.55 ms per step without it -> off
        self.side_decoders = os.environ.get("UNITE_DECODER_STREAM", "0") != "0"
        # planner weight of the weight-gradient GEMMs (None: whatever the step runs under).  They sit on a side stream off the critical chain, so
        # CU time, not latency, is what they cost an overlapped step: UNITE_WGRAD_SHARING pins the weight for A/B runs
        self.wgrad_sharing = float(os.environ["UNITE_WGRAD_SHARING"]) if "UNITE_WGRAD_SHARING" in os.environ else None
        self.wgrad_rowsum = os.environ.get("UNITE_WGRAD_ROWSUM", "1") != "0"      # bias gradients beside the weight-gradient products (0: separate column-sum kernels)
        # UNITE_GELU_DSAVE=1: fc1 saves GELU'(z) (16-bit fixed point, error 1.6e-5) instead of z, and the fc2 input-gradient epilogue multiplies by it:
        # one CDF evaluation in the forward serves both, where recomputing GELU' from the bf16 z costs the backward epilogue ~25 VALU instructions
        # per element.  Measured (round 4, profiles/r04_clock_notes.txt section 19): fc2 input gradient 58.9 against 66.7 us on one box and equal
        # on another (it is HBM-bound by the saved tensor either way), fc1 forward + 4.5 us, the step equal within 0.01 ms -> off by default.
        self.gelu_dsave = os.environ.get("UNITE_GELU_DSAVE", "0") != "0"
        self.fused_colsum = os.environ.get("UNITE_FUSED_COLSUM", "0") != "0"      # fc1 bias gradient out of the fc2-dgrad GEMM epilogue (no gain: the separate colsum hides on the side stream)
        self._side = None
        self.step_params = None          # graph_step.StepParams: stochastic depth then reads its seed from device memory

    def next_drop_path_seed(self) -> int:
        if getattr(self, "_dp_seed", None) is None:
            self._dp_seed = (torch.initial_seed() * 0x9E3779B1) & 0xFFFFFFFFFFFFFFFF      # follows torch.manual_seed(seed + rank)
        self._dp_seed = (self._dp_seed + 0x632BE59BD9B4E019) & 0xFFFFFFFFFFFFFFFF
        return self._dp_seed

    def _side_stream(self, k: int = 0):
        if self._side is None:
            self._side = []
        while len(self._side) <= k:
            self._side.append(torch.cuda.Stream(device=self.fp.device, priority=int(os.environ.get("UNITE_WGRAD_PRIO", "0"))))
        return self._side[k]

    def use_slot(self, slot: str) -> None:
        """Switch the set of saved activations / scratch buffers (stage 3 runs several forward passes of different
        shapes before their backward passes; each keeps its own slot)."""
        self._slots[self._slot] = (self.saved, self._fw)
        self._slot = slot
        self.ws.prefix = (slot + ":") if slot else ""
        self.saved, self._fw = self._slots.get(slot, ([], {}))

    # ------------------------------------------------------------------ helpers
    def _p(self, name):
        return self.fp.params[self.fp.names.index(self.pre + name)].data

    def _w(self, name):
        return self.fp.w16(self.pre + name)

    def _g(self, name):
        return self.fp.g(self.pre + name)

    def _cache_names(self):
        # resolve parameter tensors once (name lookups are off the hot path afterwards)
        if hasattr(self, "_blk"):
            return
        fp, pre = self.fp, self.pre
        idx = {n: i for i, n in enumerate(fp.names)}

        def P(n):
            return fp.params[idx[pre + n]].data

        self._blk = []
        for i in range(self.depth):
            b = f"blocks.{i}."
            d = {}
            for n in ["norm1.weight", "norm1.bias", "attn.q_bias", "attn.v_bias", "attn.proj.bias", "norm2.weight", "norm2.bias",
                      "mlp.fc1.bias", "mlp.fc2.bias"]:
                d[n] = P(b + n)
                d["g:" + n] = fp.g(pre + b + n)
            for n in ["attn.qkv.weight", "attn.proj.weight", "mlp.fc1.weight", "mlp.fc2.weight"]:
                d[n] = fp.w16(pre + b + n)
                d["g:" + n] = fp.g(pre + b + n)
            d["qkv_bias"] = fp.packed_qkv_bias(pre + b + "attn.")        # (q_bias, 0, v_bias), modeling_finetune.py:104
            d["g:qkv_bias"] = fp.packed_qkv_bias_grad(pre + b + "attn.")
            self._blk.append(d)
        self._pe_w = fp.w16(pre + "patch_embed.proj.weight")
        self._pe_b = P("patch_embed.proj.bias")
        self._pe_gw = fp.g(pre + "patch_embed.proj.weight")
        self._pe_gb = fp.g(pre + "patch_embed.proj.bias")

    def drop_path_scales(self, B: int, training: bool) -> Optional[torch.Tensor]:
        """(depth, 2, B) f32 multipliers floor(keep + U)/keep (timm drop_path as used at modeling_finetune.py:50), or None."""
        if not training or max(self.dp_rates) == 0.0:
            return None
        if getattr(self, "_keep", None) is None:        # uploaded once: a host -> device copy here would block the host every step
            self._keep = (1.0 - torch.tensor(self.dp_rates, dtype=F32)).to(self.fp.device)
        out = self.ws.get("dp.scales", (self.depth, 2, B), F32)
        if self.step_params is not None:       # captured step: the host advanced and published the seed before the launch
            return ops.drop_path_scales(self._keep, 0, out, seed_dev=self.step_params.seed_dp_dev)
        return ops.drop_path_scales(self._keep, self.next_drop_path_seed(), out)

    # ------------------------------------------------------------------ forward
    def embed(self, videos: torch.Tensor, tokens: Optional[torch.Tensor], M: int) -> torch.Tensor:
        """patch-embed GEMM on the listed tokens + bias + sinusoid position rows -> x0 f32 [M, D]."""
        self._cache_names()
        D, ws = self.D, self.ws
        Kpe = 3 * self.P * self.P
        if videos.shape[1] != 3 or videos.dtype != F32:
            raise ValueError("videos must be float32 (B,3,T,H,W)")
        cols = ws.get("pe.cols", (M, Kpe), BF16)
        ops.im2col_gather(videos.contiguous(), tokens, cols, self.P)
        pos = ws.get("pe.pos", (M, D), F32)
        ops.gather_rows(self.pos_table, tokens, pos, modulo=self.num_patches)
        x0 = ws.get("x.0", (M, D), F32)
        ops.gemm(cols, self._pe_w.view(D, Kpe), x0, bias=self._pe_b, residual=pos)
        return x0

    def blocks_forward(self, x0: torch.Tensor, B: int, N: int, n_blocks: int, dp: Optional[torch.Tensor], save: bool,
                       after_block=None) -> List[torch.Tensor]:
        """Runs blocks 0..n_blocks-1; returns [x0, x_out(0), ..., x_out(n_blocks-1)] (f32 [M, D] each).  ``after_block(i, x_out)`` is
        called when block i has been enqueued (the student hangs its decoders there)."""
        self._cache_names()
        D, H, Hd, ws = self.D, self.H, self.Hd, self.ws
        M = B * N
        xs = [x0]
        self.saved = []
        x = x0
        for i in range(n_blocks):
            w = self._blk[i]
            t = f"b{i}." if save else "t."          # without save, activations share one set of buffers
            h1 = ws.get(t + "h1", (M, D), BF16)
            mean1, rstd1 = ws.get(t + "mean1", (M,), F32), ws.get(t + "rstd1", (M,), F32)
            ops.layernorm_fwd(x, w["norm1.weight"], w["norm1.bias"], self.eps, h1, mean=mean1, rstd=rstd1)
            qkv = ws.get(t + "qkv", (M, 3 * D), BF16)
            ops.gemm(h1, w["attn.qkv.weight"], qkv, bias=w["qkv_bias"])
            o = ws.get(t + "o", (M, D), BF16)
            lse = ws.get(t + "lse", (B, H, N), F32)
            ops.attn_fwd(qkv, o, lse, B, N, H, self.scale)
            x1 = ws.get(t + "x1", (M, D), F32)
            ops.gemm(o, w["attn.proj.weight"], x1, bias=w["attn.proj.bias"], residual=x,
                     row_scale=None if dp is None else dp[i, 0], rows_per_scale=N)
            h2 = ws.get(t + "h2", (M, D), BF16)
            mean2, rstd2 = ws.get(t + "mean2", (M,), F32), ws.get(t + "rstd2", (M,), F32)
            ops.layernorm_fwd(x1, w["norm2.weight"], w["norm2.bias"], self.eps, h2, mean=mean2, rstd=rstd2)
            z = ws.get(t + "z", (M, Hd), BF16) if save else None
            a = ws.get(t + "a", (M, Hd), BF16)
            ops.gemm(h2, w["mlp.fc1.weight"], a, bias=w["mlp.fc1.bias"], act=ops.ACT_GELU_DSAVE if (save and self.gelu_dsave) else ops.ACT_GELU, aux_out=z)
            x2 = ws.get(f"x.{i + 1}" if save else f"t.x{(i + 1) & 1}", (M, D), F32)
            ops.gemm(a, w["mlp.fc2.weight"], x2, bias=w["mlp.fc2.bias"], residual=x1,
                     row_scale=None if dp is None else dp[i, 1], rows_per_scale=N)
            if save:
                self.saved.append(dict(x_in=x, h1=h1, mean1=mean1, rstd1=rstd1, qkv=qkv, o=o, lse=lse, x1=x1, h2=h2,
                                       mean2=mean2, rstd2=rstd2, z=z, a=a))
            xs.append(x2)
            x = x2
            if after_block is not None:
                after_block(i, x2)
        self._fw = dict(B=B, N=N, dp=dp)
        return xs

    # ------------------------------------------------------------------ backward
    def blocks_backward(self, dx: torch.Tensor, dxb: torch.Tensor, n_blocks: int, tap_layers=(), tap_hook=None, layer_done=None):
        """dx: f32 [M,D] gradient w.r.t. x_out(n_blocks-1); dxb: its bf16 copy already multiplied by the drop-path scale of
        that block's MLP branch (whoever produced dxb also produced its column sums = the fc2 bias gradient).
        For i-1 in tap_layers, tap_hook(i-1, dx, scale, dxsum) must add the tap's gradient to dx and return the final
        (dx, dxb) for block i-1 (and write dxsum = column sums of dxb).  Writes every block's parameter gradients and the
        patch-embed bias gradient; returns (dx0 f32, dx0 bf16 unscaled) w.r.t. the block-0 input."""
        fw = self._fw
        B, N, dp = fw["B"], fw["N"], fw["dp"]
        D, H, Hd, ws, fp = self.D, self.H, self.Hd, self.ws, self.fp
        M = B * N
        acc = fp.accumulate
        lnws = ws.bytes_("ln.ws", ops.layernorm_bwd_workspace(M, max(D, 1)))
        nside = self.wgrad_streams if (dx.is_cuda and self.wgrad_stream) else 1
        gws_k = [ws.bytes_("gemm.ws" if k == 0 else f"gemm.ws.{k}", SPLITK_WS_BYTES) for k in range(nside)]      # split-K slabs: one set per stream
        csws_k = [ws.bytes_(f"cs.ws.{k}", ops.colsum_workspace(M, max(Hd, 3 * D))) for k in range(nside)] if not self.wgrad_rowsum else None
        gcws = ws.bytes_("gemm.cs.ws", ops.gemm_colsum_workspace(M, Hd))
        # Weight-gradient GEMMs and bias column sums are off the critical path (nothing in the backward chain reads them): they
        # run on a side HIP stream behind events, concurrently with the next dgrad GEMM / LayerNorm backward / attention backward
        # on the main stream, which leave MFMA or HBM headroom.  Their operands (dz, dx1b, dqkv, the incoming dxb) live in
        # buffers alternated by block parity, so the main stream only has to wait for the side work of two blocks ago.
        side = self._side_stream() if dx.is_cuda and self.wgrad_stream else None
        sides = [self._side_stream(k) for k in range(nside)] if side is not None else []
        main = torch.cuda.current_stream() if side is not None else None
        done_ev = {}                                # block -> events, one per side stream that worked for it

        def on_side(ev_fn, k=0, blk=None):
            """run ev_fn(k) on side stream k (modulo the number in use) after everything issued so far on the main stream; with ``blk``
            the stream's position afterwards is one of the events that mark block blk's weight gradients as written"""
            if side is None:
                ev_fn(0)
                return
            k %= nside
            ev = torch.cuda.Event()
            ev.record(main)
            sides[k].wait_event(ev)
            with torch.cuda.stream(sides[k]), ops.plan(sharing=self.wgrad_sharing):
                ev_fn(k)
                if blk is not None:
                    e2 = torch.cuda.Event()
                    e2.record(sides[k])
                    done_ev.setdefault(blk, {})[k] = e2      # a later call on the same stream supersedes the earlier event

        def retire(j):
            """the main stream may overwrite the parity buffers block j's side-stream work read"""
            if side is not None and j in done_ev:
                for e in done_ev.pop(j).values():
                    main.wait_event(e)

        def report(j):
            """every gradient of block j is written: the side stream's part at done_ev[j], the main stream's part (LayerNorm
            gamma / beta, bias column sums) by now.  The reducer's collective waits for exactly these two events, so a bucket
            can start as soon as its last layer's weight gradients finish, not when the main stream next meets them."""
            if layer_done is None:
                return
            if side is None:
                layer_done(j)
                return
            ev = torch.cuda.Event()
            ev.record(main)
            layer_done(j, (ev, *done_ev[j].values()))

        for i in reversed(range(n_blocks)):
            w, s = self._blk[i], self.saved[i]
            par = i & 1
            if side is not None and i + 2 < n_blocks:
                retire(i + 2)                       # its side work read the parity buffers this block is about to overwrite
            # ---- MLP branch
            dz = ws.get(f"bw.dz{par}", (M, Hd), BF16)
            # the fc1 bias gradient = column sums of dz comes out of this GEMM's epilogue (dz is not read again for it)
            dact = ops.ACT_MULAUX if self.gelu_dsave else ops.ACT_DGELU        # s["z"] holds GELU'(z) in the first case, z in the second
            if self.fused_colsum:
                ops.gemm(dxb, w["mlp.fc2.weight"], dz, trans_b=True, act=dact, aux_in=s["z"], workspace=gcws,
                         colsum_out=w["g:mlp.fc1.bias"], colsum_accumulate=acc)
            else:
                ops.gemm(dxb, w["mlp.fc2.weight"], dz, trans_b=True, act=dact, aux_in=s["z"])
            # (one grouped launch of the block's four weight gradients -- ops.gemm_grouped -- is 12 % faster in isolation at 10 240
            # tokens but slower in the step: its 252 MB of operands have left the Infinity Cache by the end of the block)

            def fc2_wgrad(k, dxb=dxb, w=w, s=s):
                ops.gemm(dxb, s["a"], w["g:mlp.fc2.weight"], trans_a=True, trans_b=True, accumulate=acc, workspace=gws_k[k])

            def fc1_wgrad(k, dz=dz, w=w, s=s):
                # the fc1 bias gradient = column sums of dz = row sums of this product's A operand (dz^T): taken from the A tiles in LDS
                if self.fused_colsum:
                    ops.gemm(dz, s["h2"], w["g:mlp.fc1.weight"], trans_a=True, trans_b=True, accumulate=acc, workspace=gws_k[k])
                elif not self.wgrad_rowsum:
                    ops.gemm(dz, s["h2"], w["g:mlp.fc1.weight"], trans_a=True, trans_b=True, accumulate=acc, workspace=gws_k[k])
                    ops.colsum(dz, w["g:mlp.fc1.bias"], csws_k[k], accumulate=acc)
                else:
                    ops.gemm(dz, s["h2"], w["g:mlp.fc1.weight"], trans_a=True, trans_b=True, accumulate=acc, workspace=gws_k[k],
                             rowsum_out=w["g:mlp.fc1.bias"], rowsum_accumulate=acc)
            on_side(fc2_wgrad, 0, i)
            on_side(fc1_wgrad, 1, i)
            dh2 = ws.get("bw.dh", (M, D), BF16)
            ops.gemm(dz, w["mlp.fc1.weight"], dh2, trans_b=True)
            dx1 = ws.get("bw.dx1", (M, D), F32)
            dx1b = ws.get(f"bw.dx1b{par}", (M, D), BF16)
            ops.layernorm_bwd(dh2, s["x1"], s["mean2"], s["rstd2"], w["norm2.weight"], dx_residual=dx, dx_out=dx1, dx_bf16=dx1b,
                              row_scale=None if dp is None else dp[i, 0], rows_per_scale=N,
                              dgamma=w["g:norm2.weight"], dbeta=w["g:norm2.bias"], dxsum=w["g:attn.proj.bias"],
                              accumulate=acc, workspace=lnws)
            # ---- attention branch
            do = ws.get("bw.do", (M, D), BF16)
            ops.gemm(dx1b, w["attn.proj.weight"], do, trans_b=True)
            on_side(lambda k, dx1b=dx1b, w=w, s=s: ops.gemm(dx1b, s["o"], w["g:attn.proj.weight"], trans_a=True, trans_b=True, accumulate=acc,
                                                           workspace=gws_k[k]), 2, i)
            dqkv = ws.get(f"bw.dqkv{par}", (M, 3 * D), BF16)
            delta = ws.get("bw.delta", (B, H, N), F32)
            ops.attn_bwd(s["qkv"], s["o"], do, s["lse"], delta, dqkv, B, N, H, self.scale)
            dh1 = ws.get("bw.dh", (M, D), BF16)
            ops.gemm(dqkv, w["attn.qkv.weight"], dh1, trans_b=True)

            def qkv_wgrads(k, dqkv=dqkv, w=w, s=s):
                if not self.wgrad_rowsum:
                    ops.gemm(dqkv, s["h1"], w["g:attn.qkv.weight"], trans_a=True, trans_b=True, accumulate=acc, workspace=gws_k[k])
                    ops.colsum(dqkv, w["g:qkv_bias"], csws_k[k], accumulate=acc, zero_range=(D, 2 * D))
                    return
                ops.gemm(dqkv, s["h1"], w["g:attn.qkv.weight"], trans_a=True, trans_b=True, accumulate=acc, workspace=gws_k[k],
                         rowsum_out=w["g:qkv_bias"], rowsum_accumulate=acc, rowsum_zero_range=(D, 2 * D))     # (dq_bias, 0, dv_bias)
            on_side(qkv_wgrads, 3, i)
            # gradient w.r.t. this block's input; its bf16 copy feeds block i-1's MLP branch (scaled by that
            # branch's drop-path factor) or the patch-embed weight gradient (unscaled)
            dx0 = ws.get(f"bw.dx{i & 1}", (M, D), F32)
            dx0b = ws.get(f"bw.dxb{i % 3}", (M, D), BF16)     # read by block i-1's side work, rewritten by block i-3 (after retire(i-1))
            nxt_scale = None if (dp is None or i == 0) else dp[i - 1, 1]
            nxt_bias_g = self._blk[i - 1]["g:mlp.fc2.bias"] if i > 0 else self._pe_gb
            tapped = i > 0 and (i - 1) in tap_layers      # the tap hook then emits the final bf16 copy and its column sums
            ops.layernorm_bwd(dh1, s["x_in"], s["mean1"], s["rstd1"], w["norm1.weight"], dx_residual=dx1, dx_out=dx0,
                              dx_bf16=None if tapped else dx0b, row_scale=nxt_scale, rows_per_scale=N,
                              dgamma=w["g:norm1.weight"], dbeta=w["g:norm1.bias"], dxsum=None if tapped else nxt_bias_g,
                              accumulate=acc, workspace=lnws)
            dx, dxb = dx0, dx0b
            if tapped:
                dx, dxb = tap_hook(i - 1, dx, nxt_scale, nxt_bias_g)
            report(i)
        if side is not None:
            for j in sorted(done_ev, reverse=True):
                retire(j)
        return dx, dxb

    def embed_backward(self, dxb: torch.Tensor):
        """patch-embed weight/bias gradients from the bf16 gradient w.r.t. x0 (no input gradient: the clip is data)."""
        ws, fp, D = self.ws, self.fp, self.D
        cols = ws.peek("pe.cols")
        M = cols.shape[0]
        ops.gemm(dxb, cols, self._pe_gw.view(D, -1), trans_a=True, trans_b=True, accumulate=fp.accumulate,
                 workspace=ws.bytes_("gemm.ws", SPLITK_WS_BYTES))      # the bias gradient came with dxb (blocks_backward)
