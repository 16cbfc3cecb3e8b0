"""Frozen CLIP image teacher on the gfx950 kernels -- drop-in for reference src/models/clip.py.

Same factories (``clip_b16`` / ``clip_l14`` / ``clip_l14_336``), keyword arguments, ``state_dict`` keys (OpenAI
``visual.*`` layout with the 2-D conv1 inflated to 3-D) and ``forward(x) -> (feats (K,B,T*HW,C), attn (B*T,HW))``.
The nn modules only hold parameters; the forward is a fixed launch schedule over libunite_hip.so:
  conv1-as-GEMM -> [cls ; patches] + pos -> ln_pre -> 12 x (LN, qkv GEMM, fused attention, out_proj GEMM(+res),
  LN, c_fc GEMM(+QuickGELU), c_proj GEMM(+res)) -> CLS-row attention probabilities of the last block ->
  ln_post + proj + L2-norm on the VISIBLE tokens only (the reference computes the tail for all 1568, clip.py:168-173).
The last block's head-averaged attention map (clip.py:95-96) is never materialised: only its CLS row is computed.

Built for head_dim 64: clip_b16 (patch 16) and clip_l14 (patch 14: the 588-wide im2col rows and conv1 rows are zero-padded
to K = 592 so they stay 16-byte multiples).  Frames of up to 1024 patches (+CLS): 224 @ 16, 196 @ 14 (SURVEY 8d cfg 4/5), 336 @ 14 (clip_l14_336).
"""
from __future__ import annotations

import os
from collections import OrderedDict
from typing import List, Optional

import torch
import torch.nn as nn

from . import ops
from .flat_params import FlatParams
from .vit_runner import Workspace, BF16, F32

MODEL_PATH = os.environ.get("UNITE_CLIP_PATH", "your_model_path/clip_visual_encoder")
_MODELS = {
    "ViT-B/16": os.path.join(MODEL_PATH, "vit_b16.pth"),
    "ViT-L/14": os.path.join(MODEL_PATH, "vit_l14.pth"),
    "ViT-L/14_336": os.path.join(MODEL_PATH, "vit_l14_336.pth"),
}


class LayerNorm(nn.LayerNorm):
    """parameter holder (reference clip.py:20-26 runs it in fp32; so do the kernels)."""


class QuickGELU(nn.Module):
    """marker module: x * sigmoid(1.702 x) is fused into the c_fc GEMM epilogue (reference clip.py:29-31)."""


class ResidualAttentionBlock(nn.Module):
    def __init__(self, d_model, n_head, attn_mask=None):
        super().__init__()
        self.attn = nn.MultiheadAttention(d_model, n_head)      # holds in_proj_weight/in_proj_bias/out_proj.*
        self.ln_1 = LayerNorm(d_model)
        self.mlp = nn.Sequential(OrderedDict([("c_fc", nn.Linear(d_model, d_model * 4)), ("gelu", QuickGELU()),
                                              ("c_proj", nn.Linear(d_model * 4, d_model))]))
        self.ln_2 = LayerNorm(d_model)
        self.attn_mask = attn_mask


class Transformer(nn.Module):
    def __init__(self, width, layers, heads, return_attn=False, clip_return_layers=[6, 7, 8, 9, 10, 11],
                 clip_return_interval=1, return_cls=False):
        super().__init__()
        self.layers, self.return_attn, self.return_cls = layers, return_attn, return_cls
        self.resblocks = nn.ModuleList([ResidualAttentionBlock(width, heads) for _ in range(layers)])
        self.return_index = list(clip_return_layers)


class VisionTransformer(nn.Module):
    def __init__(self, input_resolution, patch_size, width, layers, heads, output_dim, clip_norm_type='l2', kernel_size=1,
                 return_attn=False, clip_return_layers=[6, 7, 8, 9, 10, 11], clip_return_interval=1, return_cls=False):
        super().__init__()
        if clip_norm_type != 'l2':
            raise NotImplementedError("clip_norm_type must be 'l2'")
        if kernel_size != 1 or return_cls:
            raise NotImplementedError("kernel_size != 1 / return_cls are not built (unused by the UNITE configs)")
        self.clip_norm_type, self.return_attn, self.return_cls = clip_norm_type, return_attn, return_cls
        self.input_resolution, self.patch_size, self.width, self.heads = input_resolution, patch_size, width, heads
        self.output_dim = output_dim
        self.conv1 = nn.Conv3d(3, width, (kernel_size, patch_size, patch_size), (kernel_size, patch_size, patch_size), (0, 0, 0),
                               bias=False)
        scale = width ** -0.5
        self.class_embedding = nn.Parameter(scale * torch.randn(width))
        self.positional_embedding = nn.Parameter(scale * torch.randn((input_resolution // patch_size) ** 2 + 1, width))
        self.ln_pre = LayerNorm(width)
        self.transformer = Transformer(width, layers, heads, return_attn=return_attn, clip_return_layers=clip_return_layers,
                                       clip_return_interval=clip_return_interval, return_cls=return_cls)
        self.ln_post = LayerNorm(width)
        self.proj = nn.Parameter(scale * torch.randn(width, output_dim))
        self._rt: Optional[_TeacherRuntime] = None

    def runtime(self) -> "_TeacherRuntime":
        if self._rt is None:
            dev = self.class_embedding.device
            if dev.type != "cuda":
                raise RuntimeError("unite_amd models run on a MI355X only: move the teacher to 'cuda' first (no CPU path)")
            self._rt = _TeacherRuntime(self, dev)
        return self._rt

    def _apply(self, fn, *a, **k):
        if self._rt is not None:
            raise RuntimeError("teacher.to()/cuda() after the first forward would detach the flat parameter buffer")
        return super()._apply(fn, *a, **k)

    @torch.no_grad()
    def forward(self, x, mask=None):
        """reference clip.py:145-188 (mask=None path).  Returns feats (K,B,T*HW,C) [and attn (B*T,HW) if return_attn]."""
        if mask is not None:
            raise NotImplementedError("teacher-side token masking (clip.py:154-160) is unused by run_stage1/3 and not built")
        rt = self.runtime()
        attn = rt.forward_taps(x)
        B, T = x.shape[0], x.shape[2]
        feats = rt.targets(rt.all_rows(B * T), B * T * rt.HW).view(len(rt.taps), B, T * rt.HW, self.output_dim)
        return (feats, attn) if self.return_attn else feats

    @torch.no_grad()
    def forward_attention(self, x):
        """Engine entry: run the transformer once, keep the taps on device, return the CLS attention (B*T, HW)."""
        return self.runtime().forward_taps(x)

    @torch.no_grad()
    def encode_image(self, videos):
        """(B,3,T,H,W) -> (B*T, output_dim) L2-normalised frame embeddings (zero-shot CLIP image side, src/utils.py:55-61)."""
        return self.runtime().cls_features(videos)

    @torch.no_grad()
    def visible_targets(self, vis_rows_cls, n_rows, slot=0):
        """ln_post + proj + L2-norm of the taps at the listed rows only -> f32 [K*n_rows, C] (run_stage1.py:389-397).
        ``slot`` picks the output buffer (the engine's teacher-ahead mode keeps two batches' targets alive)."""
        return self.runtime().targets(vis_rows_cls, n_rows, slot)


class _TeacherRuntime:
    DEFAULT_RES16 = "f16"        # residual-stream type when UNITE_TEACHER_RES16 is unset (see __init__; tools/teacher_stream_error.py)

    def __init__(self, model: VisionTransformer, dev):
        if model.patch_size % 2:
            raise NotImplementedError("odd patch sizes are not built")
        if model.width != model.heads * 64:
            raise NotImplementedError("head_dim must be 64")
        self.model, self.dev = model, dev
        self.fp = FlatParams(model, dev, with_grad=False)
        for p in model.parameters():
            p.requires_grad_(False)          # frozen (SURVEY A-18)
        self.ws = Workspace(dev)
        self.D, self.H, self.C, self.P = model.width, model.heads, model.output_dim, model.patch_size
        self.HW = (model.input_resolution // model.patch_size) ** 2
        self.L = self.HW + 1
        self.layers = model.transformer.layers
        self.taps: List[int] = list(model.transformer.return_index)
        self.eps = model.ln_pre.eps
        fp = self.fp
        idx = {n: i for i, n in enumerate(fp.names)}

        def P(n):
            return fp.params[idx[n]].data

        self.cls, self.pos = P("class_embedding"), P("positional_embedding")
        self.ln_pre = (P("ln_pre.weight"), P("ln_pre.bias"))
        self.ln_post = (P("ln_post.weight"), P("ln_post.bias"))
        self.conv_w = fp.w16("conv1.weight")
        self.proj_w = fp.w16("proj")
        self.blk = []
        for i in range(self.layers):
            b = f"transformer.resblocks.{i}."
            self.blk.append(dict(
                ln1=(P(b + "ln_1.weight"), P(b + "ln_1.bias")), ln2=(P(b + "ln_2.weight"), P(b + "ln_2.bias")),
                w_in=fp.w16(b + "attn.in_proj_weight"), b_in=P(b + "attn.in_proj_bias"),
                w_out=fp.w16(b + "attn.out_proj.weight"), b_out=P(b + "attn.out_proj.bias"),
                w_fc=fp.w16(b + "mlp.c_fc.weight"), b_fc=P(b + "mlp.c_fc.bias"),
                w_pr=fp.w16(b + "mlp.c_proj.weight"), b_pr=P(b + "mlp.c_proj.bias")))
        self._rows_cache = {}
        self.n_streams = max(1, int(os.environ.get("UNITE_TEACHER_STREAMS", "3")))
        self.fused_qkv = os.environ.get("UNITE_TEACHER_FUSED", "1") != "0" and 192 < self.L <= 224
        # Residual stream of the FROZEN teacher (UNITE_TEACHER_RES16): x, x1 and the taps are written and re-read as 16-bit rows, which takes
        # 464 MB per block off the HBM traffic of the two residual GEMMs and the two LayerNorms (2.0 -> 1.55 GB per block at B = 32; -0.5 ms of
        # a 19.4-ms step).  The sums themselves stay f32 (accumulator + residual are added in f32 in the GEMM epilogue, rounded once on the
        # store; LayerNorm statistics in f32).
        #   "f16"       (default) IEEE half rows -- the type OpenAI's CLIP keeps its own residual stream in (clip.load() on a GPU: model.half(),
        #               LayerNorm computed in fp32), 11 significant bits.  Over 24 seeded towers against the fp32 oracle the CLS attention error is
        #               1.43e-3 rms with it and 1.45e-3 with f32 rows, the target features' mean cosine 0.9999433 and 0.9999432
        #               (profiles/r04_teacher_stream_error.txt): the bf16 operands of the products set the error, not this stream
        #   "1"/"bf16"  bf16 rows (round 3): 8 significant bits; 1.74e-3 rms / 0.9999269 on the same towers, and on the tiny golden teacher the CLS
        #               attention moves by up to 1e-2 absolute -- never the default
        #   "0"/"f32"   f32 rows, what the reference's autocast path keeps (x + fp16 branch output -> fp32)
        mode = os.environ.get("UNITE_TEACHER_RES16", self.DEFAULT_RES16).lower()
        kinds = {"0": False, "": False, "f32": False, "1": True, "bf16": True, "f16": "f16"}
        if mode not in kinds:
            raise ValueError(f"UNITE_TEACHER_RES16={mode!r}: expected f16, bf16 (or 1), f32 (or 0)")
        self.res16 = kinds[mode]
        self.min_frames_per_stream = 64
        self._side = []
        # the flat parameter buffer and its bf16 shadow were just built on the CURRENT stream: the first forward may come from another one
        # (TeacherAhead / MaskTeacherAhead call runtime() on the student's stream and run the teacher on their own), so the construction
        # counts as the first "use" the next one is ordered behind.  (Round 3: found as masks / targets of the first batch computed from
        # half-copied weights whenever the student's stream was busy at that moment.)
        self._use_ev, self._use_stream = None, None
        self._leave(torch.cuda.current_stream())

    # The workspace (x / h / qkv / o / a / taps) and the tap bookkeeping belong to ONE forward at a time.  Two callers on different
    # streams -- stage 3's mask teacher one batch ahead on its own stream and utils.clip_infer(teacher_model, ...) on the student's stream
    # -- are put in order here: a use starts behind the end of the previous use whenever that one ran on another stream.
    def _enter(self):
        cur = torch.cuda.current_stream()
        if self._use_ev is not None and self._use_stream != cur:
            cur.wait_event(self._use_ev)
        return cur

    def _leave(self, cur):
        ev = torch.cuda.Event()
        ev.record(cur)
        self._use_ev, self._use_stream = ev, cur

    @property
    def two_streams(self) -> bool:
        return self.n_streams > 1

    @two_streams.setter
    def two_streams(self, on: bool):
        self.n_streams = (max(2, int(os.environ.get("UNITE_TEACHER_STREAMS", "3"))) if on else 1)

    def _side_stream(self, i: int = 0):
        while len(self._side) <= i:
            self._side.append(torch.cuda.Stream(device=self.dev))
        return self._side[i]

    def all_rows(self, BT: int) -> torch.Tensor:
        t = self._rows_cache.get(BT)
        if t is None:
            j = torch.arange(BT * self.HW, dtype=torch.int32, device=self.dev)
            t = (j + j // self.HW + 1).contiguous()
            self._rows_cache[BT] = t
        return t

    def forward_taps(self, videos: torch.Tensor) -> torch.Tensor:
        cur = self._enter()
        try:
            return self._forward_taps(videos)
        finally:
            self._leave(cur)

    def _forward_taps(self, videos: torch.Tensor) -> torch.Tensor:
        self.fp.refresh_if_stale()
        ws, D, H, L, HW = self.ws, self.D, self.H, self.L, self.HW
        B, Cc, T, Hh, Ww = videos.shape
        if Hh != self.model.input_resolution or Ww != self.model.input_resolution:
            raise ValueError("clip resolution mismatch")
        BT = B * T
        Mp, Mt = BT * HW, BT * L
        Kpe = 3 * self.P * self.P
        Kp = (Kpe + 7) // 8 * 8                       # 588 -> 592 for patch 14: rows stay 16-byte multiples, the pad is zeros
        cols = ws.get("cols", (Mp, Kp), BF16)
        ops.im2col_gather(videos.contiguous(), None, cols, self.P)
        conv_w = self.conv_w.view(D, Kpe)
        if Kp != Kpe:
            wpad = ws.bufs.get(ws.prefix + "conv_w.pad")
            if wpad is None:
                wpad = ws.get("conv_w.pad", (D, Kp), BF16)
                wpad.zero_()
            wpad[:, :Kpe].copy_(conv_w)               # 1.2 MB; follows a reloaded state_dict
            conv_w = wpad
        patches = ws.get("patches", (Mp, D), BF16)
        ops.gemm(cols, conv_w, patches)
        RES = torch.float16 if self.res16 == "f16" else BF16 if self.res16 else F32
        x = ws.get("x.a", (Mt, D), RES)
        ops.clip_embed_ln(patches, self.cls, self.pos, self.ln_pre[0], self.ln_pre[1], self.eps, x, BT, HW, D)
        h = ws.get("h", (Mt, D), BF16)
        qkv = ws.get("qkv", (Mt, 3 * D), BF16)
        o = ws.get("o", (Mt, D), BF16)
        lse = ws.get("lse", (BT, H, L), F32)
        x1 = ws.get("x1", (Mt, D), RES)
        a = ws.get("a", (Mt, 4 * D), BF16)
        scale = 64 ** -0.5
        last = self.layers - 1
        pruned_tap = last in self.taps
        taps_full = {i: ws.get(f"tap.{i}", (Mt, D), RES) for i in self.taps if i != last}     # kept until targets() gathers rows
        self._tap_bufs = [taps_full[i] for i in sorted(taps_full)]
        self._last = dict(x=None, o=o) if pruned_tap else None

        def run_layers(f0: int, f1: int):
            """all blocks for frames [f0, f1): every buffer is used through its row slice, so two frame ranges are independent"""
            r0, r1, nf = f0 * L, f1 * L, f1 - f0
            xs = x[r0:r1]
            for i in range(self.layers):
                w = self.blk[i]
                ops.layernorm_fwd(xs, w["ln1"][0], w["ln1"][1], self.eps, h[r0:r1])
                if i != last and self.fused_qkv:
                    # projection + attention of a (frame, head) in one workgroup: qkv never goes to memory (464 MB per block)
                    ops.teacher_qkv_attn(h[r0:r1], w["w_in"], w["b_in"], o[r0:r1], nf, L, H, scale)
                else:
                    ops.gemm(h[r0:r1], w["w_in"], qkv[r0:r1], bias=w["b_in"])
                if i == last:
                    # Only the CLS attention row of the last block is needed for every token (the mask weights, clip.py:95-96,183).
                    # If the block is a tap, the rest of it (out_proj, MLP) runs in targets() on the rows whose features are used
                    # (320 of 1568 per clip); if not, nothing else of it is needed at all.
                    if pruned_tap:
                        ops.attn_fwd(qkv[r0:r1], o[r0:r1], lse[f0:f1], nf, L, H, scale)
                    return xs
                if not self.fused_qkv:
                    ops.attn_fwd(qkv[r0:r1], o[r0:r1], lse[f0:f1], nf, L, H, scale)
                ops.gemm(o[r0:r1], w["w_out"], x1[r0:r1], bias=w["b_out"], residual=xs)
                ops.layernorm_fwd(x1[r0:r1], w["ln2"][0], w["ln2"][1], self.eps, h[r0:r1])
                ops.gemm(h[r0:r1], w["w_fc"], a[r0:r1], bias=w["b_fc"], act=ops.ACT_QUICKGELU)
                xo = taps_full[i][r0:r1] if i in taps_full else x[r0:r1]      # may alias xs: the block input is dead after out_proj
                ops.gemm(a[r0:r1], w["w_pr"], xo, bias=w["b_pr"], residual=x1[r0:r1])
                xs = xo
            return xs

        # Independent ranges of the frames on their own HIP streams: while one range is in a LayerNorm / attention kernel (HBM- or
        # latency-bound), in a GEMM's store-bound epilogue round or in its ragged last round, the others' workgroups fill the idle CUs.
        # (measured: 3 ranges beat 2 at 256 frames, 23.3 -> 22.8 ms per step; 4 lose; at 128 frames 2 beat 3 -- a range wants >= 64 frames)
        n = min(self.n_streams, max(1, BT // self.min_frames_per_stream))
        if n > 1 and videos.is_cuda:
            main = torch.cuda.current_stream()
            cut = [BT * k // n for k in range(n + 1)]
            ev = torch.cuda.Event()
            ev.record(main)
            parts, joins = [None] * n, []
            for k in range(1, n):
                side = self._side_stream(k - 1)
                side.wait_event(ev)
                with torch.cuda.stream(side):
                    parts[k] = run_layers(cut[k], cut[k + 1])
                    ev2 = torch.cuda.Event()
                    ev2.record(side)
                    joins.append(ev2)
            parts[0] = run_layers(cut[0], cut[1])
            for ev2 in joins:
                main.wait_event(ev2)
            x_last = tuple(parts)
        else:
            x_last = (run_layers(0, BT),)
        if pruned_tap:
            # the block-input rows of both halves live in ONE of the full buffers (x.a or the previous tap): the same for both
            base = x_last[0]._base if x_last[0]._base is not None else x_last[0]
            self._last["x"] = base
        attn = ws.get("attn", (BT, HW), F32)
        ops.attn_cls_probs(qkv, attn, BT, L, H, scale)          # qkv still holds the last block's projections
        return attn

    def cls_features(self, videos: torch.Tensor) -> torch.Tensor:
        """L2-normalised image embeddings of every frame, (B*T, C): OpenAI CLIP's ``encode_image`` (ln_post(x[:, 0]) @ proj) as
        utils.clip_infer uses it (src/utils.py:55-61) -- this tower with the last block evaluated on the CLS rows only."""
        taps = self.taps
        self.taps = [self.layers - 1]
        try:
            self.forward_taps(videos)
            BT = videos.shape[0] * videos.shape[2]
            rows = (torch.arange(BT, dtype=torch.int32, device=self.dev) * self.L).contiguous()
            return self.targets(rows, BT)
        finally:
            self.taps = taps

    def _last_block_rows(self, rows: torch.Tensor, n_rows: int) -> torch.Tensor:
        """out_proj + MLP of the last block on the listed token rows only -> x_out [n_rows, D] (same arithmetic per row)."""
        ws, D, w = self.ws, self.D, self.blk[self.layers - 1]
        st = self._last
        RES = st["x"].dtype
        o_v = ops.gather_rows(st["o"], rows, ws.get("last.o", (n_rows, D), BF16))
        x_v = ops.gather_rows(st["x"], rows, ws.get("last.x", (n_rows, D), RES))
        x1 = ws.get("last.x1", (n_rows, D), RES)
        ops.gemm(o_v, w["w_out"], x1, bias=w["b_out"], residual=x_v)
        hh = ws.get("last.h", (n_rows, D), BF16)
        ops.layernorm_fwd(x1, w["ln2"][0], w["ln2"][1], self.eps, hh)
        aa = ws.get("last.a", (n_rows, 4 * D), BF16)
        ops.gemm(hh, w["w_fc"], aa, bias=w["b_fc"], act=ops.ACT_QUICKGELU)
        xo = ws.get("last.xo", (n_rows, D), RES)
        ops.gemm(aa, w["w_pr"], xo, bias=w["b_pr"], residual=x1)
        return xo

    def targets(self, rows: torch.Tensor, n_rows: int, slot: int = 0) -> torch.Tensor:
        cur = self._enter()
        try:
            return self._targets(rows, n_rows, slot)
        finally:
            self._leave(cur)

    def _targets(self, rows: torch.Tensor, n_rows: int, slot: int = 0) -> torch.Tensor:
        ws, D, C = self.ws, self.D, self.C
        K = len(self.taps)
        out = ws.get("targets" if slot == 0 else f"targets.{slot}", (K * n_rows, C), F32)
        xn = ws.get("tail.xn", (n_rows, D), BF16)
        for k in range(K):
            if self._last is not None and k == K - 1:          # the last block's tap exists for these rows only
                ops.layernorm_fwd(self._last_block_rows(rows, n_rows), self.ln_post[0], self.ln_post[1], self.eps, xn)
            else:
                ops.layernorm_fwd(self._tap_bufs[k], self.ln_post[0], self.ln_post[1], self.eps, xn, row_index=rows)
            ops.gemm(xn, self.proj_w, out[k * n_rows:(k + 1) * n_rows], trans_b=True)
        ops.l2_normalize_rows(out)
        return out


# ----------------------------------------------------------------------------- weights (reference clip.py:191-231)
def inflate_weight(weight_2d, time_dim, center=True):
    if center:
        weight_3d = torch.zeros(*weight_2d.shape).unsqueeze(2).repeat(1, 1, time_dim, 1, 1)
        weight_3d[:, :, time_dim // 2, :, :] = weight_2d
    else:
        weight_3d = weight_2d.unsqueeze(2).repeat(1, 1, time_dim, 1, 1) / time_dim
    return weight_3d


def load_state_dict(model, state_dict, input_resolution=224, patch_size=16, center=True):
    """OpenAI visual.* checkpoint -> this model: inflate 2-D conv weights, bicubic-resize the position table."""
    state_dict_3d = model.state_dict()
    for k in state_dict.keys():
        if k in state_dict_3d.keys() and state_dict[k].shape != state_dict_3d[k].shape:
            if len(state_dict_3d[k].shape) <= 2:
                continue
            state_dict[k] = inflate_weight(state_dict[k], state_dict_3d[k].shape[2], center=center)
    pos = state_dict['positional_embedding']
    emb = pos.shape[-1]
    num_patches = (input_resolution // patch_size) ** 2
    orig_size, new_size = int((pos.shape[-2] - 1) ** 0.5), int(num_patches ** 0.5)
    if orig_size != new_size:
        extra, tok = pos[:1], pos[1:]
        tok = tok.reshape(-1, orig_size, orig_size, emb).permute(0, 3, 1, 2)
        tok = torch.nn.functional.interpolate(tok, size=(new_size, new_size), mode='bicubic', align_corners=False)
        state_dict['positional_embedding'] = torch.cat((extra, tok.permute(0, 2, 3, 1).flatten(0, 2)), dim=0)
    model.load_state_dict(state_dict, strict=True)


def _build(name, pretrained, center, input_resolution, **kw):
    model = VisionTransformer(input_resolution=input_resolution, **kw)
    if pretrained:
        path = os.path.join(os.environ.get("UNITE_CLIP_PATH", MODEL_PATH), os.path.basename(_MODELS[name]))      # resolved at call time
        sd = torch.load(path, map_location='cpu', weights_only=True)
        load_state_dict(model, sd, input_resolution=input_resolution, patch_size=kw["patch_size"], center=center)
    return model.eval()


def clip_b16(pretrained=True, clip_norm_type='l2', input_resolution=224, kernel_size=1, return_attn=False, center=True,
             clip_return_layers=[6, 7, 8, 9, 10, 11], clip_return_interval=1, return_cls=False):
    return _build("ViT-B/16", pretrained, center, input_resolution, patch_size=16, width=768, layers=12, heads=12, output_dim=512,
                  clip_norm_type=clip_norm_type, kernel_size=kernel_size, return_attn=return_attn,
                  clip_return_layers=clip_return_layers, clip_return_interval=clip_return_interval, return_cls=return_cls)


def clip_l14(pretrained=True, clip_norm_type='l2', input_resolution=224, kernel_size=1, return_attn=False, center=True,
             clip_return_layers=[6, 7, 8, 9, 10, 11], clip_return_interval=1):
    return _build("ViT-L/14", pretrained, center, input_resolution, patch_size=14, width=1024, layers=24, heads=16, output_dim=768,
                  clip_norm_type=clip_norm_type, kernel_size=kernel_size, return_attn=return_attn,
                  clip_return_layers=clip_return_layers, clip_return_interval=clip_return_interval)


def clip_l14_336(pretrained=True, clip_norm_type='l2', input_resolution=336, kernel_size=1, return_attn=False, center=True,
                 clip_return_layers=[6, 7, 8, 9, 10, 11], clip_return_interval=1):
    return _build("ViT-L/14_336", pretrained, center, input_resolution, patch_size=14, width=1024, layers=24, heads=16, output_dim=768,
                  clip_norm_type=clip_norm_type, kernel_size=kernel_size, return_attn=return_attn,
                  clip_return_layers=clip_return_layers, clip_return_interval=clip_return_interval)
