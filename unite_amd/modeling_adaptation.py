"""Stage-1/3 student: ``AdaptationVisionTransformer`` on the gfx950 kernels.

Drop-in for reference src/models/modeling_adaptation.py: same class / factory names, constructor keywords,
``forward(x, mask, clip_only)`` contract and ``state_dict`` keys, so checkpoints and
``run_stage1.py``-style drivers interoperate.  The torch.nn modules below only HOLD parameters (and give them the
reference's initialisation); all device arithmetic is in libunite_hip.so through ``ViTRunner`` -- calling forward
on a CPU tensor raises, there is no eager fallback.

Differences from the reference, all internal:
  * only the visible tokens are patch-embedded (the reference embeds all 1568 and drops 80 %, :132,:153);
  * the sinusoid tables are device constants (the reference re-uploads them every forward, :144,:318);
  * bf16 GEMM operands / fp32 accumulation, residual stream and statistics in fp32, no GradScaler (SURVEY A-17).
Not built (raise NotImplementedError): use_cls_token=True, use_learnable_pos_emb=True, tubelet_size != 1,
head_dim != 64, use_checkpoint (no recomputation is needed with 288 GB of HBM: the flag is accepted and ignored).
"""
from __future__ import annotations

import math
from functools import partial
from typing import List, Optional, Sequence

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .flat_params import FlatParams
from .registry import register_model
from .vit_runner import ViTRunner, BF16, F32, SPLITK_WS_BYTES

DECODER_STREAM = 4        # index of the runner's side stream the decoders use (0-3: weight gradients)


def get_sinusoid_encoding_table(n_position: int, d_hid: int) -> torch.Tensor:
    """pos / 10000^(2*(j//2)/d), sin on even j, cos on odd j (reference modeling_adaptation.py:41-51); (1, n, d) f32."""
    j = np.arange(d_hid)
    table = np.arange(n_position, dtype=np.float64)[:, None] / np.power(10000.0, 2.0 * (j // 2) / d_hid)[None, :]
    table[:, 0::2] = np.sin(table[:, 0::2])
    table[:, 1::2] = np.cos(table[:, 1::2])
    return torch.tensor(table, dtype=torch.float32).unsqueeze(0)


# ----------------------------------------------------------------------------- parameter containers
class _Attention(nn.Module):
    def __init__(self, dim, qkv_bias):
        super().__init__()
        self.qkv = nn.Linear(dim, dim * 3, bias=False)
        if qkv_bias:
            self.q_bias = nn.Parameter(torch.zeros(dim))
            self.v_bias = nn.Parameter(torch.zeros(dim))
        else:
            raise NotImplementedError("qkv_bias=False is not built (every UNITE factory sets qkv_bias=True)")
        self.proj = nn.Linear(dim, dim)


class _Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)


class Block(nn.Module):
    """Parameter layout of reference Block (modeling_finetune.py:122-150), init_values = 0 branch."""

    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=True, drop_path=0., norm_layer=nn.LayerNorm):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = _Attention(dim, qkv_bias)
        self.norm2 = norm_layer(dim)
        self.mlp = _Mlp(dim, int(dim * mlp_ratio))
        self.drop_path_rate = drop_path


class PatchEmbed(nn.Module):
    """Parameter layout of reference PatchEmbed (modeling_finetune.py:153-175): Conv3d weight (D,3,tubelet,P,P)."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, num_frames=16, tubelet_size=2):
        super().__init__()
        self.img_size = (img_size, img_size)
        self.patch_size = (patch_size, patch_size)
        self.tubelet_size = int(tubelet_size)
        self.num_patches = (img_size // patch_size) ** 2 * (num_frames // self.tubelet_size)
        self.proj = nn.Conv3d(in_chans, embed_dim, kernel_size=(self.tubelet_size, patch_size, patch_size),
                              stride=(self.tubelet_size, patch_size, patch_size))


def _init_weights(m):
    # reference modeling_adaptation.py:108-115
    if isinstance(m, nn.Linear):
        nn.init.xavier_uniform_(m.weight)
        if m.bias is not None:
            nn.init.constant_(m.bias, 0)
    elif isinstance(m, nn.LayerNorm):
        nn.init.constant_(m.bias, 0)
        nn.init.constant_(m.weight, 1.0)


class AdaptationVisionTransformerEncoder(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=0, embed_dim=768, depth=12, num_heads=12,
                 mlp_ratio=4., qkv_bias=False, qk_scale=None, drop_rate=0., attn_drop_rate=0., drop_path_rate=0.,
                 norm_layer=nn.LayerNorm, init_values=None, num_frames=16, tubelet_size=2, use_checkpoint=False,
                 checkpoint_num=0, use_learnable_pos_emb=False, clip_return_layers=[6, 7, 8, 9, 10, 11],
                 clip_student_return_interval=1, use_cls_token=False):
        super().__init__()
        if use_cls_token or use_learnable_pos_emb:
            raise NotImplementedError("use_cls_token / use_learnable_pos_emb are not built (UNITE configs: both False)")
        if drop_rate or attn_drop_rate or (init_values or 0) > 0 or num_classes or qk_scale is not None:
            raise NotImplementedError("dropout, layer-scale, encoder head and qk_scale are unused by UNITE and not built")
        if in_chans != 3:
            raise NotImplementedError("in_chans must be 3")
        self.num_classes = num_classes
        self.num_features = self.embed_dim = embed_dim
        self.patch_embed = PatchEmbed(img_size, patch_size, in_chans, embed_dim, num_frames, tubelet_size)
        self.use_checkpoint, self.checkpoint_num = use_checkpoint, checkpoint_num
        self.return_index = list(clip_return_layers)
        self.use_learnable_pos_emb = False
        self.pos_embed = get_sinusoid_encoding_table(self.patch_embed.num_patches, embed_dim)    # plain tensor, not in state_dict
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, depth)]
        self.blocks = nn.ModuleList([Block(embed_dim, num_heads, mlp_ratio, qkv_bias, dpr[i], norm_layer) for i in range(depth)])
        self.norm = norm_layer(embed_dim)
        self.head = nn.Identity()
        self.num_heads, self.mlp_ratio = num_heads, mlp_ratio
        self.apply(_init_weights)

    def get_num_layers(self):
        return len(self.blocks)

    @torch.jit.ignore
    def no_weight_decay(self):
        return {'pos_embed', 'cls_token'}


class Linear_Decoder(nn.Module):
    """Parameter layout of reference Linear_Decoder (modeling_adaptation.py:182-213)."""

    def __init__(self, num_classes=768, embed_dim=768, norm_layer=nn.LayerNorm, clip_norm_type='l2'):
        super().__init__()
        if clip_norm_type != 'l2':
            raise NotImplementedError("clip_norm_type must be 'l2' (all UNITE configs)")
        self.clip_norm_type = clip_norm_type
        self.head = nn.Linear(embed_dim, num_classes)
        self.norm = norm_layer(num_classes)
        self.apply(_init_weights)


class AdaptationVisionTransformer(nn.Module):
    def __init__(self, img_size=224, patch_size=16, encoder_in_chans=3, encoder_num_classes=0, encoder_embed_dim=768,
                 encoder_depth=12, encoder_num_heads=12, mlp_ratio=4., qkv_bias=False, qk_scale=None, drop_rate=0.,
                 attn_drop_rate=0., drop_path_rate=0., norm_layer=nn.LayerNorm, init_values=0., use_learnable_pos_emb=False,
                 use_cls_token=False, use_checkpoint=False, checkpoint_num=0, num_frames=16, tubelet_size=2,
                 clip_decoder_embed_dim=768, clip_output_dim=512, clip_norm_type='l2',
                 clip_return_layers=[6, 7, 8, 9, 10, 11], clip_student_return_interval=1):
        super().__init__()
        if tubelet_size != 1:
            raise NotImplementedError("tubelet_size != 1 is not built (UNITE configs use 1)")
        if clip_decoder_embed_dim != encoder_embed_dim:
            raise ValueError("clip_decoder_embed_dim must equal the encoder width (the reference adds them, :320)")
        self.encoder = AdaptationVisionTransformerEncoder(
            img_size=img_size, patch_size=patch_size, in_chans=encoder_in_chans, num_classes=encoder_num_classes,
            embed_dim=encoder_embed_dim, depth=encoder_depth, num_heads=encoder_num_heads, mlp_ratio=mlp_ratio,
            qkv_bias=qkv_bias, qk_scale=qk_scale, drop_rate=drop_rate, attn_drop_rate=attn_drop_rate,
            drop_path_rate=drop_path_rate, norm_layer=norm_layer, init_values=init_values, num_frames=num_frames,
            tubelet_size=tubelet_size, use_checkpoint=use_checkpoint, checkpoint_num=checkpoint_num,
            use_learnable_pos_emb=use_learnable_pos_emb, clip_return_layers=clip_return_layers,
            clip_student_return_interval=clip_student_return_interval, use_cls_token=use_cls_token)
        self.clip_decoder = nn.ModuleList([
            Linear_Decoder(num_classes=clip_output_dim, embed_dim=clip_decoder_embed_dim, norm_layer=norm_layer,
                           clip_norm_type=clip_norm_type) for _ in range(len(clip_return_layers))])
        self.clip_pos_embed = get_sinusoid_encoding_table(self.encoder.patch_embed.num_patches, clip_decoder_embed_dim)
        self.ln_eps = self.encoder.norm.eps
        self.clip_output_dim = clip_output_dim
        self._rt: Optional["_StudentRuntime"] = None

    def get_num_layers(self):
        return len(self.encoder.blocks)

    @torch.jit.ignore
    def no_weight_decay(self):
        return {'pos_embed', 'cls_token', 'mask_token', 'clip_mask_token', 'clip_pos_embed'}

    # ------------------------------------------------------------------ runtime
    def runtime(self) -> "_StudentRuntime":
        """Flat parameter store + kernel schedule; created on first use, on the device the parameters live on."""
        if self._rt is None:
            dev = next(self.parameters()).device
            if dev.type != "cuda":
                raise RuntimeError("unite_amd models run on a MI355X only: move the model to 'cuda' first "
                                   "(there is no CPU path; the CPU oracle lives in oracle/ for tests)")
            self._rt = _StudentRuntime(self, dev)
        return self._rt

    def _apply(self, fn, *a, **k):
        if self._rt is not None:
            raise RuntimeError("model.to()/cuda()/float() after the first forward would detach the flat parameter buffer")
        return super()._apply(fn, *a, **k)

    def forward(self, x, mask, clip_only=False, vis_tokens=None, n_vis=None):
        """x (B,3,T,H,W) f32; mask bool (B, T*196), True = masked.  Returns x_clip (K,B,n_vis,C) if clip_only else
        (x_vis (B,n_vis,D), x_clip) -- reference :304-334.  ``vis_tokens`` (int32 [B*n_vis], ascending global token
        ids) may be given instead of ``mask`` to skip the device sync that reading n_vis out of a mask needs."""
        rt = self.runtime()
        if vis_tokens is None:
            vis_tokens, n_vis = rt.tokens_from_mask(mask)
        return _StudentFn.apply(self, x, vis_tokens, n_vis, clip_only, rt.grad_anchor)

    def forward_loss(self, x, vis_tokens, n_vis, targets, targets_ready=None):
        """Fused stage-1 objective: mean(2 - 2 <decoder(x), targets>) over (K,B,n_vis) (run_stage1.py:431) without
        materialising x_clip.  targets: f32 [K*B*n_vis, C] rows in (k, b, token) order, L2-normalised.  targets_ready: optional
        event after which `targets` is valid (the teacher's tail may still be running on another stream during the encoder)."""
        rt = self.runtime()
        rt.targets_ready = targets_ready
        return _StudentLossFn.apply(self, x, vis_tokens, n_vis, targets, rt.grad_anchor)


class _StudentRuntime:
    def __init__(self, model: AdaptationVisionTransformer, dev):
        enc = model.encoder
        self.model = model
        self.fp = FlatParams(model, dev)
        self.dev = dev
        D = enc.embed_dim
        self.D, self.C = D, model.clip_output_dim
        self.taps: List[int] = list(enc.return_index)
        self.depth = len(enc.blocks)
        self.T_N = enc.patch_embed.num_patches
        self.pos = enc.pos_embed[0].to(dev).contiguous()
        self.clip_pos = model.clip_pos_embed[0].to(dev).contiguous()
        self.runner = ViTRunner(self.fp, "encoder.", D, self.depth, enc.num_heads, int(D * enc.mlp_ratio), model.ln_eps,
                                enc.patch_embed.patch_size[0], self.T_N, self.pos,
                                [b.drop_path_rate for b in enc.blocks])
        self.ws = self.runner.ws
        # autograd anchor: gradients are written into the flat buffer by the backward itself; the anchor only makes
        # autograd call it (loss.backward() works as in the reference engine, utils.py:609)
        self.grad_anchor = torch.zeros((), device=dev, requires_grad=True)
        fp = self.fp
        idx = {n: i for i, n in enumerate(fp.names)}
        self.norm_w, self.norm_b = fp.params[idx["encoder.norm.weight"]].data, fp.params[idx["encoder.norm.bias"]].data
        self.g_norm_w, self.g_norm_b = fp.g("encoder.norm.weight"), fp.g("encoder.norm.bias")
        self.dec = []
        for k in range(len(self.taps)):
            p = f"clip_decoder.{k}."
            self.dec.append(dict(w=fp.w16(p + "head.weight"), b=fp.params[idx[p + "head.bias"]].data,
                                 gw=fp.g(p + "head.weight"), gb=fp.g(p + "head.bias"),
                                 nw=fp.params[idx[p + "norm.weight"]].data, nb=fp.params[idx[p + "norm.bias"]].data,
                                 gnw=fp.g(p + "norm.weight"), gnb=fp.g(p + "norm.bias")))
        self.layer_done_hook = None      # set by the data-parallel reducer
        self.frame_tokens = (enc.patch_embed.img_size[0] // enc.patch_embed.patch_size[0]) ** 2

    def tag_ranges(self):
        """(tag, start, end) flat-buffer ranges of the parameter layers in backward-completion order (ddp.GradReducer)."""
        from .ddp import student_tag_ranges
        return student_tag_ranges(self)

    # -- mask -> token list (device sync: the number of visible tokens has to reach the host, as in x[~mask] of the reference)
    def tokens_from_mask(self, mask: torch.Tensor):
        B, L = mask.shape
        m8 = mask.to(device=self.dev).to(torch.uint8).contiguous()
        n_vis = int((m8[0] == 0).sum().item())
        N = self.frame_tokens
        T = L // N
        if n_vis % T:
            raise NotImplementedError("masks must keep the same number of tokens in every frame (attention masking does)")
        vis = torch.empty(B * n_vis, dtype=torch.int32, device=self.dev)
        ops.mask_to_tokens(m8.view(-1), vis, n_vis // T, B * T, N)
        return vis, n_vis

    # -- encoder only (stage 3): x_vis = encoder.norm(blocks(x)) for the listed tokens, in its own activation slot
    def encode(self, videos, vis_tokens, n_vis, slot: str, training: bool, save: bool = True):
        """reference modeling_adaptation.py:171-179 with the Identity head: returns x_vis f32 [B*n_vis, D].  vis_tokens = None
        keeps every token (the 'full_vis_mask' passes of run_stage3.py:468-483)."""
        fp, r = self.fp, self.runner
        r.use_slot(slot)
        ws = self.ws
        fp.refresh_if_stale()
        fp.unused_prefixes = ("clip_decoder.",)      # the encoder-only passes never reach the decoders: no gradient for them (run_stage3.py:475)
        B = videos.shape[0]
        M, D = B * n_vis, self.D
        dp = r.drop_path_scales(B, training)
        x0 = r.embed(videos, vis_tokens, M)
        xs = r.blocks_forward(x0, B, n_vis, self.depth, dp, save=save)
        xv = ws.get("enc.xv", (M, D), F32)
        mean, rstd = ws.get("enc.mean", (M,), F32), ws.get("enc.rstd", (M,), F32)
        ops.layernorm_fwd(xs[-1], self.norm_w, self.norm_b, self.model.ln_eps, xv, mean=mean, rstd=rstd)
        self._enc = getattr(self, "_enc", {})
        self._enc[slot] = dict(B=B, n_vis=n_vis, M=M, dp=dp, x_last=xs[-1], mean=mean, rstd=rstd, saved=save)
        r.use_slot("")
        return xv

    def encode_backward(self, dxv, slot: str, notify: bool = True):
        """gradient of an encode() pass from d(x_vis) (f32 [M, D]); parameter gradients follow fp.accumulate.  notify=False keeps
        the gradient reducer quiet (a further pass will still add to the same buckets)."""
        fp, r = self.fp, self.runner
        c = self._enc[slot]
        if not c["saved"]:
            raise RuntimeError("encode(..., save=False) keeps no activations")
        r.use_slot(slot)
        ws = self.ws
        M, D, N = c["M"], self.D, c["n_vis"]
        acc = fp.accumulate
        lnws = ws.bytes_("ln.ws", ops.layernorm_bwd_workspace(M, max(D, self.C)))
        last = self.depth - 1
        dx = ws.get("bw.dxtop", (M, D), F32)
        dxb = ws.get("bw.dxtopb", (M, D), BF16)
        ops.layernorm_bwd(dxv, c["x_last"], c["mean"], c["rstd"], self.norm_w, dx_out=dx, dx_bf16=dxb,
                          row_scale=None if c["dp"] is None else c["dp"][last, 1], rows_per_scale=N,
                          dgamma=self.g_norm_w, dbeta=self.g_norm_b, dxsum=r._blk[last]["g:mlp.fc2.bias"], accumulate=acc, workspace=lnws)
        hook = self.layer_done_hook if notify else None
        if hook is not None:
            hook("norm")
        dx0, dx0b = r.blocks_backward(dx, dxb, self.depth, layer_done=hook)
        r.embed_backward(dx0b)
        if hook is not None:
            hook("patch_embed")
        r.use_slot("")
        fp.accumulate = True
        fp.ensure_grad_views()

    # -- forward up to the decoder pre-activations
    def forward_features(self, videos, vis_tokens, n_vis, clip_only, training, tail=None, tail_wait=None):
        """``tail(k, y_k)`` runs right behind decoder k's projection on whatever stream that ran on (the fused normalise + loss kernel of
        stage 1); ``tail_wait``: an event the tails have to wait for (the targets)."""
        fp, r, ws = self.fp, self.runner, self.ws
        fp.refresh_if_stale()
        B = videos.shape[0]
        M = B * n_vis
        D, C = self.D, self.C
        n_blocks = (max(self.taps) + 1) if clip_only else self.depth       # early break, reference :165-166
        # blocks above the highest tap are not executed: their parameters get no gradient (p.grad stays None in the reference, so
        # AdamW and the gradient norm skip them); the optimizer reads this list
        fp.unused_prefixes = tuple(f"encoder.blocks.{i}." for i in range(n_blocks, self.depth))
        dp = r.drop_path_scales(B, training)
        x0 = r.embed(videos, vis_tokens, M)
        cpos = ws.get("dec.pos", (M, D), F32)
        ops.gather_rows(self.clip_pos, vis_tokens, cpos, modulo=self.T_N)
        K = len(self.taps)
        ys, taps = [None] * K, [None] * K
        # A decoder reads one tap and nothing reads the decoder before the loss: the decoders of the taps below the last block run on a
        # side stream while the main stream goes on with the next encoder blocks (0.3 ms of the student's serial chain at B = 32).
        side = r._side_stream(DECODER_STREAM) if (x0.is_cuda and r.side_decoders) else None
        main = torch.cuda.current_stream() if side is not None else None
        if side is not None and tail_wait is not None:
            side.wait_event(tail_wait)
        used_side = [False]

        def decoder(k, xt):
            xn = ws.get(f"dec.xn{k}", (M, D), BF16)
            mean, rstd = ws.get(f"dec.mean{k}", (M,), F32), ws.get(f"dec.rstd{k}", (M,), F32)
            ops.layernorm_fwd(xt, self.norm_w, self.norm_b, self.model.ln_eps, xn, post_add=cpos, mean=mean, rstd=rstd)
            y = ws.get(f"dec.y{k}", (M, C), F32)
            d = self.dec[k]
            ops.gemm(xn, d["w"], y, bias=d["b"])
            ys[k] = y
            taps[k] = dict(x=xt, xn=xn, mean=mean, rstd=rstd, y=y)
            if tail is not None:
                tail(k, y)

        def after_block(i, xt):
            if i not in self.taps:
                return
            k = self.taps.index(i)
            if side is None or i == n_blocks - 1:      # nothing left to run beside it
                decoder(k, xt)
                return
            ev = torch.cuda.Event()
            ev.record(main)
            side.wait_event(ev)
            with torch.cuda.stream(side):
                decoder(k, xt)
            used_side[0] = True

        xs = r.blocks_forward(x0, B, n_vis, n_blocks, dp, save=True, after_block=after_block)
        for k, li in enumerate(self.taps):           # taps above the last executed block (none in the shipped configs)
            if ys[k] is None:
                decoder(k, xs[li + 1])
        if used_side[0]:
            ev = torch.cuda.Event()
            ev.record(side)
            main.wait_event(ev)
        self._tap = taps
        self._ctx = dict(B=B, n_vis=n_vis, M=M, n_blocks=n_blocks, xs=xs, dp=dp, clip_only=clip_only)
        return ys, xs

    # -- backward from the decoder pre-activation gradients dy_k (bf16 [M,C])
    def backward_from_dy(self, dxv: Optional[torch.Tensor] = None, tail_bwd=None):
        """dxv: optional f32 [M,D] gradient w.r.t. the returned x_vis = encoder.norm(x_out(last)) (clip_only=False).
        ``tail_bwd(k, lnws) -> dy_k`` launches the backward of decoder k's tail (norm / loss) on the current stream with the given
        LayerNorm workspace; without it the dy_k are taken as already computed into ws "dec.dy{k}" on the current stream."""
        fp, r, ws, ctx = self.fp, self.runner, self.ws, self._ctx
        M, D, C, N = ctx["M"], self.D, self.C, ctx["n_vis"]
        acc = fp.accumulate
        dp = ctx["dp"]
        lnws = ws.bytes_("ln.ws", ops.layernorm_bwd_workspace(M, max(D, C)))
        csws = ws.bytes_("cs.ws", ops.colsum_workspace(M, max(r.Hd, 3 * D)))
        gws = ws.bytes_("gemm.ws", SPLITK_WS_BYTES)
        last = ctx["n_blocks"] - 1
        K = len(self.taps)

        # decoder k: tail backward -> dgrad (gradient of the normalised tap, bf16) -> wgrad / bias into the flat buffer.  Only the decoder
        # of the last executed block is needed at once; the others run on the decoders' side stream while the main stream is in the
        # encoder blocks above their tap, each joined where its tap's gradient enters the chain (tap_grad).
        def dec_dgrad(k, lnws_k):
            d = self.dec[k]
            dy = tail_bwd(k, lnws_k) if tail_bwd is not None else ws.peek(f"dec.dy{k}")
            dxn = ws.get(f"dec.dxn{k}", (M, D), BF16)
            ops.gemm(dy, d["w"], dxn, trans_b=True)
            return dy

        def dec_wgrad(k, dy, gws_k):
            ops.gemm(dy, self._tap[k]["xn"], self.dec[k]["gw"], trans_a=True, trans_b=True, accumulate=acc, workspace=gws_k)   # head bias grad: dysum of the tail

        order = sorted(range(K), key=lambda k: -self.taps[k])
        side = r._side_stream(DECODER_STREAM) if (fp.device.type == "cuda" and r.side_decoders and K > 1) else None
        dec_ev, dec_all = {}, None
        if side is None:
            for k in order:
                dec_wgrad(k, dec_dgrad(k, lnws), gws)
            if self.layer_done_hook is not None:
                self.layer_done_hook("clip_decoder")
        else:
            main = torch.cuda.current_stream()
            lnws_d = ws.bytes_("ln.ws.dec", ops.layernorm_bwd_workspace(M, max(D, C)))
            gws_d = ws.bytes_("gemm.ws.dec", SPLITK_WS_BYTES)
            ev0 = torch.cuda.Event()
            ev0.record(main)
            side.wait_event(ev0)
            first = order[0] if self.taps[order[0]] == last else None
            dy_first, ev_first = None, None
            if first is not None:
                dy_first = dec_dgrad(first, lnws)
                ev_first = torch.cuda.Event()
                ev_first.record(main)
            with torch.cuda.stream(side):
                for k in order:
                    if k == first:
                        continue
                    dy = dec_dgrad(k, lnws_d)
                    dec_ev[k] = torch.cuda.Event()
                    dec_ev[k].record(side)
                    dec_wgrad(k, dy, gws_d)
                if first is not None:
                    side.wait_event(ev_first)
                    dec_wgrad(first, dy_first, gws_d)
                dec_all = torch.cuda.Event()
                dec_all.record(side)
            if self.layer_done_hook is not None:
                ev_m = torch.cuda.Event()
                ev_m.record(main)
                self.layer_done_hook("clip_decoder", (ev_m, dec_all))
        norm_first = [True]

        def tap_grad(li, dx_in, scale, dxsum):
            """adds d/dx_out(li) of the shared encoder.norm branch of tap li; returns (dx f32, dx bf16 * scale) and writes the
            column sums of the bf16 copy (= fc2 bias gradient of block li) into dxsum."""
            k = self.taps.index(li)
            t = self._tap[k]
            if k in dec_ev:
                torch.cuda.current_stream().wait_event(dec_ev.pop(k))      # decoder k's dgrad ran on the side stream
            out = ws.get(f"bw.dxt{li & 1}", (M, D), F32)
            outb = ws.get(f"bw.dxtb{li % 3}", (M, D), BF16)      # rotation of 3: read by block li's side-stream weight gradients
            ops.layernorm_bwd(ws.peek(f"dec.dxn{k}"), t["x"], t["mean"], t["rstd"], self.norm_w, dx_residual=dx_in, dx_out=out,
                              dx_bf16=outb, row_scale=scale, rows_per_scale=N, workspace=lnws,
                              dgamma=self.g_norm_w, dbeta=self.g_norm_b, dxsum=dxsum,
                              # encoder.norm's gamma/beta ADD over the taps (bit 0); the bias column sums follow zero_grad (bit 1)
                              accumulate=(1 if (acc or not norm_first[0]) else 0) | (2 if acc else 0))
            norm_first[0] = False
            if li == min(self.taps) and self.layer_done_hook is not None:
                self.layer_done_hook("norm")      # encoder.norm's gradient is complete after the lowest tap
            return out, outb

        dx_top = None
        if dxv is not None:      # x_vis = encoder.norm(x_out(depth-1)) was returned and has an upstream gradient
            dx_top = ws.get("bw.dxv", (M, D), F32)
            xl = ctx["xs"][last + 1]
            mean, rstd = ws.get("vis.mean", (M,), F32), ws.get("vis.rstd", (M,), F32)
            ops.layernorm_bwd(dxv, xl, mean, rstd, self.norm_w, dx_out=dx_top, dgamma=self.g_norm_w, dbeta=self.g_norm_b,
                              accumulate=acc, workspace=lnws)
            norm_first[0] = False
        s_last = None if dp is None else dp[last, 1]
        if last in self.taps:
            dx, dxb = tap_grad(last, dx_top, s_last, r._blk[last]["g:mlp.fc2.bias"])
        else:
            raise NotImplementedError("the last executed block must be a tap or carry an x_vis gradient")

        def hook(li, dx_in, scale, dxsum):
            return tap_grad(li, dx_in, scale, dxsum)

        def done(i, events=None):
            if self.layer_done_hook is not None:
                self.layer_done_hook(i, events)

        dx0, dx0b = r.blocks_backward(dx, dxb, ctx["n_blocks"], tap_layers=set(self.taps), tap_hook=hook, layer_done=done)
        r.embed_backward(dx0b)
        if dec_all is not None:
            torch.cuda.current_stream().wait_event(dec_all)      # the decoders' weight gradients are written before anything reads the flat buffer
        if self.layer_done_hook is not None:
            self.layer_done_hook("patch_embed")
        fp.accumulate = True        # a second backward before zero_grad() adds, as autograd would
        fp.ensure_grad_views()      # p.grad may have been set to None by nn.Module.zero_grad()


class _StudentFn(torch.autograd.Function):
    """model(x, mask[, clip_only]) with autograd: returns the decoder outputs; backward takes d(x_clip)."""

    @staticmethod
    def forward(ctx, model, videos, vis_tokens, n_vis, clip_only, anchor):
        ops.keep_plan(ctx)
        rt = model.runtime()
        ys, xs = rt.forward_features(videos, vis_tokens, n_vis, clip_only, model.training)
        B, M, C, K = videos.shape[0], videos.shape[0] * n_vis, rt.C, len(rt.taps)
        out = torch.empty(K, B, n_vis, C, dtype=F32, device=rt.dev)
        for k in range(K):
            d = rt.dec[k]
            ops.decoder_tail_fwd(ys[k], d["nw"], d["nb"], model.ln_eps, None, out[k].view(M, C), None)
        ctx.model, ctx.clip_only = model, clip_only
        if clip_only:
            return out
        xv = torch.empty(B, n_vis, rt.D, dtype=F32, device=rt.dev)
        mean, rstd = rt.ws.get("vis.mean", (M,), F32), rt.ws.get("vis.rstd", (M,), F32)
        ops.layernorm_fwd(xs[-1], rt.norm_w, rt.norm_b, model.ln_eps, xv.view(M, rt.D), mean=mean, rstd=rstd)
        return xv, out

    @staticmethod
    def backward(ctx, *grads):
        with ops.kept_plan(ctx):
            model = ctx.model
            rt = model.runtime()
            c = rt._ctx
            M, C = c["M"], rt.C
            dout = grads[0] if ctx.clip_only else grads[1]
            dxv = None if ctx.clip_only else grads[0]
            dout_c = None if dout is None else dout.contiguous()
            acc = rt.fp.accumulate

            def tail_bwd(k, lnws):
                d = rt.dec[k]
                dy = rt.ws.get(f"dec.dy{k}", (M, C), BF16)
                if dout_c is None:
                    dy.zero_()
                else:
                    ops.decoder_tail_bwd(rt._tap[k]["y"], d["nw"], d["nb"], model.ln_eps, None, 0.0, dout_c[k].view(M, C), dy,
                                         d["gnw"], d["gnb"], lnws, accumulate=acc, dysum=d["gb"])
                return dy

            rt.backward_from_dy(None if dxv is None else dxv.contiguous().view(M, rt.D), tail_bwd=tail_bwd)
            return None, None, None, None, None, None


class _StudentLossFn(torch.autograd.Function):
    """Fused decoder tail + UMT loss (run_stage1.py:431); backward needs no d(x_clip) tensor."""

    @staticmethod
    def forward(ctx, model, videos, vis_tokens, n_vis, targets, anchor):
        ops.keep_plan(ctx)
        rt = model.runtime()
        M, C, K = videos.shape[0] * n_vis, rt.C, len(rt.taps)
        loss_sum = rt.ws.get("loss.sum", (1,), F32)
        loss_sum.zero_()
        ev = getattr(rt, "targets_ready", None)      # the targets may still be on their way on the teacher's side stream
        rt.targets_ready = None
        tg = targets.view(K, M, C)
        waited = [False]

        def tail(k, y):
            # normalise + loss of decoder k right behind its projection (same stream); the partial sums meet in loss_sum by float atomics
            if ev is not None and not waited[0] and torch.cuda.current_stream() == main:
                main.wait_event(ev)
                waited[0] = True
            d = rt.dec[k]
            ops.decoder_tail_fwd(y, d["nw"], d["nb"], model.ln_eps, tg[k], None, loss_sum)

        main = torch.cuda.current_stream() if videos.is_cuda else None
        rt.forward_features(videos, vis_tokens, n_vis, True, model.training, tail=tail, tail_wait=ev)
        ctx.model, ctx.targets = model, tg
        return loss_sum[0] / float(K * M)

    @staticmethod
    def backward(ctx, gloss):
        with ops.kept_plan(ctx):
            model = ctx.model
            rt = model.runtime()
            c = rt._ctx
            M, C, K = c["M"], rt.C, len(rt.taps)
            g = gloss.contiguous().to(F32)
            acc, targets = rt.fp.accumulate, ctx.targets

            def tail_bwd(k, lnws):
                d = rt.dec[k]
                dy = rt.ws.get(f"dec.dy{k}", (M, C), BF16)
                ops.decoder_tail_bwd(rt._tap[k]["y"], d["nw"], d["nb"], model.ln_eps, targets[k], 1.0 / float(K * M), None, dy,
                                     d["gnw"], d["gnb"], lnws, accumulate=acc, loss_scale_dev=g, dysum=d["gb"])
                return dy

            rt.backward_from_dy(None, tail_bwd=tail_bwd)
            return None, None, None, None, None, None


# ----------------------------------------------------------------------------- factories (reference :337-378)
@register_model
def adaptation_umt_base_patch16_224(pretrained=False, **kwargs):
    kwargs.pop("drop_block_rate", None)
    init_ckpt = kwargs.pop("init_ckpt", None)
    model = AdaptationVisionTransformer(img_size=224, patch_size=16, encoder_embed_dim=768, encoder_depth=12,
                                        encoder_num_heads=12, encoder_num_classes=0, mlp_ratio=4, qkv_bias=True,
                                        norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)
    if pretrained:
        model.load_state_dict(torch.load(init_ckpt, map_location="cpu", weights_only=True)["model"])
    return model


@register_model
def adaptation_umt_large_patch16_224(pretrained=False, **kwargs):
    kwargs.pop("drop_block_rate", None)
    init_ckpt = kwargs.pop("init_ckpt", None)
    model = AdaptationVisionTransformer(img_size=224, patch_size=16, encoder_embed_dim=1024, encoder_depth=24,
                                        encoder_num_heads=16, encoder_num_classes=0, mlp_ratio=4, qkv_bias=True,
                                        norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)
    if pretrained:
        model.load_state_dict(torch.load(init_ckpt, map_location="cpu", weights_only=True)["model"])
    return model
