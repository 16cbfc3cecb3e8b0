"""Stage-2 video classifier: ``VisionTransformer`` on the gfx950 kernels -- drop-in for the reference's
src/models/modeling_finetune.py:237-419 (same class / factory names, constructor keywords, ``forward(x) -> (B, classes)``,
``state_dict`` keys: patch_embed.*, blocks.N.*, fc_norm.*, head.*).

All tokens are kept (1568 at 8 frames, 3136 at 16 frames): blocks run through ``ViTRunner`` with the tiled flash-style
attention kernels (the reference materialises a (B,H,N,N) probability tensor per layer, :111-114), then
``fc_norm(mean over tokens)`` (:374-376) and the linear head (:382) in fp32.
Not built (raise NotImplementedError): use_mean_pooling=False (CLS token), classifier_type='mlp', learnable pos-emb,
dropout > 0, tubelet_size != 1, head_dim != 64.
"""
from __future__ import annotations

from functools import partial
from typing import Optional

import torch
import torch.nn as nn

from . import ops
from .flat_params import FlatParams
from .modeling_adaptation import Block, PatchEmbed, get_sinusoid_encoding_table
from .registry import register_model
from .vit_runner import ViTRunner, BF16, F32


def _trunc_normal_(t, std=.02):
    return nn.init.trunc_normal_(t, std=std)


class VisionTransformer(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dim=768, depth=12, num_heads=12,
                 mlp_ratio=4., qkv_bias=False, qk_scale=None, fc_drop_rate=0., drop_rate=0., attn_drop_rate=0., drop_path_rate=0.,
                 norm_layer=nn.LayerNorm, init_values=0., use_learnable_pos_emb=False, init_scale=0., all_frames=16, tubelet_size=2,
                 use_checkpoint=False, checkpoint_num=0, use_mean_pooling=True, classifier_type='linear', classifier_hidden_dim=256):
        super().__init__()
        if not use_mean_pooling or classifier_type != 'linear' or use_learnable_pos_emb:
            raise NotImplementedError("only use_mean_pooling=True, classifier_type='linear', sinusoid positions are built (UNITE stage-2 config)")
        if fc_drop_rate or drop_rate or attn_drop_rate or (init_values or 0) > 0 or qk_scale is not None or tubelet_size != 1 or num_classes <= 0:
            raise NotImplementedError("dropout, layer-scale, qk_scale, tubelet_size != 1 and num_classes = 0 are not built")
        self.num_classes = num_classes
        self.num_features = self.embed_dim = embed_dim
        self.tubelet_size = tubelet_size
        self.patch_embed = PatchEmbed(img_size, patch_size, in_chans, embed_dim, all_frames, tubelet_size)
        num_patches = self.patch_embed.num_patches
        self.use_checkpoint, self.checkpoint_num = use_checkpoint, checkpoint_num
        self.classifier_type = classifier_type
        pre_n_position = 2048 if patch_size == 14 else num_patches                      # reference :289-297
        self.pos_embed = get_sinusoid_encoding_table(pre_n_position, embed_dim)           # plain tensor, not in state_dict
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, depth)]
        self.blocks = nn.ModuleList([Block(embed_dim, num_heads, mlp_ratio, qkv_bias, dpr[i], norm_layer) for i in range(depth)])
        self.norm = nn.Identity()
        self.fc_norm = norm_layer(embed_dim)
        self.fc_dropout = nn.Identity()
        self.head = nn.Linear(embed_dim, num_classes)
        self.num_heads, self.mlp_ratio = num_heads, mlp_ratio
        self.apply(self._init_weights)
        self.head.weight.data.mul_(init_scale)                                            # reference :325-327
        self.head.bias.data.mul_(init_scale)
        self._rt: Optional[_VitRuntime] = None

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            _trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def get_num_layers(self):
        return len(self.blocks)

    @torch.jit.ignore
    def no_weight_decay(self):
        return {'pos_embed', 'cls_token'}

    def get_classifier(self):
        return self.head

    def runtime(self) -> "_VitRuntime":
        if self._rt is None:
            dev = next(self.parameters()).device
            if dev.type != "cuda":
                raise RuntimeError("unite_amd models run on a MI355X only: move the model to 'cuda' first (no CPU path)")
            self._rt = _VitRuntime(self, dev)
        return self._rt

    def _apply(self, fn, *a, **k):
        if self._rt is not None:
            raise RuntimeError("model.to()/cuda() after the first forward would detach the flat parameter buffer")
        return super()._apply(fn, *a, **k)

    def forward(self, x):
        """x (B,3,T,H,W) f32 -> logits (B, num_classes) f32 (reference :380-383)."""
        rt = self.runtime()
        return _VitFn.apply(self, x, rt.grad_anchor)

    def forward_loss(self, x, targets, loss_scale: float = 1.0):
        """fused mean cross-entropy (run_stage2.py:681, no label smoothing / mixup): returns (loss, logits)."""
        rt = self.runtime()
        return _VitLossFn.apply(self, x, targets, float(loss_scale), rt.grad_anchor)


class _VitRuntime:
    def __init__(self, model: VisionTransformer, dev):
        self.model, self.dev = model, dev
        self.fp = FlatParams(model, dev)
        D = model.embed_dim
        self.D, self.C, self.depth = D, model.num_classes, len(model.blocks)
        self.N = model.patch_embed.num_patches
        self.eps = model.fc_norm.eps
        self.pos = model.pos_embed[0, :self.N].to(dev).contiguous()
        self.runner = ViTRunner(self.fp, "", D, self.depth, model.num_heads, int(D * model.mlp_ratio), self.eps,
                                model.patch_embed.patch_size[0], self.N, self.pos, [b.drop_path_rate for b in model.blocks])
        self.ws = self.runner.ws
        self.grad_anchor = torch.zeros((), device=dev, requires_grad=True)
        fp = self.fp
        idx = {n: i for i, n in enumerate(fp.names)}
        P = lambda n: fp.params[idx[n]].data
        self.fcn_w, self.fcn_b, self.head_w, self.head_b = P("fc_norm.weight"), P("fc_norm.bias"), P("head.weight"), P("head.bias")
        self.g_fcn_w, self.g_fcn_b = fp.g("fc_norm.weight"), fp.g("fc_norm.bias")
        self.g_head_w, self.g_head_b = fp.g("head.weight"), fp.g("head.bias")
        self.layer_done_hook = None

    def tag_ranges(self):
        """(tag, start, end) flat-buffer ranges in backward-completion order: head + fc_norm, blocks depth-1 .. 0, patch embed
        (flat order = named_parameters() order: patch_embed, blocks.*, fc_norm, head)."""
        fp = self.fp
        (nlo, _), (_, hhi) = fp.layer_ranges(["fc_norm.", "head."])
        out = [("head", nlo, hhi)]
        for i in reversed(range(self.depth)):
            (lo, hi), = fp.layer_ranges([f"blocks.{i}."])
            out.append((i, lo, hi))
        (lo, hi), = fp.layer_ranges(["patch_embed."])
        out.append(("patch_embed", lo, hi))
        return out

    def forward_logits(self, videos, training):
        fp, r, ws = self.fp, self.runner, self.ws
        fp.refresh_if_stale()
        B = videos.shape[0]
        N, D, C = self.N, self.D, self.C
        if videos.shape[2] * (videos.shape[3] // r.P) * (videos.shape[4] // r.P) != N:
            raise ValueError("clip shape does not match the model's token count")
        dp = r.drop_path_scales(B, training)
        x0 = r.embed(videos, None, B * N)
        xs = r.blocks_forward(x0, B, N, self.depth, dp, save=training)
        pooled = ws.get("cls.pooled", (B, D), F32)
        ops.token_mean_fwd(xs[-1].view(B, N, D), pooled)                       # x.mean(1), reference :376
        feat = ws.get("cls.feat", (B, D), F32)
        mean, rstd = ws.get("cls.mean", (B,), F32), ws.get("cls.rstd", (B,), F32)
        ops.layernorm_fwd(pooled, self.fcn_w, self.fcn_b, self.eps, feat, mean=mean, rstd=rstd)
        logits = torch.empty(B, C, dtype=F32, device=self.dev)
        ops.linear_f32_fwd(feat, self.head_w, self.head_b, logits)
        self._ctx = dict(B=B, dp=dp, pooled=pooled, feat=feat, mean=mean, rstd=rstd, x_last=xs[-1])
        return logits

    def backward_from_dlogits(self, dlogits):
        fp, r, ws, c = self.fp, self.runner, self.ws, self._ctx
        B, N, D = c["B"], self.N, self.D
        M = B * N
        acc = fp.accumulate
        dfeat = ws.get("cls.dfeat", (B, D), F32)
        ops.linear_f32_bwd(c["feat"], self.head_w, dlogits, dx=dfeat, dW=self.g_head_w, db=self.g_head_b, accumulate=acc)
        lnws = ws.bytes_("ln.ws", ops.layernorm_bwd_workspace(M, D))
        dpool = ws.get("cls.dpool", (B, D), F32)
        ops.layernorm_bwd(dfeat, c["pooled"], c["mean"], c["rstd"], self.fcn_w, dx_out=dpool, dgamma=self.g_fcn_w, dbeta=self.g_fcn_b,
                          accumulate=acc, workspace=lnws)
        if self.layer_done_hook is not None:
            self.layer_done_hook("head")
        # d/dx of the token mean: every token of a clip receives dpool[b] / N
        dx = ws.get("bw.dxtop", (M, D), F32)
        ops.token_mean_bwd(dpool, dx.view(B, N, D))
        # bf16 copy for the last block's fc2 GEMMs (scaled by its drop-path factor) + its column sums (fc2 bias gradient):
        # an identity "LayerNorm backward" is not available, so cast through the GEMM-free path: gamma = 1, rstd = 1, mean = 0
        dxb = ws.get("bw.dxtopb", (M, D), BF16)
        last = self.depth - 1
        scale = None if c["dp"] is None else c["dp"][last, 1]
        ops.scale_cast_colsum(dx, dxb, r._blk[last]["g:mlp.fc2.bias"], ws.bytes_("cs.ws", ops.colsum_workspace(M, max(r.Hd, 3 * D))),
                              row_scale=scale, rows_per_scale=N, accumulate=acc)
        done = (lambda i, events=None: self.layer_done_hook(i, events)) if self.layer_done_hook is not None else None
        dx0, dx0b = r.blocks_backward(dx, dxb, self.depth, layer_done=done)
        r.embed_backward(dx0b)
        if self.layer_done_hook is not None:
            self.layer_done_hook("patch_embed")
        fp.accumulate = True
        fp.ensure_grad_views()


class _VitFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, videos, anchor):
        ops.keep_plan(ctx)
        ctx.model = model
        return model.runtime().forward_logits(videos, model.training)

    @staticmethod
    def backward(ctx, dlogits):
        with ops.kept_plan(ctx):
            ctx.model.runtime().backward_from_dlogits(dlogits.contiguous().to(F32))
            return None, None, None


class _VitLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, videos, targets, loss_scale, anchor):
        ops.keep_plan(ctx)
        rt = model.runtime()
        logits = rt.forward_logits(videos, model.training)
        B = logits.shape[0]
        loss_sum = rt.ws.get("cls.loss", (1,), F32)
        loss_sum.zero_()
        dlog = rt.ws.get("cls.dlogits", tuple(logits.shape), F32)
        ops.softmax_ce(logits, targets, loss_sum, dlog, grad_scale=loss_scale / B)      # mean CE, gradient kept for backward
        ctx.model = model
        ctx.mark_non_differentiable(logits)
        return loss_sum[0] * (loss_scale / B), logits

    @staticmethod
    def backward(ctx, gloss, _glogits):
        with ops.kept_plan(ctx):
            rt = ctx.model.runtime()
            dlog = rt.ws.bufs["cls.dlogits"]
            rt.backward_from_dlogits(dlog * gloss)          # gloss is 1 unless the caller scaled the loss again
            return None, None, None, None, None


def _factory(embed_dim, depth, num_heads, img_size=224, **kwargs):
    return VisionTransformer(img_size=img_size, patch_size=16, embed_dim=embed_dim, depth=depth, num_heads=num_heads, mlp_ratio=4,
                             qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)


@register_model
def vit_base_patch16_224(pretrained=False, **kwargs):
    return _factory(768, 12, 12, **kwargs)


@register_model
def vit_base_patch16_384(pretrained=False, **kwargs):
    return _factory(768, 12, 12, img_size=384, **kwargs)


@register_model
def vit_large_patch16_224(pretrained=False, **kwargs):
    return _factory(1024, 24, 16, **kwargs)


@register_model
def vit_large_patch16_384(pretrained=False, **kwargs):
    return _factory(1024, 24, 16, img_size=384, **kwargs)
