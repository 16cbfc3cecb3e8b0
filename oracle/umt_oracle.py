"""CPU oracle: fp32 restatement of UNITE's training hot path (reddyav1/unite).

TEST INFRASTRUCTURE -- not product code.  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this module.

Everything here is written functionally over a flat ``state_dict`` (the
reference's own key names) with plain torch fp32 ops on
the CPU.  Each function cites the reference lines it restates (paths relative
to the reference checkout).  The restatement is pinned against the reference
itself by ``oracle/make_golden.py`` (run in the build container, where
/root/reference exists) -> ``tests/golden/*.npz`` -> ``tests/test_oracle_golden.py``.

Parity status:
  * teacher (src/models/clip.py)            : pinned, reference imported as-is.
  * student / stage-2 ViT (modeling_*.py)   : pinned, reference imported with
    in-memory stand-ins for the 4 ``timm`` symbols it imports
    (to_2tuple, trunc_normal_, register_model, drop_path); none of them takes
    part in the arithmetic the fixtures exercise (drop_path_rate = 0).
  * drop_path (timm 0.4.12, not in /root/reference): restated from its
    published algorithm, "parity unpinned".
  * optimizer: torch.optim.AdamW is the reference's optimizer
    (src/optim_factory.py:162-163); ``adamw_step`` is checked against it.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]


# --------------------------------------------------------------------------
# configs
# --------------------------------------------------------------------------
@dataclass
class StudentCfg:
    """kwargs of AdaptationVisionTransformer (src/models/modeling_adaptation.py:219-247)."""
    img_size: int = 224
    patch_size: int = 16
    embed_dim: int = 768
    depth: int = 12
    num_heads: int = 12
    mlp_ratio: float = 4.0
    num_frames: int = 8
    tubelet_size: int = 1
    clip_decoder_embed_dim: int = 768
    clip_output_dim: int = 512
    clip_return_layers: Sequence[int] = (6, 7, 8, 9, 10, 11)
    clip_norm_type: str = "l2"
    ln_eps: float = 1e-6          # partial(nn.LayerNorm, eps=1e-6), modeling_adaptation.py:348

    @property
    def grid(self) -> int:
        return self.img_size // self.patch_size

    @property
    def num_patches(self) -> int:
        # modeling_finetune.py:161
        return self.grid * self.grid * (self.num_frames // self.tubelet_size)


@dataclass
class TeacherCfg:
    """kwargs of clip.VisionTransformer (src/models/clip.py:107-112)."""
    input_resolution: int = 224
    patch_size: int = 16
    width: int = 768
    layers: int = 12
    heads: int = 12
    output_dim: int = 512
    kernel_size: int = 1
    clip_return_layers: Sequence[int] = (6, 7, 8, 9, 10, 11)
    clip_norm_type: str = "l2"
    ln_eps: float = 1e-5          # nn.LayerNorm default (clip.py:20-26)


@dataclass
class VitCfg:
    """kwargs of the stage-2 VisionTransformer (src/models/modeling_finetune.py:240-265)."""
    img_size: int = 224
    patch_size: int = 16
    embed_dim: int = 768
    depth: int = 12
    num_heads: int = 12
    num_classes: int = 8
    all_frames: int = 16
    tubelet_size: int = 1
    ln_eps: float = 1e-6

    @property
    def num_patches(self) -> int:
        g = self.img_size // self.patch_size
        return g * g * (self.all_frames // self.tubelet_size)


# --------------------------------------------------------------------------
# small pieces
# --------------------------------------------------------------------------
def sinusoid_table(n_position: int, d_hid: int) -> Tensor:
    """modeling_adaptation.py:41-51 -- angle = pos / 10000^(2*(j//2)/d); sin on even j, cos on odd j.
    Computed in float64 (numpy, as the reference) then cast to float32; shape (1, n_position, d_hid)."""
    j = np.arange(d_hid)
    denom = np.power(10000.0, 2.0 * (j // 2) / d_hid)
    table = np.arange(n_position, dtype=np.float64)[:, None] / denom[None, :]
    table[:, 0::2] = np.sin(table[:, 0::2])
    table[:, 1::2] = np.cos(table[:, 1::2])
    return torch.tensor(table, dtype=torch.float32).unsqueeze(0)


def layer_norm(x: Tensor, w: Tensor, b: Tensor, eps: float) -> Tensor:
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)      # biased, as nn.LayerNorm
    return (x - mu) * torch.rsqrt(var + eps) * w + b


def gelu_erf(x: Tensor) -> Tensor:
    """nn.GELU() default (exact erf form), modeling_finetune.py:57,62."""
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def quick_gelu(x: Tensor) -> Tensor:
    """clip.py:29-31."""
    return x * torch.sigmoid(1.702 * x)


def drop_path(x: Tensor, drop_prob: float, training: bool, rand: Optional[Tensor] = None) -> Tensor:
    """timm 0.4.12 ``drop_path`` as called at modeling_finetune.py:50 (parity unpinned: timm is
    not part of /root/reference).  identity if p == 0 or not training, else
    x / keep * floor(keep + U[0,1)), one uniform per sample broadcast over the other dims.
    ``rand`` (B,) supplies the uniforms so that tests are deterministic."""
    if drop_prob == 0.0 or not training:
        return x
    keep = 1.0 - drop_prob
    if rand is None:
        rand = torch.rand(x.shape[0], dtype=x.dtype)
    shape = (x.shape[0],) + (1,) * (x.ndim - 1)
    mask = torch.floor(keep + rand.reshape(shape).to(x.dtype))
    return x / keep * mask


def im2col(videos: Tensor, patch: int, tubelet: int) -> Tensor:
    """Rows of the patch-embedding GEMM.  Conv3d with kernel == stride == (tubelet, p, p)
    (modeling_finetune.py:165-167, clip.py:123-128) is a GEMM over rows ordered (c, dt, ph, pw);
    token order is t*gh*gw + h*gw + w (flatten(2).transpose(1,2), modeling_finetune.py:174).
    (B,C,T,H,W) -> (B, T'*gh*gw, C*tubelet*p*p)."""
    B, C, T, H, W = videos.shape
    gt, gh, gw = T // tubelet, H // patch, W // patch
    x = videos.reshape(B, C, gt, tubelet, gh, patch, gw, patch)
    x = x.permute(0, 2, 4, 6, 1, 3, 5, 7)           # B, gt, gh, gw, C, dt, ph, pw
    return x.reshape(B, gt * gh * gw, C * tubelet * patch * patch)


def mha(x: Tensor, w_qkv: Tensor, b_qkv: Optional[Tensor], w_o: Tensor, b_o: Tensor, heads: int,
        return_probs: bool = False):
    """softmax(q k^T / sqrt(hd)) v then output projection, batch-first x (B, N, D).
    Student: modeling_finetune.py:100-119 (q scaled by head_dim**-0.5 before the product).
    Teacher: nn.MultiheadAttention with packed in_proj (clip.py:38,48-53) -- same algebra."""
    B, N, D = x.shape
    hd = D // heads
    qkv = F.linear(x, w_qkv, b_qkv).reshape(B, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    attn = (q * hd ** -0.5) @ k.transpose(-2, -1)
    attn = attn.softmax(dim=-1)
    out = (attn @ v).transpose(1, 2).reshape(B, N, D)
    out = F.linear(out, w_o, b_o)
    if return_probs:
        return out, attn
    return out


# --------------------------------------------------------------------------
# student  (src/models/modeling_adaptation.py + modeling_finetune.py)
# --------------------------------------------------------------------------
def vit_block(x: Tensor, sd: SD, p: str, heads: int, eps: float,
              dp_rate: float = 0.0, training: bool = False,
              dp_rand: Optional[Tuple[Tensor, Tensor]] = None) -> Tensor:
    """Block.forward, gamma_1 is None branch (modeling_finetune.py:143-146)."""
    h = layer_norm(x, sd[p + "norm1.weight"], sd[p + "norm1.bias"], eps)
    # qkv bias = cat(q_bias, zeros, v_bias)  (modeling_finetune.py:102-106)
    qb, vb = sd[p + "attn.q_bias"], sd[p + "attn.v_bias"]
    b_qkv = torch.cat((qb, torch.zeros_like(vb), vb))
    a = mha(h, sd[p + "attn.qkv.weight"], b_qkv, sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"], heads)
    x = x + drop_path(a, dp_rate, training, None if dp_rand is None else dp_rand[0])
    h = layer_norm(x, sd[p + "norm2.weight"], sd[p + "norm2.bias"], eps)
    h = F.linear(h, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"])
    h = gelu_erf(h)
    h = F.linear(h, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
    x = x + drop_path(h, dp_rate, training, None if dp_rand is None else dp_rand[1])
    return x


def student_forward(sd: SD, videos: Tensor, mask: Tensor, cfg: StudentCfg, clip_only: bool = True,
                    drop_path_rate: float = 0.0, training: bool = False,
                    dp_rand: Optional[Tensor] = None):
    """AdaptationVisionTransformer.forward (modeling_adaptation.py:304-334) without cls token.

    videos (B,3,T,H,W) f32; mask bool (B, T*gh*gw), True = masked (dropped).
    dp_rand: optional (depth, 2, B) uniforms for stochastic depth.
    Returns x_clip (K,B,n_vis,C_clip) if clip_only else (x_vis (B,n_vis,D), x_clip)."""
    B = videos.shape[0]
    D = cfg.embed_dim
    # encoder.forward_features (modeling_adaptation.py:131-169)
    cols = im2col(videos, cfg.patch_size, cfg.tubelet_size)
    w = sd["encoder.patch_embed.proj.weight"].reshape(D, -1)
    x = cols @ w.t() + sd["encoder.patch_embed.proj.bias"]
    x = x + sinusoid_table(cfg.num_patches, D)                       # :144
    x_vis = x[~mask].reshape(B, -1, D)                                # :153 ascending token order
    rates = [r.item() for r in torch.linspace(0, drop_path_rate, cfg.depth)]   # :93
    taps: List[Tensor] = []
    ret = list(cfg.clip_return_layers)
    for i in range(cfg.depth):
        rnd = None if dp_rand is None else (dp_rand[i, 0], dp_rand[i, 1])
        x_vis = vit_block(x_vis, sd, f"encoder.blocks.{i}.", cfg.num_heads, cfg.ln_eps, rates[i], training, rnd)
        if i in ret:
            taps.append(x_vis)                                        # :163-164
        if i == max(ret) and clip_only:
            break                                                     # :165-166
    x_clip_vis = layer_norm(torch.stack(taps), sd["encoder.norm.weight"], sd["encoder.norm.bias"], cfg.ln_eps)  # :168
    # encoder.head is Identity (encoder_num_classes = 0, :101,173)
    # decoders (:316-325)
    K = x_clip_vis.shape[0]
    pos = sinusoid_table(cfg.num_patches, cfg.clip_decoder_embed_dim).repeat(B, 1, 1)
    pos_vis = pos[~mask].view(B, -1, cfg.clip_decoder_embed_dim).unsqueeze(0)
    x_full = x_clip_vis + pos_vis
    outs = []
    for k in range(K):
        y = F.linear(x_full[k], sd[f"clip_decoder.{k}.head.weight"], sd[f"clip_decoder.{k}.head.bias"])
        y = layer_norm(y, sd[f"clip_decoder.{k}.norm.weight"], sd[f"clip_decoder.{k}.norm.bias"], cfg.ln_eps)  # :204
        if cfg.clip_norm_type == "l2":
            y = y / y.norm(dim=-1, keepdim=True)                      # :206-207
        outs.append(y)
    x_clip = torch.stack(outs)
    if clip_only:
        return x_clip
    x_out = layer_norm(x_vis, sd["encoder.norm.weight"], sd["encoder.norm.bias"], cfg.ln_eps)   # :178 (head = Identity)
    return x_out, x_clip


# --------------------------------------------------------------------------
# teacher  (src/models/clip.py)
# --------------------------------------------------------------------------
def teacher_forward(sd: SD, videos: Tensor, cfg: TeacherCfg, return_attn: bool = True):
    """clip.VisionTransformer.forward, mask=None path (clip.py:145-188).

    Returns feats (K, B, T*HW, output_dim) L2-normalised and, if return_attn,
    attn (B*T, HW): head-averaged last-layer softmax probabilities, CLS row, patch columns."""
    B, C, T, H, W = videos.shape
    Wd = cfg.width
    cols = im2col(videos, cfg.patch_size, cfg.kernel_size)               # conv1, no bias (:123-128,146)
    x = cols @ sd["conv1.weight"].reshape(Wd, -1).t()                    # (B, T*HW, Wd), t-major
    HW = (H // cfg.patch_size) * (W // cfg.patch_size)
    x = x.reshape(B * T, HW, Wd)                                         # :148
    cls = sd["class_embedding"].reshape(1, 1, Wd).expand(B * T, 1, Wd)
    x = torch.cat([cls, x], dim=1) + sd["positional_embedding"]          # :150-151
    x = layer_norm(x, sd["ln_pre.weight"], sd["ln_pre.bias"], cfg.ln_eps)
    taps = []
    attn_last = None
    for i in range(cfg.layers):
        p = f"transformer.resblocks.{i}."
        h = layer_norm(x, sd[p + "ln_1.weight"], sd[p + "ln_1.bias"], cfg.ln_eps)
        last = (i == cfg.layers - 1) and return_attn
        a = mha(h, sd[p + "attn.in_proj_weight"], sd[p + "attn.in_proj_bias"],
                sd[p + "attn.out_proj.weight"], sd[p + "attn.out_proj.bias"], cfg.heads, return_probs=last)
        if last:
            a, probs = a
            attn_last = probs.mean(dim=1)                                # need_weights=True averages heads (:51,95-96)
        x = x + a
        h = layer_norm(x, sd[p + "ln_2.weight"], sd[p + "ln_2.bias"], cfg.ln_eps)
        h = quick_gelu(F.linear(h, sd[p + "mlp.c_fc.weight"], sd[p + "mlp.c_fc.bias"]))
        x = x + F.linear(h, sd[p + "mlp.c_proj.weight"], sd[p + "mlp.c_proj.bias"])
        if i in cfg.clip_return_layers:
            taps.append(x)                                               # :99-100
    z = torch.stack(taps)                                                # (K, BT, 1+HW, Wd)
    K = z.shape[0]
    z = layer_norm(z[:, :, 1:, :], sd["ln_post.weight"], sd["ln_post.bias"], cfg.ln_eps)   # :168
    z = z.reshape(K, B, T * HW, Wd)                                      # :169
    z = z @ sd["proj"]                                                   # :170
    if cfg.clip_norm_type == "l2":
        z = z / z.norm(dim=-1, keepdim=True)                             # :172-173
    if return_attn:
        return z, attn_last[:, 0, 1:]                                    # :183
    return z


# --------------------------------------------------------------------------
# stage-1 engine pieces  (run_stage1.py:379-435)
# --------------------------------------------------------------------------
def mask_from_importance(importance: Tensor, n_vis: int, B: int) -> Tensor:
    """run_stage1.py:383-387: importance (BT, N) is a per-frame permutation (torch.multinomial without
    replacement); its first n_vis entries are visible.  Returns bool (B, T*N), True = masked."""
    BT, N = importance.shape
    m = torch.ones((BT, N))
    rows = torch.arange(BT).view(-1, 1).repeat(1, n_vis)
    m[rows, importance[:, :n_vis]] = 0
    return m.view(B, -1).to(torch.bool)


def n_visible(N: int, mask_ratio: float) -> int:
    return N - int(N * mask_ratio)                                        # run_stage1.py:380


def gather_targets(norm_clip: Tensor, mask: Tensor) -> Tensor:
    """run_stage1.py:389-397: (K,B,THW,C)[~mask] -> (K,B,n_vis,C), ascending token order."""
    K, B, _, C = norm_clip.shape
    m = mask.unsqueeze(0).repeat(K, 1, 1)
    return norm_clip[~m].reshape(K, B, -1, C)


def umt_loss(outputs_clip: Tensor, targets_clip: Tensor, clip_loss_type: str = "l2") -> Tensor:
    """run_stage1.py:403-408,431-436: 'l2' (every shipped config) or nn.MSELoss / nn.SmoothL1Loss / nn.L1Loss on the same tensors."""
    if clip_loss_type == "l2":
        return (2 - 2 * (outputs_clip * targets_clip).sum(dim=-1)).mean()
    fn = {"mse": F.mse_loss, "smooth_l1": F.smooth_l1_loss, "l1": F.l1_loss}[clip_loss_type]
    return fn(input=outputs_clip, target=targets_clip)


def teacher_resize(videos: Tensor, resolution: int) -> Tensor:
    """run_stage1.py:362-370: per-plane bicubic resize of the teacher's input when its resolution differs."""
    B, C, T, H, W = videos.shape
    if H == resolution:
        return videos
    out = F.interpolate(videos.reshape(B, C * T, H, W), size=(resolution, resolution), mode="bicubic", align_corners=False)
    return out.view(B, C, T, resolution, resolution)


def stage1_loss(student_sd: SD, teacher_sd: SD, videos: Tensor, mask: Tensor,
                scfg: StudentCfg, tcfg: TeacherCfg, clip_loss_type: str = "l2"):
    """teacher -> gather -> student -> loss for an explicit mask (clip_loss_data='mixed')."""
    with torch.no_grad():
        feats, attn = teacher_forward(teacher_sd, teacher_resize(videos, tcfg.input_resolution), tcfg, return_attn=True)
        tgt = gather_targets(feats, mask)
    out = student_forward(student_sd, videos, mask, scfg, clip_only=True)
    return umt_loss(out, tgt, clip_loss_type), out, tgt, attn


# --------------------------------------------------------------------------
# stage-3 / shared utilities (src/utils.py)
# --------------------------------------------------------------------------
def get_greedy_masks(attn: Tensor, mask_ratio: float, k: int) -> Tensor:
    """utils.py:89-120: member i unmasks attention ranks i, i+k, ... (first N_unmask of them).
    Returns bool (k, BT, N), True = masked."""
    BT, N = attn.shape
    n_unmask = N - int(N * mask_ratio)
    order = attn.sort(dim=1, descending=True)[1]
    masks = torch.ones((k, BT, N), dtype=torch.bool)
    for i in range(k):
        idx = order[:, i::k][:, :n_unmask]
        masks[i].scatter_(1, idx, False)
    return masks


def cosine_scheduler(base_value, final_value, epochs, niter_per_ep, warmup_epochs=0,
                     start_warmup_value=0, warmup_steps=-1) -> np.ndarray:
    """utils.py:646-663."""
    warmup_schedule = np.array([])
    warmup_iters = warmup_epochs * niter_per_ep
    if warmup_steps > 0:
        warmup_iters = warmup_steps
    if warmup_epochs > 0:
        warmup_schedule = np.linspace(start_warmup_value, base_value, warmup_iters)
    iters = np.arange(epochs * niter_per_ep - warmup_iters)
    schedule = np.array([final_value + 0.5 * (base_value - final_value) * (1 + math.cos(math.pi * i / len(iters)))
                         for i in iters])
    schedule = np.concatenate((warmup_schedule, schedule))
    assert len(schedule) == epochs * niter_per_ep
    return schedule


def get_num_layer_for_vit(var_name: str, num_max_layer: int) -> int:
    """optim_factory.py:44-62."""
    if var_name in ("cls_token", "mask_token", "pos_embed"):
        return 0
    if var_name.startswith("patch_embed"):
        return 0
    if var_name.startswith("rel_pos_bias"):
        return num_max_layer - 1
    if var_name.startswith("blocks"):
        return int(var_name.split(".")[1]) + 1
    if var_name.startswith("transformer.resblocks"):
        return int(var_name.split(".")[2]) + 1
    if var_name in ("class_embedding", "positional_embedding", "temporal_positional_embedding"):
        return 0
    if var_name.startswith("conv1"):
        return 0
    return num_max_layer - 1


def parameter_group_names(named_shapes: Sequence[Tuple[str, Tuple[int, ...]]], weight_decay: float,
                          skip_list=(), layer_scales: Optional[Sequence[float]] = None):
    """optim_factory.py:76-118 -- returns {group_name: dict(weight_decay, lr_scale, params=[names])}
    in first-seen order."""
    groups: Dict[str, dict] = {}
    for name, shape in named_shapes:
        if len(shape) == 1 or name.endswith(".bias") or name in skip_list:
            g, wd = "no_decay", 0.0
        else:
            g, wd = "decay", weight_decay
        scale = 1.0
        if layer_scales is not None:
            lid = get_num_layer_for_vit(name, len(layer_scales))
            g = "layer_%d_%s" % (lid, g)
            scale = layer_scales[lid]
        if g not in groups:
            groups[g] = {"weight_decay": wd, "lr_scale": scale, "params": []}
        groups[g]["params"].append(name)
    return groups


def adamw_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float,
               beta1: float, beta2: float, eps: float, wd: float) -> None:
    """One torch.optim.AdamW update (single-tensor form), in place.  step is 1-based."""
    p.mul_(1 - lr * wd)
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)


def grad_norm(grads: Sequence[Tensor]) -> Tensor:
    """utils.get_grad_norm_ with norm_type 2 (utils.py:631-643)."""
    return torch.norm(torch.stack([torch.norm(g, 2.0) for g in grads]), 2.0)


# --------------------------------------------------------------------------
# stage 2  (modeling_finetune.VisionTransformer, use_mean_pooling=True)
# --------------------------------------------------------------------------
def vit_classifier_forward(sd: SD, videos: Tensor, cfg: VitCfg) -> Tensor:
    """modeling_finetune.py:356-383: all tokens, fc_norm(mean over tokens), head."""
    D = cfg.embed_dim
    cols = im2col(videos, cfg.patch_size, cfg.tubelet_size)
    x = cols @ sd["patch_embed.proj.weight"].reshape(D, -1).t() + sd["patch_embed.proj.bias"]
    x = x + sinusoid_table(cfg.num_patches, D)
    for i in range(cfg.depth):
        x = vit_block(x, sd, f"blocks.{i}.", cfg.num_heads, cfg.ln_eps)
    x = layer_norm(x.mean(1), sd["fc_norm.weight"], sd["fc_norm.bias"], cfg.ln_eps)
    return F.linear(x, sd["head.weight"], sd["head.bias"])


# --------------------------------------------------------------------------
# stage 3 selection math (run_stage3.py:486-625), default 'clip_matchORconf'
# --------------------------------------------------------------------------
def stage3_select(logits_full_t: Tensor, clip_probs_t: Tensor, clip_threshold: float):
    """run_stage3.py:488-490,556-576, selection_strategy == 'clip_matchORconf'.
    match = CLIP pred == student pred; conf = exactly one of (student msp >= thr, CLIP msp >= thr)
    and not match; selected = match | conf; the pseudo-label is the student prediction
    (the torch.where at :575 is overwritten at :576)."""
    probs = logits_full_t.softmax(dim=-1)
    msp_t, preds_t = probs.max(dim=-1)
    clip_msp, clip_preds = clip_probs_t.max(dim=-1)
    match = clip_preds == preds_t
    student_conf = msp_t >= clip_threshold
    clip_conf = clip_msp >= clip_threshold
    conf = torch.logical_xor(student_conf, clip_conf) & torch.logical_not(match)
    sel = torch.logical_or(conf, match)
    return sel, preds_t, msp_t


def stage3_committee_select(logits_full_t: Tensor, logits_masked_t: Tensor, threshold: float = 0.5):
    """run_stage3.py:508-531: consistency (all k committee members agree with the full-video
    prediction) and confidence (msp >= 0.5, global_threshold is overwritten at :522) masks."""
    probs = logits_full_t.softmax(dim=-1)
    msp_t, preds_t = probs.max(dim=-1)
    k = logits_masked_t.shape[0]
    votes = torch.zeros_like(preds_t)
    for i in range(k):
        votes += (logits_masked_t[i].argmax(dim=-1) == preds_t).long()
    return votes >= k, msp_t >= threshold


def stage3_target_loss(logits_masked_last: Tensor, sel: Tensor, labels: Tensor, msp_t: Tensor,
                       tgt_ratio: float) -> Tensor:
    """run_stage3.py:599-616: ratio * sel_ratio * mean(msp * CE(masked logits[-1][sel], pl[sel]))
    (train_masked and conf_weighted_loss both on); zeros(1) when nothing is selected."""
    if sel.sum() == 0:
        return torch.zeros(1)
    ce = F.cross_entropy(logits_masked_last[sel], labels[sel], reduction="none")
    sel_ratio = sel.float().mean()
    return tgt_ratio * sel_ratio * (msp_t[sel] * ce).mean()


def stage3_pseudo_labels(logits_full_t: Tensor, logits_masked_t: Tensor, strategy: str, clip_probs_t: Optional[Tensor] = None,
                         labels_t: Optional[Tensor] = None, clip_threshold: float = 0.5):
    """run_stage3.py:488-593, every selection_strategy.  Returns (sel bool (B,), pseudo-label (B,), msp (B,))."""
    probs = logits_full_t.softmax(dim=-1)
    msp_t, preds_t = probs.max(dim=-1)
    cons, conf = stage3_committee_select(logits_full_t, logits_masked_t, 0.5)
    if strategy == "conf":
        sel = conf
    elif strategy == "cons":
        sel = cons
    elif strategy == "consORconf":
        sel = cons | conf                                                 # :533-534
    elif strategy == "consANDconf":
        sel = cons & conf                                                 # :535-536
    elif strategy == "clip_only":
        clip_msp, _ = clip_probs_t.max(dim=-1)                            # :551-554: CLIP confident at global_threshold (0.5, :522);
        sel = clip_msp >= 0.5                                             # the label stays the student's prediction (:603)
    elif strategy == "clip_matchORconf":
        sel, _, _ = stage3_select(logits_full_t, clip_probs_t, clip_threshold)
    elif strategy == "oracle":
        sel = preds_t == labels_t                                         # :585-587
    else:
        raise ValueError(strategy)
    return sel, preds_t, msp_t


def stage3_loss(student_sd: SD, teacher_sd: SD, cls_w: Tensor, cls_b: Tensor, videos_s: Tensor, labels_s: Tensor,
                videos_t: Tensor, videos_t_aug: Tensor, labels_t: Tensor, scfg: StudentCfg, tcfg: TeacherCfg, mask_ratio: float,
                strategy: str, clip_probs_t: Optional[Tensor] = None, clip_threshold: float = 0.5, src_ratio_pl: float = 1.0,
                tgt_ratio: float = 1.0, conf_weighted: bool = True, attn: Optional[Tensor] = None):
    """One stage-3 loss evaluation, run_stage3.py:434-625 (train_masked, k = 2 committee, mean-pooled tokens, linear src_classifier).
    A composition of pieces that are each pinned on the reference's vectors (student x_vis, teacher attention, greedy masks,
    selection); the composition itself follows the cited lines.  Returns (loss, loss_s, loss_t, sel)."""
    B_s, B_t = videos_s.shape[0], videos_t.shape[0]
    if attn is None:
        _, attn = teacher_forward(teacher_sd, videos_t_aug, tcfg, return_attn=True)            # :434-451, (B_t*T, N)
    N_all = scfg.num_patches
    full = torch.zeros(1, N_all, dtype=torch.bool)

    def classify(x_vis):
        return F.linear(x_vis.mean(dim=1), cls_w, cls_b)                                       # :333-338, :477

    x_s, _ = student_forward(student_sd, videos_s, full.expand(B_s, -1), scfg, clip_only=False)       # :475
    logits_s = classify(x_s)
    with torch.no_grad():
        x_t, _ = student_forward(student_sd, videos_t, full.expand(B_t, -1), scfg, clip_only=False)   # :480-483
        logits_full_t = classify(x_t)
    loss_s = F.cross_entropy(logits_s, labels_s)                                               # :486
    k = 2
    masks = get_greedy_masks(attn, mask_ratio, k)                                              # :497, (k, B_t*T, N)
    logits_masked = []
    for i in range(k):
        m = masks[i].reshape(B_t, -1)                                                          # 'k (B T) N -> (k B) (T N)'
        x_m, _ = student_forward(student_sd, videos_t_aug, m, scfg, clip_only=False)           # :503
        logits_masked.append(classify(x_m))
    logits_masked = torch.stack(logits_masked)
    sel, pseudo, msp = stage3_pseudo_labels(logits_full_t, logits_masked.detach(), strategy, clip_probs_t, labels_t, clip_threshold)
    if sel.sum() > 0:
        w = msp if conf_weighted else torch.ones_like(msp)
        ce = F.cross_entropy(logits_masked[-1][sel], pseudo[sel], reduction="none")
        loss_t = tgt_ratio * sel.float().mean() * (w[sel] * ce).mean()                         # :599-613
    else:
        loss_t = torch.zeros(())
    return src_ratio_pl * loss_s + loss_t, loss_s, loss_t, sel


def clip_to_tensor(frames_u8: Tensor, mean: Sequence[float], std: Sequence[float], flip: Optional[Tensor] = None) -> Tensor:
    """Input path after decode/crop/resize: GroupRandomHorizontalFlip (transforms.py:68-79), Stack (:209-223), ToTorchFormatTensor
    (:226-245: HWC -> CHW, .float().div(255.)), GroupNormalize (:82-96: t.sub_(m).div_(s) per channel), then mae.py:218-219
    view (T,3,H,W) -> transpose -> (3,T,H,W).  frames_u8: (B,T,H,W,3) uint8; flip: bool/uint8 (B,) -> (B,3,T,H,W) f32."""
    out = []
    for b in range(frames_u8.shape[0]):
        fr = frames_u8[b]
        if flip is not None and bool(flip[b]):
            fr = fr.flip(2)                                            # FLIP_LEFT_RIGHT of every frame
        stacked = torch.cat([f for f in fr], dim=2)                    # H x W x 3T
        img = stacked.permute(2, 0, 1).contiguous().float().div(255.)  # 3T x H x W
        rep_mean, rep_std = list(mean) * (img.shape[0] // 3), list(std) * (img.shape[0] // 3)
        for t, m, s in zip(img, rep_mean, rep_std):
            t.sub_(m).div_(s)
        out.append(img.view((fr.shape[0], 3) + img.shape[-2:]).transpose(0, 1))
    return torch.stack(out)


def clip_encode_image(sd: SD, frames: Tensor, cfg: TeacherCfg) -> Tensor:
    """OpenAI CLIP ``VisionTransformer.forward`` (package `clip`, pinned as git+https://github.com/openai/CLIP.git in the reference's
    environment.yaml:353; not vendored): conv1 -> [cls; patches] + pos -> ln_pre -> blocks -> ln_post(x[:, 0]) @ proj.  Same blocks as
    teacher_forward (the reference's clip.py is derived from it).  frames (N,3,H,W) -> (N, output_dim), not normalised."""
    W = cfg.width
    cols = im2col(frames.unsqueeze(2), cfg.patch_size, 1)                       # (N, HW, 3*P*P)
    x = cols @ sd["conv1.weight"].reshape(W, -1).t()
    x = torch.cat([sd["class_embedding"].expand(x.shape[0], 1, W), x], dim=1) + sd["positional_embedding"]
    x = layer_norm(x, sd["ln_pre.weight"], sd["ln_pre.bias"], cfg.ln_eps)
    for i in range(cfg.layers):
        p = f"transformer.resblocks.{i}."
        h = layer_norm(x, sd[p + "ln_1.weight"], sd[p + "ln_1.bias"], cfg.ln_eps)
        x = x + mha(h, sd[p + "attn.in_proj_weight"], sd[p + "attn.in_proj_bias"], sd[p + "attn.out_proj.weight"],
                    sd[p + "attn.out_proj.bias"], cfg.heads)
        h = layer_norm(x, sd[p + "ln_2.weight"], sd[p + "ln_2.bias"], cfg.ln_eps)
        x = x + F.linear(quick_gelu(F.linear(h, sd[p + "mlp.c_fc.weight"], sd[p + "mlp.c_fc.bias"])),
                         sd[p + "mlp.c_proj.weight"], sd[p + "mlp.c_proj.bias"])
    return layer_norm(x[:, 0], sd["ln_post.weight"], sd["ln_post.bias"], cfg.ln_eps) @ sd["proj"]


def clip_infer(sd: SD, videos: Tensor, text_features: Tensor, cfg: TeacherCfg) -> Tensor:
    """src/utils.py:55-68: per-frame similarities (x100, soft-max over classes) averaged over the frames of each clip."""
    B, C, T, H, Wd = videos.shape
    images = videos.permute(0, 2, 1, 3, 4).reshape(B * T, C, H, Wd)
    return clip_similarity(clip_encode_image(sd, images, cfg), text_features, B)


def clip_similarity(image_features: Tensor, text_features: Tensor, B: int) -> Tensor:
    """src/utils.py:61-68, the arithmetic behind encode_image: normalise both sides, softmax(100 cos) per frame, mean over a clip's frames.
    image_features (B*T, C), text_features (n_cls, C) -> (B, n_cls).  Pinned on the reference's own clip_infer (tests/golden/stage3_step.npz)."""
    f = image_features / image_features.norm(dim=-1, keepdim=True)
    t = text_features / text_features.norm(dim=-1, keepdim=True)
    sim = (100 * f @ t.t()).softmax(dim=-1)
    return sim.view(B, -1, sim.shape[-1]).mean(dim=1)
