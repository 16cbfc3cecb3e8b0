#!/usr/bin/env python
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE (reddyav1/unite) on CPU.

TEST INFRASTRUCTURE.  Runs only in the build container (needs /root/reference); the
reference's Python never travels -- only the inputs/outputs written here are committed.

How the reference is imported (nothing is written to /root/reference, nothing is copied):
  * src/models/clip.py                    -- loaded by file path, needs no stand-ins.
  * src/models/modeling_finetune.py,
    src/models/modeling_adaptation.py     -- loaded by file path under in-memory stand-ins for the
    ``timm`` symbols they import: to_2tuple, trunc_normal_ (-> torch.nn.init.trunc_normal_),
    register_model (identity decorator), drop_path (timm 0.4.12 restated; NOT exercised: all
    fixtures use drop_path_rate = 0 / eval).  timm is not installed in this image.
  * src/utils.py, src/optim_factory.py    -- additionally stand-ins for timm.utils.get_state_dict,
    torch._six.inf, the OpenAI ``clip`` package (unused by the functions called),
    tensorboardX.SummaryWriter and the ten timm.optim.* classes (only torch.optim.AdamW is used).
The run_stage*.py scripts are not importable here (wandb, decord, src/knn.py missing);
the stage-1 step below restates run_stage1.py:379-456 around the reference's own model classes,
torch.multinomial replaced by an explicit permutation stored in the fixture.

Usage:  python oracle/make_golden.py   (writes tests/golden/*.npz)
"""
from __future__ import annotations

import importlib.util
import math
import os
import sys
import types
from functools import partial

import numpy as np
import torch
import torch.nn as nn

REF = os.environ.get("UNITE_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
sys.path.insert(0, os.path.dirname(HERE))

from oracle.filler import fill_state_dict, make_videos, make_importance  # noqa: E402


# ---------------------------------------------------------------- stand-ins
def _install_standins():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    def drop_path(x, drop_prob: float = 0.0, training: bool = False):
        if drop_prob == 0.0 or not training:
            return x
        keep = 1 - drop_prob
        shape = (x.shape[0],) + (1,) * (x.ndim - 1)
        r = keep + torch.rand(shape, dtype=x.dtype, device=x.device)
        r.floor_()
        return x.div(keep) * r

    def to_2tuple(x):
        return tuple(x) if isinstance(x, (tuple, list)) else (x, x)

    mod("timm")
    mod("timm.models")
    mod("timm.models.layers", drop_path=drop_path, to_2tuple=to_2tuple,
        trunc_normal_=torch.nn.init.trunc_normal_)
    mod("timm.models.registry", register_model=lambda f: f)
    mod("timm.utils", get_state_dict=lambda m, *a, **k: m.state_dict())
    mod("torch._six", inf=math.inf)
    mod("clip")
    mod("tensorboardX", SummaryWriter=object)
    mod("timm.optim")
    for sub, cls in [("adafactor", "Adafactor"), ("adahessian", "Adahessian"), ("adamp", "AdamP"),
                     ("lookahead", "Lookahead"), ("nadam", "Nadam"), ("novograd", "NovoGrad"),
                     ("nvnovograd", "NvNovoGrad"), ("radam", "RAdam"), ("rmsprop_tf", "RMSpropTF"),
                     ("sgdp", "SGDP")]:
        mod("timm.optim." + sub, **{cls: type(cls, (), {})})
    # package shells so that relative imports inside src/models resolve without running
    # src/models/__init__.py (which pulls the unused pretrain models)
    pkg = mod("src")
    pkg.__path__ = [os.path.join(REF, "src")]
    pkgm = mod("src.models")
    pkgm.__path__ = [os.path.join(REF, "src", "models")]


def _load(name, relpath):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, relpath))
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def _np(d):
    return {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in d.items()}


def _shapes(model):
    return [(k, tuple(v.shape)) for k, v in model.state_dict().items()]


def _ref_stage1_step(student, teacher, videos, importance, mask_ratio):
    """run_stage1.py:360-435 with explicit ``importance`` (stands for torch.multinomial(attn, N), :382)."""
    B = videos.shape[0]
    with torch.no_grad():
        norm_clip, attn = teacher(videos)
        BT, N = attn.shape
        N_vis = N - int(N * mask_ratio)
        bool_masked_pos = torch.ones((BT, N))
        pos1 = torch.arange(BT).view(-1, 1).repeat(1, N_vis)
        pos2 = importance[:, :N_vis]
        bool_masked_pos[pos1, pos2] = 0
        bool_masked_pos = bool_masked_pos.view(B, -1).to(torch.bool)
        C_CLIP = norm_clip.shape[-1]
        K = norm_clip.shape[0]
        clip_bool_masked_pos = bool_masked_pos.unsqueeze(0).repeat(K, 1, 1)
        targets_clip = norm_clip[~clip_bool_masked_pos].reshape(K, B, -1, C_CLIP)
    outputs_clip = student(videos, bool_masked_pos, clip_only=True)
    loss = (2 - 2 * (outputs_clip * targets_clip).sum(dim=-1)).mean()
    return loss, outputs_clip, targets_clip, attn, norm_clip, bool_masked_pos


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    os.makedirs(OUT, exist_ok=True)
    _install_standins()
    clip_ref = _load("src.models.clip", "src/models/clip.py")
    ft_ref = _load("src.models.modeling_finetune", "src/models/modeling_finetune.py")
    ad_ref = _load("src.models.modeling_adaptation", "src/models/modeling_adaptation.py")
    utils_ref = _load("src.utils", "src/utils.py")
    optim_ref = _load("src.optim_factory", "src/optim_factory.py")

    # ------------------------------------------------------------ 1. teacher, tiny
    # head_dim 64 (width 128 / 2 heads) so that the same fixture also runs through the gfx950 attention kernels
    tkw = dict(input_resolution=32, patch_size=16, width=128, layers=3, heads=2, output_dim=64,
               return_attn=True, clip_return_layers=[1, 2])
    teacher = clip_ref.VisionTransformer(**tkw).eval()
    tsd = fill_state_dict(_shapes(teacher), seed=1)
    teacher.load_state_dict(tsd)
    vid = make_videos(2, 2, 32, 32, seed=2)
    with torch.no_grad():
        feats, attn = teacher(vid)
    # weights are NOT stored: fill_state_dict(shapes, seed) regenerates them bit-exactly on both sides
    np.savez_compressed(os.path.join(OUT, "teacher_tiny.npz"),
                        **_np({"in.videos": vid, "in.seed_weights": 1, "out.feats": feats, "out.attn": attn}))
    print("teacher_tiny", feats.shape, attn.shape)

    # ------------------------------------------------------------ 2. student, tiny (fwd + bwd)
    skw = dict(img_size=32, patch_size=16, encoder_embed_dim=128, encoder_depth=3, encoder_num_heads=2,
               encoder_num_classes=0, mlp_ratio=4, qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6),
               num_frames=2, tubelet_size=1, clip_decoder_embed_dim=128, clip_output_dim=64,
               clip_return_layers=[1, 2])
    student = ad_ref.AdaptationVisionTransformer(**skw).train()
    ssd = fill_state_dict(_shapes(student), seed=3)
    student.load_state_dict(ssd)
    importance = make_importance(2 * 2, 4, seed=4)
    loss, out_clip, tgt_clip, attn2, norm_clip, mask = _ref_stage1_step(student, teacher, vid, importance, 0.5)
    loss.backward()
    grads = {"g." + k: p.grad for k, p in student.named_parameters()}
    with torch.no_grad():
        x_vis, x_clip2 = student(vid, mask, clip_only=False)
    # parameter groups + 3 AdamW steps through the reference's own factory (optim_factory.py:121-163)
    args = types.SimpleNamespace(opt="adamw", weight_decay=0.05, lr=1e-3, opt_eps=1e-8, opt_betas=[0.9, 0.95], momentum=0.9)
    groups = optim_ref.get_parameter_groups(student, 0.05, student.no_weight_decay())
    names = dict((id(p), n) for n, p in student.named_parameters())
    opt = optim_ref.create_optimizer(args, student, skip_list=student.no_weight_decay())
    losses = [loss.item()]
    gnorms = [utils_ref.get_grad_norm_(student.parameters()).item()]
    opt.step()
    for it in range(2):
        opt.zero_grad()
        l2, *_ = _ref_stage1_step(student, teacher, vid, importance, 0.5)
        l2.backward()
        losses.append(l2.item())
        gnorms.append(utils_ref.get_grad_norm_(student.parameters()).item())
        opt.step()
    after = {"after3." + k: v for k, v in student.state_dict().items()}
    AFTER_KEYS = ["encoder.patch_embed.proj.bias", "encoder.blocks.0.attn.q_bias", "encoder.blocks.0.attn.qkv.weight",
                  "encoder.blocks.1.mlp.fc2.weight", "encoder.blocks.2.norm1.weight", "encoder.norm.bias",
                  "clip_decoder.0.head.weight", "clip_decoder.1.norm.weight"]
    np.savez_compressed(
        os.path.join(OUT, "student_tiny.npz"),
        **_np({"in.videos": vid, "in.importance": importance, "in.mask": mask, "in.mask_ratio": 0.5,
               "out.loss": loss, "out.x_clip": out_clip, "out.targets": tgt_clip, "out.x_vis": x_vis,
               "out.losses3": np.array(losses), "out.gnorms3": np.array(gnorms),
               "opt.lr": 1e-3, "opt.wd": 0.05, "opt.betas": np.array([0.9, 0.95]), "opt.eps": 1e-8,
               "groups.decay": np.array([names[id(p)] for p in groups[0]["params"]] if groups[0]["weight_decay"] > 0
                                        else [names[id(p)] for p in groups[1]["params"]]),
               "groups.no_decay": np.array([names[id(p)] for p in groups[1]["params"]] if groups[0]["weight_decay"] > 0
                                           else [names[id(p)] for p in groups[0]["params"]])}),
        **_np({"in.seed_weights": 3}), **_np(grads),
        **_np({k: v for k, v in after.items() if k.split(".", 1)[1] in AFTER_KEYS}))
    print("student_tiny loss", losses, "gnorm", gnorms)

    # ------------------------------------------------------------ 3. stage-2 ViT, tiny (fwd + CE + bwd)
    vkw = dict(img_size=32, patch_size=16, embed_dim=128, depth=2, num_heads=2, mlp_ratio=4, qkv_bias=True,
               norm_layer=partial(nn.LayerNorm, eps=1e-6), num_classes=5, all_frames=4, tubelet_size=1,
               use_mean_pooling=True, init_scale=0.001)
    vit = ft_ref.VisionTransformer(**vkw).train()
    vsd = fill_state_dict(_shapes(vit), seed=5)
    vit.load_state_dict(vsd)
    vid4 = make_videos(3, 4, 32, 32, seed=6)
    labels = torch.tensor([1, 4, 0])
    logits = vit(vid4)
    ce = nn.CrossEntropyLoss()(logits, labels)
    ce.backward()
    np.savez_compressed(os.path.join(OUT, "vit_stage2_tiny.npz"),
                        **_np({"in.videos": vid4, "in.labels": labels, "in.seed_weights": 5, "out.logits": logits, "out.loss": ce}),
                        **_np({"g." + k: p.grad for k, p in vit.named_parameters()}))
    # layer-decay grouping on stage-2 names (optim_factory.py:44-73)
    nl = 2
    assigner = optim_ref.LayerDecayValueAssigner([0.65 ** (nl + 1 - i) for i in range(nl + 2)])
    lgroups = optim_ref.get_parameter_groups(vit, 0.05, vit.no_weight_decay(), assigner.get_layer_id, assigner.get_scale)
    vnames = dict((id(p), n) for n, p in vit.named_parameters())
    print("vit_stage2_tiny loss", ce.item(), "layer groups", len(lgroups))

    # ------------------------------------------------------------ 4. utils: schedules, greedy masks
    sched = utils_ref.cosine_scheduler(1.5e-4, 1e-5, 4, 5, warmup_epochs=1)
    sched_ws = utils_ref.cosine_scheduler(1.5e-4, 1e-5, 3, 7, warmup_epochs=1, warmup_steps=4, start_warmup_value=1e-6)
    g = torch.Generator().manual_seed(7)
    attn_r = torch.rand(6, 16, generator=g)
    gm = utils_ref.get_greedy_masks(attn_r, 0.75, 2)
    gm3 = utils_ref.get_greedy_masks(attn_r, 0.8, 3)
    np.savez_compressed(
        os.path.join(OUT, "utils.npz"),
        **_np({"cos.a": sched, "cos.b": sched_ws, "greedy.attn": attn_r, "greedy.k2_r075": gm, "greedy.k3_r08": gm3,
               "layer.names": np.array([vnames[id(p)] for gr in lgroups for p in gr["params"]]),
               "layer.scale": np.array([gr["lr_scale"] for gr in lgroups for _ in gr["params"]]),
               "layer.wd": np.array([gr["weight_decay"] for gr in lgroups for _ in gr["params"]])}))
    print("utils", sched.shape, gm.shape)

    # ------------------------------------------------------------ 5. full-size ViT-B/16 + CLIP-B/16 (BASELINE cfg 1, B=2)
    teacher_b = clip_ref.clip_b16(pretrained=False, return_attn=True, clip_return_layers=[6, 7, 8, 9, 10, 11])
    student_b = ad_ref.adaptation_umt_base_patch16_224(
        num_frames=8, tubelet_size=1, drop_path_rate=0.0, clip_decoder_embed_dim=768, clip_output_dim=512,
        clip_return_layers=[6, 7, 8, 9, 10, 11], use_cls_token=False).train()
    teacher_b.load_state_dict(fill_state_dict(_shapes(teacher_b), seed=11))
    student_b.load_state_dict(fill_state_dict(_shapes(student_b), seed=12))
    Bf = 2
    vidf = make_videos(Bf, 8, 224, 224, seed=13)
    impf = make_importance(Bf * 8, 196, seed=14)
    lossf, outf, tgtf, attnf, normf, maskf = _ref_stage1_step(student_b, teacher_b, vidf, impf, 0.8)
    lossf.backward()
    gn = utils_ref.get_grad_norm_(student_b.parameters()).item()
    sel = ["encoder.patch_embed.proj.weight", "encoder.blocks.0.attn.qkv.weight", "encoder.blocks.0.attn.q_bias",
           "encoder.blocks.5.mlp.fc1.weight", "encoder.blocks.11.mlp.fc2.bias", "encoder.blocks.11.norm2.weight",
           "encoder.norm.weight", "clip_decoder.0.head.weight", "clip_decoder.5.norm.bias"]
    pg = dict(student_b.named_parameters())
    np.savez_compressed(
        os.path.join(OUT, "stage1_vitb_cfg1.npz"),
        **_np({"in.B": Bf, "in.seed_teacher": 11, "in.seed_student": 12, "in.seed_videos": 13, "in.seed_importance": 14,
               "in.mask_ratio": 0.8,
               "out.loss": lossf, "out.grad_norm": gn,
               "out.attn": attnf,                                   # (16,196)
               "out.targets_corner": tgtf[:, :, :8, :8], "out.x_clip_corner": outf[:, :, :8, :8],
               "out.targets_mean": tgtf.mean(dim=(2, 3)), "out.x_clip_rowdot": (outf * tgtf).sum(-1).mean(dim=2),
               "out.feats_tapnorm": normf.abs().mean(dim=(1, 2, 3))}),
        **_np({"gnorm." + k: pg[k].grad.norm() for k in sel}),
        **_np({"gcorner." + k: pg[k].grad.reshape(pg[k].shape[0], -1)[:8, :8] for k in sel}))
    print("stage1_vitb_cfg1 loss", lossf.item(), "grad_norm", gn, "n_params", sum(p.numel() for p in student_b.parameters()))


if __name__ == "__main__":
    main()
