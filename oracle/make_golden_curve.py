#!/usr/bin/env python
"""Loss-curve fixture: N optimisation steps of the REFERENCE's own ViT-B/16 student + CLIP-B/16 teacher on CPU.

TEST INFRASTRUCTURE.  Runs only in the build container (needs /root/reference); only inputs (seeds) and outputs
(per-step loss / grad-norm / lr, a few parameter slices after the last step) are committed -> tests/golden/stage1_curve.npz.

What runs is the loop of run_stage1.py:294-505 restated around the reference's classes (the script itself cannot be imported:
wandb / decord), every numeric piece being the reference's own code:
  * models        src/models/clip.py::clip_b16, src/models/modeling_adaptation.py::adaptation_umt_base_patch16_224
  * lr schedule   src/utils.py::cosine_scheduler, written into param_groups per step as run_stage1.py:326-338
  * optimizer     src/optim_factory.py::create_optimizer (torch.optim.AdamW, betas (0.9, 0.95), wd 0.05, no-decay groups)
  * grad norm     src/utils.py::get_grad_norm_ (what NativeScalerWithGradNormCount returns with clip_grad=None, utils.py:608-622)
  * the step      run_stage1.py:360-456; torch.multinomial (:382) replaced by a stored permutation per step (seed 100 + step)
Fresh clips every step (seed 200 + step), drop_path 0 (timm's drop_path is not in the reference tree: parity unpinned).

Usage:  python oracle/make_golden_curve.py [steps]
"""
from __future__ import annotations

import os
import sys
import time
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from oracle import make_golden as G  # noqa: E402
from oracle.filler import fill_state_dict, make_importance, make_videos  # noqa: E402

STEPS, B, LR, MIN_LR, WARMUP = 24, 2, 1e-3, 1e-5, 4
SEL = ["encoder.patch_embed.proj.bias", "encoder.blocks.0.attn.qkv.weight", "encoder.blocks.6.mlp.fc1.weight",
       "encoder.blocks.11.mlp.fc2.bias", "encoder.norm.weight", "clip_decoder.0.head.weight", "clip_decoder.5.norm.bias"]


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else STEPS
    torch.manual_seed(0)
    torch.set_num_threads(8)
    G._install_standins()
    clip_ref = G._load("src.models.clip", "src/models/clip.py")
    G._load("src.models.modeling_finetune", "src/models/modeling_finetune.py")
    ad_ref = G._load("src.models.modeling_adaptation", "src/models/modeling_adaptation.py")
    utils_ref = G._load("src.utils", "src/utils.py")
    optim_ref = G._load("src.optim_factory", "src/optim_factory.py")

    teacher = clip_ref.clip_b16(pretrained=False, return_attn=True, clip_return_layers=[6, 7, 8, 9, 10, 11])
    student = ad_ref.adaptation_umt_base_patch16_224(
        num_frames=8, tubelet_size=1, drop_path_rate=0.0, clip_decoder_embed_dim=768, clip_output_dim=512,
        clip_return_layers=[6, 7, 8, 9, 10, 11], use_cls_token=False).train()
    teacher.load_state_dict(fill_state_dict(G._shapes(teacher), seed=11))
    student.load_state_dict(fill_state_dict(G._shapes(student), seed=12))
    args = types.SimpleNamespace(opt="adamw", weight_decay=0.05, lr=LR, opt_eps=1e-8, opt_betas=[0.9, 0.95], momentum=0.9)
    opt = optim_ref.create_optimizer(args, student, skip_list=student.no_weight_decay())
    # (the reference honours warmup_steps only when warmup_epochs > 0: utils.py:650-654)
    lr_sched = utils_ref.cosine_scheduler(LR, MIN_LR, 2, steps // 2, warmup_epochs=1, warmup_steps=WARMUP)
    losses, gnorms = [], []
    t0 = time.time()
    for it in range(steps):
        for g in opt.param_groups:                                  # run_stage1.py:326-338
            g["lr"] = lr_sched[it] * g.get("lr_scale", 1.0)
        vid = make_videos(B, 8, 224, 224, seed=200 + it)
        imp = make_importance(B * 8, 196, seed=100 + it)
        loss, *_ = G._ref_stage1_step(student, teacher, vid, imp, 0.8)
        opt.zero_grad()
        loss.backward()
        gnorms.append(utils_ref.get_grad_norm_(student.parameters()).item())
        opt.step()
        losses.append(loss.item())
        print(f"step {it:2d} lr {lr_sched[it]:.2e} loss {losses[-1]:.6f} grad_norm {gnorms[-1]:.6f}  ({time.time() - t0:.0f} s)", flush=True)
    sd = student.state_dict()
    np.savez_compressed(
        os.path.join(G.OUT, "stage1_curve.npz"),
        **G._np({"in.B": B, "in.steps": steps, "in.seed_teacher": 11, "in.seed_student": 12, "in.seed_videos0": 200,
                 "in.seed_importance0": 100, "in.mask_ratio": 0.8, "opt.lr": LR, "opt.min_lr": MIN_LR, "opt.warmup_steps": WARMUP,
                 "opt.wd": 0.05, "opt.betas": np.array([0.9, 0.95]), "opt.eps": 1e-8,
                 "out.lr": np.asarray(lr_sched[:steps]), "out.loss": np.array(losses), "out.grad_norm": np.array(gnorms)}),
        **G._np({"after." + k: sd[k].reshape(sd[k].shape[0], -1)[:8, :8] for k in SEL}))
    print("stage1_curve", losses[0], "->", losses[-1])


if __name__ == "__main__":
    main()
