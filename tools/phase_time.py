#!/usr/bin/env python
"""Stage-1 step by phase on one GPU (B = 32): teacher only, teacher + student forward, + backward, + grad-norm / AdamW.
Each phase set is its own steady-state loop (differences between the lines are the phases' costs with the streams on)."""
import os, sys, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from types import SimpleNamespace
import unite_amd
from unite_amd.engine_stage1 import StepState, stage1_step
from unite_amd.optim_factory import create_optimizer
from unite_amd.utils import NativeScalerWithGradNormCount

dev = torch.device("cuda", 0)
B, T = int(os.environ.get("B", 32)), 8
with contextlib.redirect_stdout(sys.stderr):
    student = unite_amd.create_model(
        "adaptation_umt_base_patch16_224", pretrained=False, drop_path_rate=0.1, drop_block_rate=None, use_learnable_pos_emb=False,
        use_checkpoint=False, checkpoint_num=0, clip_decoder_embed_dim=768, clip_output_dim=512, clip_norm_type='l2', num_frames=T,
        tubelet_size=1, clip_return_layers=[6, 7, 8, 9, 10, 11], clip_student_return_interval=1, use_cls_token=False).to(dev).train()
    teacher = unite_amd.clip.clip_b16(pretrained=False, return_attn=True, clip_return_layers=[6, 7, 8, 9, 10, 11]).to(dev)
    args = SimpleNamespace(opt="adamw", weight_decay=0.05, lr=1.5e-4 * B / 256, opt_eps=1e-8, opt_betas=[0.9, 0.95])
    opt = create_optimizer(args, student, skip_list=student.no_weight_decay())
scaler = NativeScalerWithGradNormCount()
videos = torch.randn(B, 3, T, 224, 224, device=dev)
state = StepState()


def timed(fn, n=20, w=5):
    for _ in range(w):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def teacher_only():
    teacher.forward_attention(videos)


def fwd():
    with torch.no_grad():
        pass
    return stage1_step(student, teacher, videos, B, 0.8, 'attention', None, 'mixed', state)


def fwd_bwd():
    loss = fwd()
    student.runtime().fp.accumulate = False
    loss.backward()


def full():
    loss = fwd()
    opt.zero_grad()
    scaler(loss, opt, clip_grad=None, parameters=None)


t_t, t_f, t_fb, t_all = timed(teacher_only), timed(fwd), timed(fwd_bwd), timed(full)
print(f"teacher (all ranges, to the CLS attention)   {t_t:7.2f} ms")
print(f"+ mask, targets, student forward, loss        {t_f:7.2f} ms  (+{t_f - t_t:.2f})")
print(f"+ student backward                            {t_fb:7.2f} ms  (+{t_fb - t_f:.2f})")
print(f"+ grad-norm, AdamW  (= the step)              {t_all:7.2f} ms  (+{t_all - t_fb:.2f})")
