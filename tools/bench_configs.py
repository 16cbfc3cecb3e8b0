#!/usr/bin/env python
"""Step timing of the non-headline BASELINE configs on ONE GPU (per-GPU shapes of configs[2..4]); prints one JSON line each.
    python tools/bench_configs.py stage3 [--teacher clip_b16|clip_l14] [--batch 16]
    python tools/bench_configs.py vitl   [--batch 8]
    python tools/bench_configs.py stage2 [--batch 16]
`python bench.py --config 3|4|5` runs the same code (run()) and wraps the result in bench.py's record format.
FLOP counts are SURVEY.md 8(d)'s algorithmic figures."""
import argparse
import contextlib, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from types import SimpleNamespace

PEAK = 2.5e15


def lin_flops(n, D, depth):            # per clip forward: qkv + proj + mlp, n tokens
    return depth * (2 * n * D * 3 * D + 2 * n * D * D + 4 * n * D * 4 * D)


def attn_flops(n, D, depth, frames=1):  # per clip forward; `frames` independent sequences of n tokens
    return depth * frames * 4 * n * n * D


def run(config, batch=None, teacher_name="clip_l14", steps=8, warmup=3):
    """one of "stage2" | "stage3" | "vitl" on cuda: `warmup` untimed + `steps` timed steps -> dict (ms_per_step, clips_per_s, mfma_frac, ...)"""
    import unite_amd
    from unite_amd.optim_factory import create_optimizer, LayerDecayValueAssigner
    from unite_amd.utils import NativeScalerWithGradNormCount
    dev = torch.device("cuda")
    scaler = NativeScalerWithGradNormCount()
    if config == "stage2":
        B, T = batch or 16, 16
        m = unite_amd.create_model("vit_base_patch16_224", pretrained=False, num_classes=8, all_frames=T, tubelet_size=1, drop_path_rate=0.1,
                                   use_mean_pooling=True, init_scale=0.001, use_learnable_pos_emb=False, fc_drop_rate=0.0, drop_rate=0.0,
                                   attn_drop_rate=0.0, use_checkpoint=False, checkpoint_num=0).to(dev).train()
        nl = m.get_num_layers()
        asg = LayerDecayValueAssigner([0.65 ** (nl + 1 - i) for i in range(nl + 2)])
        args = SimpleNamespace(opt="adamw", weight_decay=0.05, lr=1e-3 * B / 256, opt_eps=1e-8, opt_betas=[0.9, 0.999])
        opt = create_optimizer(args, m, skip_list=m.no_weight_decay(), get_num_layer=asg.get_layer_id, get_layer_scale=asg.get_scale)
        vid = torch.randn(B, 3, T, 224, 224, device=dev)
        lab = torch.randint(0, 8, (B,), device=dev)
        flops = 2693e9

        def step():
            opt.zero_grad()
            loss, _ = m.forward_loss(vid, lab)
            return loss, scaler(loss, opt, clip_grad=None)
        name, units = "stage2 ViT-B/16 16fx224^2 (3136 tokens), 8 classes", B
    elif config == "stage3":
        from unite_amd.engine_stage3 import stage3_step
        B, T = batch or 16, 8
        student = unite_amd.create_model("adaptation_umt_base_patch16_224", pretrained=False, drop_path_rate=0.1, drop_block_rate=None,
                                         use_learnable_pos_emb=False, use_checkpoint=False, checkpoint_num=0, clip_decoder_embed_dim=768,
                                         clip_output_dim=512, clip_norm_type='l2', num_frames=T, tubelet_size=1, clip_return_layers=[6],
                                         clip_student_return_interval=1, use_cls_token=False).to(dev).train()
        if teacher_name == "clip_l14":
            teacher, res = unite_amd.clip.clip_l14(pretrained=False, input_resolution=196, return_attn=True, clip_return_layers=[6]).to(dev), 196
            t_flops = T * (2 * 196 * 588 * 1024 + lin_flops(197, 1024, 24) + attn_flops(197, 1024, 24))
        else:
            teacher, res = unite_amd.clip.clip_b16(pretrained=False, return_attn=True, clip_return_layers=[6]).to(dev), 224
            t_flops = T * (2 * 196 * 768 * 768 + lin_flops(197, 768, 12) + attn_flops(197, 768, 12))
        cls = torch.nn.Linear(768, 8).to(dev)
        args = SimpleNamespace(opt="adamw", weight_decay=0.05, lr=1e-5, opt_eps=1e-8, opt_betas=[0.9, 0.999], masking_type="clip_attention",
                               selection_strategy="clip_matchORconf", clip_threshold=0.5, conf_weighted_loss=True, class_loss_tgt_ratio=1.0,
                               class_loss_src_ratio_pl=1.0, train_masked=True, full_oracle=False)
        opt = create_optimizer(args, student, skip_list=student.no_weight_decay())
        opt.set_unused(("clip_decoder.",))
        vs, vt, va = (torch.randn(B, 3, T, 224, 224, device=dev) for _ in range(3))
        ls, lt = torch.randint(0, 8, (B,), device=dev), torch.randint(0, 8, (B,), device=dev)
        probs = torch.rand(B, 8, device=dev).softmax(-1)          # injected zero-shot similarities (text tower out of scope)
        full = lin_flops(1568, 768, 12) + attn_flops(1568, 768, 12) + 2 * 1568 * 768 * 768
        memb = lin_flops(320, 768, 12) + attn_flops(320, 768, 12) + 2 * 320 * 768 * 768
        # per (src, tgt) pair: src fwd+bwd, tgt fwd, committee member 0 fwd, member 1 fwd+bwd, mask teacher fwd
        flops = 3 * full + full + memb + 3 * memb + t_flops

        from unite_amd.engine_stage3 import MaskTeacherAhead
        ahead = MaskTeacherAhead(teacher, student, dev, 0.8, "clip_attention", res) if os.environ.get("UNITE_TEACHER_AHEAD", "1") != "0" else None
        mouts = []

        def step():
            m = None
            if ahead is not None:          # the default schedule of engine_stage3.train_one_epoch: mask teacher of the next batch beside this step
                if not mouts:
                    mouts.append(ahead.launch(va, inputs_ready=False))
                m = mouts.pop()
                mouts.append(ahead.launch(va, inputs_ready=False))
            with (ahead.student() if ahead is not None else contextlib.nullcontext()):
                loss, *_ = stage3_step(student, teacher, cls, vs, ls, vt, va, lt, args, 0.8, clip_probs_fn=lambda v: probs, clip_input_resolution=res, masks=m)
                opt.zero_grad()
                return loss, scaler(loss, opt, clip_grad=None)
        name, units = f"stage3 ViT-B/16 student + {teacher_name} mask teacher, 8fx224^2, B={B} src + {B} tgt (zero-shot CLIP probabilities injected)", B
    else:
        from unite_amd.engine_stage1 import stage1_step, StepState
        B, T = batch or 8, 16
        taps = [18, 19, 20, 21, 22, 23]
        student = unite_amd.create_model("adaptation_umt_large_patch16_224", pretrained=False, drop_path_rate=0.1, num_frames=T, tubelet_size=1,
                                         clip_decoder_embed_dim=1024, clip_output_dim=768, clip_return_layers=taps, use_cls_token=False,
                                         use_learnable_pos_emb=False, use_checkpoint=False, checkpoint_num=0, clip_norm_type='l2',
                                         clip_student_return_interval=1, drop_block_rate=None).to(dev).train()
        teacher = unite_amd.clip.clip_l14(pretrained=False, input_resolution=196, return_attn=True, clip_return_layers=taps).to(dev)
        args = SimpleNamespace(opt="adamw", weight_decay=0.05, lr=1.5e-4 * B / 256, opt_eps=1e-8, opt_betas=[0.9, 0.95])
        opt = create_optimizer(args, student, skip_list=student.no_weight_decay())
        vid = torch.randn(B, 3, T, 224, 224, device=dev)
        st = StepState()
        n = 40 * T
        s_f = lin_flops(n, 1024, 24) + attn_flops(n, 1024, 24) + 6 * 2 * n * 1024 * 768
        t_flops = T * (2 * 196 * 588 * 1024 + lin_flops(197, 1024, 24) + attn_flops(197, 1024, 24)) + 6 * 2 * n * 1024 * 768
        flops = 3 * s_f + 2 * (2 * n * 768 * 1024) + t_flops

        from unite_amd.engine_stage1 import TeacherAhead, student_phase
        ahead = TeacherAhead(teacher, st, dev, 0.8, 'attention', clip_input_resolution=196) if os.environ.get("UNITE_TEACHER_AHEAD", "1") != "0" else None
        touts = []

        def step():
            if ahead is not None:      # the default schedule of train_one_epoch: teacher of the next batch beside the student of this one
                if not touts:
                    touts.append(ahead.launch(vid, inputs_ready=False))
                cur = touts.pop()
                touts.append(ahead.launch(vid, inputs_ready=False))
                with ahead.student():
                    loss = student_phase(student, vid, cur, B, 'mixed')
                    opt.zero_grad()
                    return loss, scaler(loss, opt, clip_grad=None)
            loss = stage1_step(student, teacher, vid, B, 0.8, 'attention', None, 'mixed', st, clip_input_resolution=196)
            opt.zero_grad()
            return loss, scaler(loss, opt, clip_grad=None)
        name, units = "stage1 ViT-L/16 student (16fx224^2, 640 visible tokens) + CLIP-L/14 teacher @196", B

    torch.cuda.synchronize()          # the synthetic clips are complete before the first teacher launch reads them on its own stream
    for _ in range(warmup):
        loss, gn = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss, gn = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return {"workload": name, "batch": B, "ms_per_step": round(dt * 1e3, 2), "clips_per_s": round(units / dt, 1),
            "gflop_per_clip": round(flops / 1e9, 1), "mfma_frac": round(units / dt * flops / PEAK, 4),
            "loss": round(loss.item(), 5), "grad_norm": round(float(gn), 5),
            "hbm_peak_GiB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("config", choices=["stage2", "stage3", "vitl"])
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--teacher", default="clip_l14")
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=3)
    a = ap.parse_args()
    print(json.dumps(run(a.config, a.batch, a.teacher, a.steps, a.warmup)))


if __name__ == "__main__":
    main()
