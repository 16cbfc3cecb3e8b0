"""Stage-1 (Unmasked-Teacher distillation) training engine -- drop-in for ``train_one_epoch`` of reference
run_stage1.py:294-505: same signature, same per-step semantics, same returned ``{meter: global_avg}`` dict.

Per step (reference line numbers in brackets):
  lr / weight-decay write-back into param_groups                      [:326-338]
  teacher forward -> CLS attention of the last block                  [:360-377]   teacher.forward_attention
  attention-guided mask: N_vis = N - int(N*ratio) visible per frame   [:379-387]   unite_mask_sample (device, no sync)
  targets = L2-normalised teacher features of the visible tokens      [:389-397]   teacher.visible_targets (visible rows only)
  student forward on the visible tokens + decoders + UMT loss         [:410-438]   model.forward_loss
  zero_grad, backward, grad-norm (or clip), AdamW                     [:451-456]   loss_scaler(...)
What changed, deliberately: no host sync per step (the reference syncs 3x: boolean indexing :393, loss.item() :440,
cuda.synchronize :458) -- loss / grad-norm scalars stay on the device and are read back every ``log_freq`` steps,
when the finite-loss check (:447-449) also runs; the mask never visits the host.
"""
from __future__ import annotations

import math
import os
import sys
import time
from typing import Iterable, Optional

import torch

from . import ops, utils


class StepState:
    """Device buffers of the stage-1 step that persist across iterations."""

    def __init__(self):
        self.mask = None
        self.vis = None
        self.rows = None
        self.seed = 0
        self.step_params = None          # graph_step.StepParams: the mask sampler then reads its seed from device memory
        self.overlap_targets = os.environ.get("UNITE_OVERLAP_TARGETS", "1") != "0"


def teacher_input(teacher_model, videos, clip_input_resolution):
    """run_stage1.py:362-370 / run_stage3.py:438-447: bicubic resize of every frame plane when the teacher's resolution differs
    (224 -> 196 for CLIP-L/14 so that its 14 x 14 grid matches the student's)."""
    B, C, T, H, W = videos.shape
    if H == clip_input_resolution:
        return videos
    rt = getattr(teacher_model, "module", teacher_model).runtime()
    out = rt.ws.get("resized", (B, C, T, clip_input_resolution, clip_input_resolution), torch.float32)
    return ops.resize_bicubic(videos.contiguous(), out)


def stage1_step(model, teacher_model, videos, n_source, mask_ratio, mask_type, bool_masked_pos, clip_loss_data, state: StepState,
                clip_input_resolution=224, importance=None):
    """teacher -> mask -> targets -> student loss (device tensors only).  Returns the 0-dim loss tensor (with grad_fn)."""
    student = getattr(model, "module", model)
    rt = student.runtime()
    dev = videos.device
    B, C, T, H, W = videos.shape
    attn = teacher_model.forward_attention(teacher_input(teacher_model, videos, clip_input_resolution))     # (B*T, N) f32
    BT, N = attn.shape
    n_vis_frame = N - int(N * mask_ratio)                        # :380
    n_vis = n_vis_frame * (BT // B)
    if state.mask is None or state.mask.numel() != BT * N:
        state.mask = torch.empty(BT * N, dtype=torch.uint8, device=dev)
        state.vis = torch.empty(BT * n_vis_frame, dtype=torch.int32, device=dev)
        state.rows = torch.empty(BT * n_vis_frame, dtype=torch.int32, device=dev)
    if importance is not None:                                   # explicit permutation (parity tests)
        ops.mask_from_importance(importance, state.mask, state.vis, n_vis_frame, vis_rows_cls=state.rows)
    elif mask_type == 'attention':
        if state.step_params is not None:      # captured step: the host advanced and published the seed before the launch
            ops.mask_sample(attn, 0, state.mask, state.vis, n_vis_frame, vis_rows_cls=state.rows, seed_dev=state.step_params.seed_mask_dev)
        else:
            state.seed += 1
            ops.mask_sample(attn, state.seed, state.mask, state.vis, n_vis_frame, vis_rows_cls=state.rows)   # :382-387
    else:
        m8 = bool_masked_pos.to(dev).flatten(1).to(torch.uint8).contiguous().view(-1)
        state.mask = m8
        ops.mask_to_tokens(m8, state.vis, n_vis_frame, BT, N, vis_rows_cls=state.rows)
    M = B * n_vis
    # The teacher's tail (last block on the visible rows, ln_post, proj, L2) is only needed by the loss: it runs on the teacher's
    # side stream under the student's encoder forward.
    trt = getattr(teacher_model, "module", teacher_model).runtime()
    ready = None
    if state.overlap_targets and videos.is_cuda:
        main, side = torch.cuda.current_stream(), trt._side_stream()
        ev = torch.cuda.Event()
        ev.record(main)
        side.wait_event(ev)
        with torch.cuda.stream(side):
            targets = teacher_model.visible_targets(state.rows, M)       # f32 [K*M, C], rows in (k, b, token) order
            ready = torch.cuda.Event()
            ready.record(side)
    else:
        targets = teacher_model.visible_targets(state.rows, M)
    # which clips take part in the loss (:418-427).  Samples are independent in the student, so restricting the loss to a
    # slice of the batch equals running the student on that slice only.
    if clip_loss_data == 'mixed':
        lo, hi = 0, B
    elif clip_loss_data == 'source':
        lo, hi = 0, n_source
    elif clip_loss_data == 'target':
        lo, hi = n_source, B
    else:
        raise NotImplementedError
    if hi <= lo:
        raise ValueError(f"clip_loss_data='{clip_loss_data}' selects no clip (no target loader?): the reference's loss is NaN here")
    if (lo, hi) != (0, B):
        if ready is not None:
            torch.cuda.current_stream().wait_event(ready)
            ready = None
        K = targets.shape[0] // M
        targets = targets.view(K, B, n_vis, -1)[:, lo:hi].contiguous().view(K * (hi - lo) * n_vis, -1)
        videos_l = videos[lo:hi].contiguous()
        vis_l = (state.vis.view(B, n_vis)[lo:hi] - lo * T * N).contiguous().view(-1)
        return model.forward_loss(videos_l, vis_l, n_vis, targets)
    return model.forward_loss(videos, state.vis, n_vis, targets, targets_ready=ready)


def train_one_epoch(model: torch.nn.Module, data_loader: Iterable, data_loader_train_target: Optional[Iterable],
                    optimizer: torch.optim.Optimizer, device: torch.device, epoch: int, loss_scaler, max_norm: float = 0,
                    log_writer=None, lr_scheduler=None, start_steps=None, lr_schedule_values=None, wd_schedule_values=None,
                    src_classifier=None, teacher_model=None, clip_input_resolution=224, clip_loss_type='l2', clip_loss_ratio=0.5,
                    mask_type='tube', mask_ratio=0., use_wandb=False, args=None):
    if clip_loss_type != 'l2':
        raise NotImplementedError("only clip_loss_type='l2' (every UNITE config) is built")
    if src_classifier is not None:
        raise NotImplementedError("stage 1 is run with src_classifier=None (run_stage1.py:858)")
    model.train()
    metric_logger = utils.MetricLogger(delimiter="  ")
    metric_logger.add_meter('lr', utils.SmoothedValue(window_size=1, fmt='{value:.6f}'))
    metric_logger.add_meter('min_lr', utils.SmoothedValue(window_size=1, fmt='{value:.6f}'))
    header = 'Epoch [{}]:'.format(epoch)
    ipe = len(data_loader)
    print_freq = args.log_freq
    start_steps = start_steps or 0
    reducer = getattr(model, "reducer", None)
    state = getattr(model, "_unite_stage1_state", None)
    if state is None:
        state = StepState()
        state.seed = int(getattr(args, "seed", 0)) * 1000003 + utils.get_rank() * 7919 + epoch * 104729
        model._unite_stage1_state = state

    target_iter = iter(data_loader_train_target) if data_loader_train_target is not None else None
    pending = []          # (loss, grad_norm) device scalars not yet read back

    def flush():
        if not pending:
            return
        vals = torch.stack([torch.stack([l.detach().float(), g.detach().float()]) for l, g in pending]).tolist()   # one sync
        pending.clear()
        for lv, gv in vals:
            if not math.isfinite(lv):
                print("Loss is {}, stopping training".format(lv))
                sys.exit(1)                                          # :447-449
            metric_logger.update(loss=lv, loss_clip=lv, grad_norm=gv)
            if log_writer is not None:
                log_writer.update(loss=lv, head="loss")
                log_writer.update(loss_clip=lv, head="loss_clip")
                log_writer.update(grad_norm=gv, head="opt")
                log_writer.set_step()

    for step, batch in enumerate(metric_logger.log_every(data_loader, print_freq, getattr(args, "epochs", None), epoch, ipe, header=header)):
        it = start_steps + step
        if lr_schedule_values is not None or wd_schedule_values is not None:
            for param_group in optimizer.param_groups:
                if lr_schedule_values is not None:
                    param_group["lr"] = lr_schedule_values[min(it, len(lr_schedule_values) - 1)] * param_group["lr_scale"]
                if wd_schedule_values is not None and param_group["weight_decay"] > 0:
                    param_group["weight_decay"] = wd_schedule_values[min(it, len(wd_schedule_values) - 1)]

        videos, bool_masked_pos, labels_s = batch
        B_s = videos.shape[0]
        if target_iter is not None:
            try:
                videos_t, bool_masked_pos_t, _ = next(target_iter)
            except StopIteration:
                target_iter = iter(data_loader_train_target)
                videos_t, bool_masked_pos_t, _ = next(target_iter)
            videos = torch.cat([videos, videos_t], dim=0)
            if mask_type != 'attention':
                bool_masked_pos = torch.cat([bool_masked_pos, bool_masked_pos_t], dim=0)
        videos = videos.to(device, non_blocking=True)

        loss = stage1_step(model, teacher_model, videos, B_s, mask_ratio, mask_type, bool_masked_pos,
                           getattr(args, "clip_loss_data", "mixed"), state, clip_input_resolution)
        optimizer.zero_grad()
        grad_norm = loss_scaler(loss, optimizer, clip_grad=max_norm, parameters=None, create_graph=False, reducer=reducer)
        pending.append((loss, grad_norm))

        if print_freq and (step % print_freq == 0 or step == ipe - 1):
            flush()
        min_lr, max_lr = 10., 0.
        for group in optimizer.param_groups:
            min_lr, max_lr = min(min_lr, group["lr"]), max(max_lr, group["lr"])
        weight_decay_value = None
        for group in optimizer.param_groups:
            if group["weight_decay"] > 0:
                weight_decay_value = group["weight_decay"]
        metric_logger.update(lr=max_lr, min_lr=min_lr, weight_decay=weight_decay_value, loss_scale=loss_scaler.state_dict()["scale"])
        if lr_scheduler is not None:
            lr_scheduler.step_update(start_steps + step)
    flush()
    metric_logger.synchronize_between_processes()
    print(f"[{time.strftime('%Y-%m-%d %H:%M:%S', time.localtime())}] Averaged stats:", metric_logger)
    return {k: meter.global_avg for k, meter in metric_logger.meters.items()}
