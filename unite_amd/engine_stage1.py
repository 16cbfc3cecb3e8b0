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
        self.mask = None                 # the buffers of the slot used last (tests read them)
        self.vis = None
        self.rows = None
        self.slots = {}                  # slot -> (mask, vis, rows): three slots when the teacher runs ahead of the student (TeacherAhead)
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


class TeacherOut:
    """What the student needs from the teacher for one batch: the visible-token list and the targets of those tokens."""
    __slots__ = ("mask", "vis", "rows", "n_vis", "targets", "ready", "foreign", "videos")   # ready: event behind `targets`; foreign: ... and behind vis too


def teacher_phase(teacher_model, videos, mask_ratio, mask_type, bool_masked_pos, state: StepState, clip_input_resolution=224,
                  importance=None, slot=0, inline_targets=False) -> TeacherOut:
    """teacher forward -> attention-guided mask -> targets of the visible tokens (run_stage1.py:360-397).  Everything is enqueued on
    the current stream, except that with ``state.overlap_targets`` (and not ``inline_targets``) the target tail goes to the teacher's
    side stream and ``ready`` is the event the consumer waits for."""
    dev = videos.device
    B, C, T, H, W = videos.shape
    attn = teacher_model.forward_attention(teacher_input(teacher_model, videos, clip_input_resolution))     # (B*T, N) f32
    BT, N = attn.shape
    n_vis_frame = N - int(N * mask_ratio)                        # :380
    n_vis = n_vis_frame * (BT // B)
    bufs = state.slots.get(slot)
    if bufs is None or bufs[0].numel() != BT * N or bufs[1].numel() != BT * n_vis_frame:
        bufs = (torch.empty(BT * N, dtype=torch.uint8, device=dev), torch.empty(BT * n_vis_frame, dtype=torch.int32, device=dev),
                torch.empty(BT * n_vis_frame, dtype=torch.int32, device=dev))
        state.slots[slot] = bufs
    mask, vis, rows = bufs
    if importance is not None:                                   # explicit permutation (parity tests)
        ops.mask_from_importance(importance, mask, vis, n_vis_frame, vis_rows_cls=rows)
    elif mask_type == 'attention':
        if state.step_params is not None:      # captured step: the host advanced and published the seed before the launch
            ops.mask_sample(attn, 0, mask, vis, n_vis_frame, vis_rows_cls=rows, seed_dev=state.step_params.seed_mask_dev)
        else:
            state.seed += 1
            ops.mask_sample(attn, state.seed, mask, vis, n_vis_frame, vis_rows_cls=rows)   # :382-387
    else:
        mask = bool_masked_pos.to(dev).flatten(1).to(torch.uint8).contiguous().view(-1)
        ops.mask_to_tokens(mask, vis, n_vis_frame, BT, N, vis_rows_cls=rows)
    state.mask, state.vis, state.rows = mask, vis, rows
    out = TeacherOut()
    out.mask, out.vis, out.rows, out.n_vis, out.ready, out.foreign, out.videos = mask, vis, rows, n_vis, None, False, videos
    M = B * n_vis
    # The teacher's tail (last block on the visible rows, ln_post, proj, L2) is only needed by the loss: it runs on the teacher's
    # side stream under the student's encoder forward.
    trt = getattr(teacher_model, "module", teacher_model).runtime()
    if state.overlap_targets and videos.is_cuda and not inline_targets:
        main, side = torch.cuda.current_stream(), trt._side_stream()
        ev = torch.cuda.Event()
        ev.record(main)
        side.wait_event(ev)
        with torch.cuda.stream(side):
            out.targets = teacher_model.visible_targets(rows, M, slot=slot)      # f32 [K*M, C], rows in (k, b, token) order
            out.ready = torch.cuda.Event()
            out.ready.record(side)
    else:
        out.targets = teacher_model.visible_targets(rows, M, slot=slot)
    return out


class _PointwiseLoss(torch.autograd.Function):
    """mean over all elements of f(outputs_clip - targets_clip): nn.MSELoss / nn.L1Loss / nn.SmoothL1Loss of run_stage1.py:403-408,433-434
    in one kernel that also leaves the gradient (unite_pointwise_loss)."""

    @staticmethod
    def forward(ctx, out, target, kind):
        out = out.contiguous()
        loss_sum = torch.zeros(1, dtype=torch.float32, device=out.device)
        grad = torch.empty_like(out)
        ops.pointwise_loss(out, target.contiguous().view_as(out), kind, loss_sum, grad, 1.0 / out.numel())
        ctx.save_for_backward(grad)
        return loss_sum[0] / out.numel()

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None, None


def student_phase(model, videos, tout: TeacherOut, n_source, clip_loss_data, clip_loss_type='l2'):
    """student forward on the visible tokens + decoders + UMT loss (run_stage1.py:410-438).  Returns the 0-dim loss (with grad_fn).
    clip_loss_type 'l2' (every shipped config) is fused into the decoders' tail kernel; 'mse' / 'l1' / 'smooth_l1' go through the decoder
    outputs (K, B, n_vis, C) and unite_pointwise_loss."""
    B, C, T, H, W = videos.shape
    n_vis, targets, ready = tout.n_vis, tout.targets, tout.ready
    if tout.foreign and ready is not None:         # the whole teacher phase ran on another stream: the token list is needed first
        torch.cuda.current_stream().wait_event(ready)
        ready = None
        if videos.is_cuda:
            videos.record_stream(torch.cuda.current_stream())       # it may have been allocated on the teacher's stream (host batch)
    M = B * n_vis
    N = tout.mask.numel() // (B * T)
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
        vis_l = (tout.vis.view(B, n_vis)[lo:hi] - lo * T * N).contiguous().view(-1)
        if clip_loss_type != 'l2':
            return _other_clip_loss(model, videos_l, vis_l, n_vis, targets, clip_loss_type)
        return model.forward_loss(videos_l, vis_l, n_vis, targets)
    if clip_loss_type != 'l2':
        if ready is not None:
            torch.cuda.current_stream().wait_event(ready)
        return _other_clip_loss(model, videos, tout.vis, n_vis, targets, clip_loss_type)
    return model.forward_loss(videos, tout.vis, n_vis, targets, targets_ready=ready)


def _other_clip_loss(model, videos, vis, n_vis, targets, clip_loss_type):
    if clip_loss_type not in ops.POINTWISE_LOSS:
        raise NotImplementedError(f"clip_loss_type={clip_loss_type!r} (run_stage1.py:432-436 knows l2, mse, smooth_l1, l1)")
    out = model(videos, None, clip_only=True, vis_tokens=vis, n_vis=n_vis)          # (K, B, n_vis, C), L2-normalised rows
    return _PointwiseLoss.apply(out, targets, clip_loss_type)


def stage1_step(model, teacher_model, videos, n_source, mask_ratio, mask_type, bool_masked_pos, clip_loss_data, state: StepState,
                clip_input_resolution=224, importance=None, clip_loss_type='l2'):
    """teacher -> mask -> targets -> student loss (device tensors only).  Returns the 0-dim loss tensor (with grad_fn)."""
    tout = teacher_phase(teacher_model, videos, mask_ratio, mask_type, bool_masked_pos, state, clip_input_resolution, importance)
    return student_phase(model, videos, tout, n_source, clip_loss_data, clip_loss_type)


class AheadStream:
    """What TeacherAhead (stage 1) and MaskTeacherAhead (stage 3) share: a stream of their own for the frozen teacher, output slots that
    rotate, the ordering of the teacher's INPUTS, and the planner hints its GEMM launches carry (per call, inside unite_gemm_args: no
    process-wide setting is touched, include/unite_hip.h ABI 2)."""

    # planner weight when UNITE_GEMM_SHARING is not set.  Re-measured at the end of round 4 (same call, three interleaved runs each; the
    # kernels had changed under the 0.8 of round 2): stage 1 ViT-B 20.06 / 20.07 / 20.09 ms at 0.8, 20.06 / 20.05 / 20.08 at 0.85,
    # 19.96 / 19.91 / 19.94 at 0.9, 21.1 at 0.95 (another box: 20.74 / 20.71 / 20.67 -> 20.45 / 20.49 / 20.47); ViT-L 34.56 / 34.39 -> 33.87 / 33.80;
    # stage 3 prefers 0.8 (59.83 / 60.35 vs 61.61 / 60.90).  At 0.9 the weight gradients run in 2-4 split-K slices instead of 3-7 and the
    # proj input gradient and the decoder products on 256^2 tiles.
    DEFAULT_SHARING = 0.8

    def __init__(self, device, n_slots: int):
        self.stream = torch.cuda.Stream(device=device, priority=int(os.environ.get("UNITE_TEACHER_AHEAD_PRIO", "0")))
        self.n_streams = max(1, int(os.environ.get("UNITE_TEACHER_AHEAD_STREAMS", "1")))
        self.gemm_policy = int(os.environ.get("UNITE_TEACHER_PP", "0"))          # -1: whatever the process-wide policy is
        # both phases share the GPU: the GEMM planner weighs the CU time of a launch against its latency (include/unite_hip.h).  The weight
        # travels with every launch of the teacher phase (hints()) and of the student step (student())
        self.sharing = float(os.environ.get("UNITE_GEMM_SHARING", str(self.DEFAULT_SHARING)))
        self.n_slots = max(2, n_slots)
        self.n = 0
        self._marks = []                   # events on the student's stream, one per launch

    def hints(self):
        """context: GEMM launches of the teacher phase"""
        return ops.plan(persistent=self.gemm_policy if self.gemm_policy >= 0 else None, sharing=self.sharing)

    def student(self):
        """context: GEMM launches of the student step that runs beside the teacher phase of the next batch"""
        return ops.plan(sharing=self.sharing)

    def close(self):
        """kept for callers of round 2 (the planner weight was a process-wide setting then; now every launch carries its own)"""

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def next_slot(self, inputs_ready=None) -> int:
        """Order this launch on the teacher's stream and return its output slot.
        Slot reuse: launch k overwrites the outputs of launch k - n_slots, read by student step k - n_slots.  A mark recorded at launch j
        has every student step <= j - 2 in front of it (step j - 1 is enqueued right after launch j), so the mark of launch
        k - n_slots + 2 is late enough: the oldest of the n_slots - 1 marks kept.  With three slots the teacher may start on batch i+1
        while the student is still on batch i-1, and neither stream waits for the other at every step (two slots: the mark of this call).
        Inputs: ``inputs_ready`` None = unknown producer: the teacher's stream waits for everything enqueued so far on the caller's
        stream (a device batch produced right before this call must not be read early -- round-2 advisor finding); an event = wait for
        that; False = the caller vouches that the inputs are host memory, were produced on ``self.stream`` or are long complete."""
        slot = self.n % self.n_slots
        self.n += 1
        main = torch.cuda.current_stream()
        ev = torch.cuda.Event()
        ev.record(main)
        self._marks.append(ev)
        if len(self._marks) > self.n_slots - 1:
            self._marks.pop(0)
        self.stream.wait_event(self._marks[0])
        if inputs_ready is None:
            self.stream.wait_event(ev)
        elif inputs_ready is not False:
            self.stream.wait_event(inputs_ready)
        return slot


class TeacherAhead(AheadStream):
    """The frozen teacher one batch ahead of the student: ``launch(videos)`` enqueues the whole teacher phase of a batch on a stream of
    its own and returns its TeacherOut; the caller then trains the student on the PREVIOUS batch, whose teacher phase was launched an
    iteration earlier.  Nothing in the teacher depends on the student (frozen, no_grad: run_stage1.py:371), so the arithmetic of every
    step is that of stage1_step; what changes is that the student's many short, partially filled launches (and, on N GPUs, its gradient
    all-reduces) share the GPU with the teacher's long GEMMs.  Outputs rotate through ``n_slots`` (three) slots.

    Measured on MI355X (B = 32): 23.7 -> 21.6 ms per step, 20.4 with the GEMM planner told that launches share the
    GPU (plan_sharing: larger tiles, fewer split-K slices).  Beside a concurrent student the teacher is best left on ONE
    stream (its three frame-range streams: +0.4 ms) and on the tile GEMM kernels (the persistent kernel keeps every CU for a whole launch,
    so nothing of the student slips in between its tiles: +0.2 ms); UNITE_TEACHER_AHEAD_STREAMS / UNITE_TEACHER_PP change that."""

    DEFAULT_SHARING = 0.9                  # stage 1 (AheadStream.DEFAULT_SHARING has the measurements)

    def __init__(self, teacher_model, state: StepState, device, mask_ratio, mask_type, clip_input_resolution=224):
        super().__init__(device, int(os.environ.get("UNITE_TEACHER_AHEAD_SLOTS", "3")))
        self.teacher, self.state = teacher_model, state
        self.mask_ratio, self.mask_type, self.res = mask_ratio, mask_type, clip_input_resolution

    def launch(self, videos, bool_masked_pos=None, importance=None, inputs_ready=None) -> TeacherOut:
        """``videos`` may still be on the host: the copy then goes on the teacher's stream too (TeacherOut.videos is the device tensor).
        ``inputs_ready``: see AheadStream.next_slot (default: safe for a device batch produced on the current stream just now)."""
        slot = self.next_slot(inputs_ready)
        trt = getattr(self.teacher, "module", self.teacher).runtime()
        keep = trt.n_streams
        trt.n_streams = self.n_streams
        try:
            with torch.cuda.stream(self.stream), self.hints():
                if not videos.is_cuda:
                    videos = videos.to(self.stream.device, non_blocking=True)
                out = teacher_phase(self.teacher, videos, self.mask_ratio, self.mask_type, bool_masked_pos, self.state, self.res,
                                    importance, slot=slot, inline_targets=True)
                out.ready = torch.cuda.Event()
                out.ready.record(self.stream)
                out.foreign = True
                out.videos = videos
        finally:
            trt.n_streams = keep
        return out


class _Ahead:
    """iterates (batch, TeacherOut) with the teacher phase of the following batch already enqueued.  A batch is FETCHED with the
    teacher's stream current: whatever the loader produces on the device at that moment (synthetic clips, a device-side transform) is
    then ordered in front of the teacher phase by the stream itself, and the student reaches it behind TeacherOut.ready."""

    def __init__(self, loader, prepare, ahead: TeacherAhead, mask_type):
        self.loader, self.prepare, self.ahead, self.mask_type = loader, prepare, ahead, mask_type

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        prev = None
        it = iter(self.loader)
        while True:
            with torch.cuda.stream(self.ahead.stream):
                try:
                    batch = next(it)
                except StopIteration:
                    break
                cur = self.prepare(batch)
            tout = self.ahead.launch(cur[0], cur[1] if self.mask_type != 'attention' else None, inputs_ready=False)
            if prev is not None:
                yield prev
            prev = (cur, tout)
        if prev is not None:
            yield prev


def train_one_epoch(model: torch.nn.Module, data_loader: Iterable, data_loader_train_target: Optional[Iterable],
                    optimizer: torch.optim.Optimizer, device: torch.device, epoch: int, loss_scaler, max_norm: float = 0,
                    log_writer=None, lr_scheduler=None, start_steps=None, lr_schedule_values=None, wd_schedule_values=None,
                    src_classifier=None, teacher_model=None, clip_input_resolution=224, clip_loss_type='l2', clip_loss_ratio=0.5,
                    mask_type='tube', mask_ratio=0., use_wandb=False, args=None):
    if clip_loss_type != 'l2' and clip_loss_type not in ops.POINTWISE_LOSS:
        raise NotImplementedError(f"clip_loss_type={clip_loss_type!r} (run_stage1.py:432-436 knows l2, mse, smooth_l1, l1)")
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

    # teacher one batch ahead (default on a GPU; UNITE_TEACHER_AHEAD=0 or args.teacher_ahead=False restores the strictly sequential step)
    ahead_on = getattr(args, "teacher_ahead", None)
    if ahead_on is None:
        ahead_on = os.environ.get("UNITE_TEACHER_AHEAD", "1") != "0"
    ahead_on = bool(ahead_on) and torch.device(device).type == "cuda" and state.step_params is None

    def prepare(batch):
        """source (+ target) clips of one iteration (:344-358) -> (videos, bool_masked_pos, n_source)"""
        nonlocal target_iter
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
        if not ahead_on:
            videos = videos.to(device, non_blocking=True)
        return videos, bool_masked_pos, B_s          # teacher-ahead: TeacherAhead.launch copies the batch on the teacher's stream

    clip_loss_data = getattr(args, "clip_loss_data", "mixed")
    if ahead_on:
        ahead = TeacherAhead(teacher_model, state, device, mask_ratio, mask_type, clip_input_resolution)
        source = _Ahead(data_loader, prepare, ahead, mask_type)
    else:
        source = data_loader

    import contextlib
    ring = getattr(loss_scaler, "RING", 256) - 2      # grad-norm results live in a ring of device slots: read them back before it wraps
    # the student's GEMM launches carry the shared-GPU planner weight while the teacher runs ahead; left on any exit, also when the loop raises
    with (ahead.student() if ahead_on else contextlib.nullcontext()):
        for step, item in enumerate(metric_logger.log_every(source, print_freq, getattr(args, "epochs", None), epoch, ipe, header=header)):
            it = start_steps + step
            if lr_schedule_values is not None or wd_schedule_values is not None:
                for param_group in optimizer.param_groups:
                    if lr_schedule_values is not None:
                        param_group["lr"] = lr_schedule_values[min(it, len(lr_schedule_values) - 1)] * param_group["lr_scale"]
                    if wd_schedule_values is not None and param_group["weight_decay"] > 0:
                        param_group["weight_decay"] = wd_schedule_values[min(it, len(wd_schedule_values) - 1)]

            if ahead_on:
                (_, bool_masked_pos, B_s), tout = item
                videos = tout.videos
                loss = student_phase(model, videos, tout, B_s, clip_loss_data, clip_loss_type)
            else:
                videos, bool_masked_pos, B_s = prepare(item)
                loss = stage1_step(model, teacher_model, videos, B_s, mask_ratio, mask_type, bool_masked_pos, clip_loss_data, state,
                                   clip_input_resolution, clip_loss_type=clip_loss_type)
            optimizer.zero_grad()
            grad_norm = loss_scaler(loss, optimizer, clip_grad=max_norm, parameters=None, create_graph=False, reducer=reducer)
            pending.append((loss, grad_norm))

            if (print_freq and (step % print_freq == 0 or step == ipe - 1)) or len(pending) >= ring:
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
