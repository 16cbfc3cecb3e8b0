"""Stage-3 (collaborative self-training) engine -- drop-in for ``train_one_epoch`` / ``pool_outputs`` of the reference's
run_stage3.py:333-710.

Per step (reference lines in brackets):
  teacher CLS attention on the AUGMENTED target clips                               [:434-451]  teacher.forward_attention
  student, source clips, all tokens -> mean-pool -> src_classifier -> CE            [:475-477,486]
  student, target clips, all tokens, no grad -> max-softmax-prob + prediction       [:480-490]
  k = 2 committee: greedy attention-rank masks, 320 visible tokens each             [:493-506]  unite_greedy_masks
  selection (conf / cons / consORconf / consANDconf / clip_only / clip_matchORconf / oracle) [:508-593]  unite_pseudo_label_select
  target loss = ratio * sel_ratio * mean(msp * CE(last member's logits, pseudo-label)) [:599-613]
  loss = class_loss_src_ratio_pl * CE_s + loss_t -> backward -> grad-norm -> AdamW  [:625-644]
Kept quirks: ``src_classifier`` is used but never optimised; only the LAST committee member's
logits enter the loss (the first member's pass therefore runs without saving activations -- samples are independent, its
gradient in the reference's batched pass is exactly zero); the pseudo-label is always the student's own prediction;
``global_threshold`` is 0.5.  The zero-shot CLIP probabilities of the clip_* strategies come from ``clip_probs_fn(videos_t) -> (B_t, C)``:
``functools.partial(utils.clip_infer, clip_model, text_features=...)`` runs the CLIP image tower on the HIP kernels (unite_amd.clip
VisionTransformer.encode_image + unite_clip_similarity); the class text embeddings are the caller's (tokenizer / text tower: out of scope).
"""
from __future__ import annotations

import math
import os
import sys
import time
from typing import Callable, Iterable, Optional

import torch

from . import ops, utils
from .engine_stage1 import AheadStream, teacher_input

F32 = torch.float32


def pool_outputs(outputs, use_cls_token):
    """run_stage3.py:333-338 (mean over tokens; the CLS-token variant is not built)."""
    if use_cls_token:
        raise NotImplementedError("use_cls_token is not built")
    B, N, D = outputs.shape
    out = torch.empty(B, D, dtype=F32, device=outputs.device)
    return ops.token_mean_fwd(outputs.contiguous(), out)


def committee_masks(teacher_model, videos_t_aug, masking_type, mask_ratio, clip_input_resolution, N, ws, slot=0):
    """k = 2 greedy attention-rank masks of the augmented target clips (run_stage3.py:434-506) -> (cmask u8 [k, B*T, N], cvis i32 [k, B*T*n_vis])."""
    B_t, T = videos_t_aug.shape[0], videos_t_aug.shape[2]
    if masking_type == "clip_attention":
        attn_t = teacher_model.forward_attention(teacher_input(teacher_model, videos_t_aug, clip_input_resolution))   # (B_t*T, N)
    elif masking_type == "random":
        attn_t = torch.rand(B_t * T, N, device=videos_t_aug.device)                # run_stage3.py:455
    else:
        raise NotImplementedError(masking_type)
    k = 2
    n_vis_frame = N - int(N * mask_ratio)
    sfx = "" if slot == 0 else f".{slot}"
    cmask = ws.get("s3.cmask" + sfx, (k, B_t * T, N), torch.uint8)
    cvis = ws.get("s3.cvis" + sfx, (k, B_t * T * n_vis_frame), torch.int32)
    ops.greedy_masks(attn_t, k, cmask, cvis, n_vis_frame)
    return cmask, cvis


class MaskOut:
    __slots__ = ("cmask", "cvis", "ready", "videos_t_aug")


class MaskTeacherAhead(AheadStream):
    """Stage 3's frozen mask teacher one batch ahead of the student, on a stream of its own (the stage-1 scheme, engine_stage1.TeacherAhead):
    ``launch(videos_t_aug)`` enqueues the CLIP forward + greedy masks of a batch; the caller then runs the student passes of the PREVIOUS
    batch.  The teacher's attention depends on nothing the student changes (run_stage3.py:434-451: no_grad, frozen), so every step computes
    what stage3_step computes.  Three output slots; the GEMM launches of both sides carry the shared-GPU planner weight (hints() /
    student()).  ``clip_probs_fn`` may run the SAME CLIP tower on the student's stream (utils.clip_infer(teacher_model, ...)): the teacher
    runtime orders its uses across streams itself (clip._TeacherRuntime._enter / _leave)."""

    def __init__(self, teacher_model, student, device, mask_ratio, masking_type, clip_input_resolution):
        super().__init__(device, 3)
        self.teacher, self.ws = teacher_model, getattr(student, "module", student).runtime().ws
        self.N = getattr(student, "module", student).runtime().frame_tokens
        self.mask_ratio, self.masking_type, self.res = mask_ratio, masking_type, clip_input_resolution

    def launch(self, videos_t_aug, inputs_ready=None) -> MaskOut:
        """``inputs_ready``: see AheadStream.next_slot (default: the teacher's stream waits for everything enqueued so far on the caller's)"""
        slot = self.next_slot(inputs_ready)
        trt = getattr(self.teacher, "module", self.teacher).runtime() if self.masking_type == "clip_attention" else None
        keep = trt.n_streams if trt is not None else None
        out = MaskOut()
        try:
            if trt is not None:
                trt.n_streams = self.n_streams
            with torch.cuda.stream(self.stream), self.hints():
                if not videos_t_aug.is_cuda:
                    videos_t_aug = videos_t_aug.to(self.stream.device, non_blocking=True)
                out.cmask, out.cvis = committee_masks(self.teacher, videos_t_aug, self.masking_type, self.mask_ratio, self.res, self.N, self.ws, slot)
                out.ready = torch.cuda.Event()
                out.ready.record(self.stream)
                out.videos_t_aug = videos_t_aug
        finally:
            if trt is not None:
                trt.n_streams = keep
        return out


class _Stage3LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, teacher_model, src_classifier, videos_s, labels_s, videos_t, videos_t_aug, labels_t, cfg, anchor):
        ops.keep_plan(ctx)
        student = getattr(model, "module", model)
        rt = student.runtime()
        dev = videos_s.device
        B_s, B_t, T = videos_s.shape[0], videos_t.shape[0], videos_s.shape[2]
        N, D = rt.frame_tokens, rt.D
        W, bcls = src_classifier.weight.detach().float().contiguous(), src_classifier.bias.detach().float().contiguous()
        C = W.shape[0]
        training = student.training
        ws = rt.ws

        def classify(xv, B, n, tag):
            pooled = ws.get(f"s3.pool.{tag}", (B, D), F32)
            ops.token_mean_fwd(xv.view(B, n, D), pooled)
            logits = ws.get(f"s3.logits.{tag}", (B, C), F32)
            ops.linear_f32_fwd(pooled, W, bcls, logits)
            return pooled, logits

        # masks for the committee from the teacher's CLS attention on the augmented target clips: computed here, or already under way
        # on the mask teacher's own stream (MaskTeacherAhead)
        k = 2
        n_vis_frame = N - int(N * cfg["mask_ratio"])
        n_vis = n_vis_frame * T
        masks = cfg.get("masks")
        if masks is None:
            cmask, cvis = committee_masks(teacher_model, videos_t_aug, cfg["masking_type"], cfg["mask_ratio"], cfg["clip_input_resolution"], N, ws)
        else:
            torch.cuda.current_stream().wait_event(masks.ready)
            cmask, cvis = masks.cmask, masks.cvis
        # student passes
        xv_s = rt.encode(videos_s, None, T * N, "s3src", training, save=True)
        _, logits_s = classify(xv_s, B_s, T * N, "src")
        xv_t = rt.encode(videos_t, None, T * N, "s3tgt", training, save=False)
        _, logits_full_t = classify(xv_t, B_t, T * N, "tgt")
        logits_masked = ws.get("s3.logits.masked", (k, B_t, C), F32)
        for i in range(k):
            last = i == k - 1
            xv_m = rt.encode(videos_t_aug, cvis[i], n_vis, f"s3cm{i}", training, save=last)
            pooled = ws.get(f"s3.pool.cm{i}", (B_t, D), F32)
            ops.token_mean_fwd(xv_m.view(B_t, n_vis, D), pooled)
            ops.linear_f32_fwd(pooled, W, bcls, logits_masked[i])
        # selection + losses
        clip_probs = cfg["clip_probs_fn"](videos_t) if cfg["selection_strategy"] in ("clip_only", "clip_matchORconf") else None
        pseudo = ws.get("s3.pseudo", (B_t,), torch.int64)
        weight = ws.get("s3.weight", (B_t,), F32)
        sel = ws.get("s3.sel", (B_t,), torch.uint8)
        msp = ws.get("s3.msp", (B_t,), F32)
        ops.pseudo_label_select(logits_full_t, logits_masked, cfg["selection_strategy"], 0.5, cfg["clip_threshold"],
                                cfg["conf_weighted_loss"], pseudo, weight, clip_probs=clip_probs, labels_t=labels_t, sel=sel, msp=msp)
        sums = ws.get("s3.sums", (2,), F32)
        sums.zero_()
        dlog_s = ws.get("s3.dlog.s", (B_s, C), F32)
        dlog_t = ws.get("s3.dlog.t", (B_t, C), F32)
        src_scale = cfg["class_loss_src_ratio_pl"] / B_s
        tgt_scale = cfg["class_loss_tgt_ratio"] / B_t
        ops.softmax_ce(logits_s, labels_s, sums[0:1], dlog_s, grad_scale=src_scale)
        ce_in = logits_masked[k - 1] if cfg["train_masked"] else logits_full_t
        if cfg["full_oracle"]:
            ops.softmax_ce(ce_in, labels_t, sums[1:2], dlog_t, grad_scale=1.0 / B_t)
            tgt_scale = 1.0 / B_t
        else:
            ops.softmax_ce(ce_in, pseudo, sums[1:2], dlog_t, row_weight=weight, grad_scale=tgt_scale)
        if not cfg["train_masked"]:
            raise NotImplementedError("train_masked=False (loss on the no-grad full-clip logits) trains nothing but the frozen classifier")
        ctx.model, ctx.W, ctx.dims = model, W, (B_s, B_t, T * N, n_vis, D, k)
        loss_s, loss_t = sums[0] / B_s, sums[1] * tgt_scale
        ctx.mark_non_differentiable(loss_s, loss_t, sel)
        return cfg["class_loss_src_ratio_pl"] * loss_s + loss_t, loss_s, loss_t, sel

    @staticmethod
    def backward(ctx, gloss, *_):
        with ops.kept_plan(ctx):
            student = getattr(ctx.model, "module", ctx.model)
            rt = student.runtime()
            ws = rt.ws
            B_s, B_t, n_full, n_vis, D, k = ctx.dims
            fp = rt.fp
            if not fp.accumulate:       # clip_decoder.* gets no gradient in stage 3 (its outputs are discarded, run_stage3.py:475): keep
                (lo, hi), = fp.layer_ranges(["clip_decoder."])      # the flat buffer's slice at zero for the grad-norm and the all-reduce
                fp.grad[lo:hi].zero_()
            for pi, (tag, slot, B, n, dl) in enumerate((("src", "s3src", B_s, n_full, ws.peek("s3.dlog.s")), (f"cm{k - 1}", f"s3cm{k - 1}", B_t, n_vis, ws.peek("s3.dlog.t")))):
                dpool = ws.get(f"s3.dpool.{tag}", (B, D), F32)
                ops.linear_f32_bwd(ws.peek(f"s3.pool.{tag}"), ctx.W, dl * gloss, dx=dpool)
                dxv = ws.get(f"s3.dxv.{tag}", (B, n, D), F32)
                ops.token_mean_bwd(dpool, dxv)
                rt.encode_backward(dxv.view(B * n, D), slot, notify=pi == 1)
            return (None,) * 10


def stage3_step(model, teacher_model, src_classifier, videos_s, labels_s, videos_t, videos_t_aug, labels_t, args, mask_ratio,
                clip_probs_fn: Optional[Callable] = None, clip_input_resolution: Optional[int] = None, masks: Optional[MaskOut] = None):
    """``masks``: the committee masks of this batch if a MaskTeacherAhead already launched them (videos_t_aug is then masks.videos_t_aug)."""
    student = getattr(model, "module", model)
    if masks is not None:
        videos_t_aug = masks.videos_t_aug
        videos_t_aug.record_stream(torch.cuda.current_stream())
    cfg = dict(clip_input_resolution=clip_input_resolution or videos_t_aug.shape[-1],
               masking_type=getattr(args, "masking_type", "clip_attention"), mask_ratio=mask_ratio,
               selection_strategy=getattr(args, "selection_strategy", "clip_matchORconf"), clip_threshold=float(getattr(args, "clip_threshold", 0.5)),
               conf_weighted_loss=bool(getattr(args, "conf_weighted_loss", True)), class_loss_tgt_ratio=float(getattr(args, "class_loss_tgt_ratio", 1.0)),
               class_loss_src_ratio_pl=float(getattr(args, "class_loss_src_ratio_pl", 1.0)), train_masked=bool(getattr(args, "train_masked", True)),
               full_oracle=bool(getattr(args, "full_oracle", False)), clip_probs_fn=clip_probs_fn, masks=masks)
    if cfg["selection_strategy"] in ("clip_only", "clip_matchORconf") and clip_probs_fn is None:
        raise NotImplementedError("selection_strategy '%s' needs zero-shot CLIP probabilities: pass clip_probs_fn (the OpenAI CLIP "
                                  "text/image towers of utils.setup_clip are outside the built path)" % cfg["selection_strategy"])
    return _Stage3LossFn.apply(model, teacher_model, src_classifier, videos_s, labels_s, videos_t, videos_t_aug, labels_t, cfg,
                               student.runtime().grad_anchor)


def train_one_epoch(model: torch.nn.Module, data_loader: Iterable, data_loader_train_target: Iterable, optimizer: torch.optim.Optimizer,
                    device: torch.device, epoch: int, loss_scaler, max_norm: float = 0, log_writer=None, lr_scheduler=None,
                    start_steps=None, lr_schedule_values=None, wd_schedule_values=None, src_classifier=None, teacher_model=None,
                    clip_input_resolution=224, clip_loss_type='l2', clip_loss_ratio=0.5, mask_type='tube', mask_ratio=0.,
                    use_wandb=False, args=None, classwise_thresholds=None, global_threshold=None, clip_probs_fn=None):
    model.train()
    if hasattr(optimizer, "set_unused"):
        optimizer.set_unused(("clip_decoder.",))        # p.grad stays None for them in the reference: AdamW skips them entirely
    if args.class_loss_src_ratio <= 0 or src_classifier is None:
        raise NotImplementedError("stage 3 without a source classifier (class_loss_src_ratio <= 0) has no loss in the reference either")
    if data_loader_train_target is None:
        raise ValueError("stage 3 needs the target loader")
    src_classifier.train()
    metric_logger = utils.MetricLogger(delimiter="  ")
    metric_logger.add_meter('lr', utils.SmoothedValue(window_size=1, fmt='{value:.6f}'))
    metric_logger.add_meter('min_lr', utils.SmoothedValue(window_size=1, fmt='{value:.6f}'))
    header = 'Epoch [{}]:'.format(epoch)
    ipe = len(data_loader)
    print_freq = args.log_freq
    start_steps = start_steps or 0
    reducer = getattr(model, "reducer", None)
    tgt_iter = iter(data_loader_train_target)
    pending = []

    def flush():
        if not pending:
            return
        vals = torch.stack([torch.stack([x.detach().float().reshape(()) for x in row]) for row in pending]).tolist()
        pending.clear()
        for lv, ls, lt, sr, gn in vals:
            if not math.isfinite(lv):
                print("Loss is {}, stopping training".format(lv))
                sys.exit(1)
            metric_logger.update(loss=lv, loss_class=ls, loss_class_t=lt, select_ratio=sr, grad_norm=gn)

    def prepare(batch):
        """one iteration's source + target batch (:396-431) -> (videos_s, labels_s, videos_t, videos_t_aug, labels_t), still where the loaders put them"""
        nonlocal tgt_iter
        videos_s, labels_s = batch[0], batch[1]
        try:
            tb = next(tgt_iter)
        except StopIteration:
            tgt_iter = iter(data_loader_train_target)
            tb = next(tgt_iter)
        videos_t = tb[0]
        if getattr(args, "return_aug_for_val", True):
            videos_t_aug, labels_t = tb[1], tb[2]
        else:
            videos_t_aug, labels_t = tb[0], tb[1]          # the reference would fail on cat(None) here (:413); use the clip itself
        return videos_s, labels_s, videos_t, videos_t_aug, labels_t

    # mask teacher one batch ahead (default on a GPU; args.teacher_ahead=False / UNITE_TEACHER_AHEAD=0: the order of run_stage3.py)
    ahead_on = getattr(args, "teacher_ahead", None)
    if ahead_on is None:
        ahead_on = os.environ.get("UNITE_TEACHER_AHEAD", "1") != "0"
    ahead_on = bool(ahead_on) and torch.device(device).type == "cuda"
    masking_type = getattr(args, "masking_type", "clip_attention")
    ahead = MaskTeacherAhead(teacher_model, model, device, mask_ratio, masking_type, clip_input_resolution) if ahead_on else None

    class _Ahead:
        def __len__(self):
            return len(data_loader)

        def __iter__(self):
            prev = None
            for batch in data_loader:
                cur = prepare(batch)
                m = ahead.launch(cur[3])
                if prev is not None:
                    yield prev
                prev = (cur, m)
            if prev is not None:
                yield prev

    source = _Ahead() if ahead_on else data_loader
    import contextlib
    ring = getattr(loss_scaler, "RING", 256) - 2      # grad-norm results live in a ring of device slots: read them back before it wraps
    with (ahead.student() if ahead is not None else contextlib.nullcontext()):
        for step, item in enumerate(metric_logger.log_every(source, print_freq, getattr(args, "epochs", None), epoch, ipe, header=header)):
            it = start_steps + step
            if lr_schedule_values is not None or wd_schedule_values is not None:
                for param_group in optimizer.param_groups:
                    if lr_schedule_values is not None:
                        param_group["lr"] = lr_schedule_values[min(it, len(lr_schedule_values) - 1)] * param_group["lr_scale"]
                    if wd_schedule_values is not None and param_group["weight_decay"] > 0:
                        param_group["weight_decay"] = wd_schedule_values[min(it, len(wd_schedule_values) - 1)]
            (videos_s, labels_s, videos_t, videos_t_aug, labels_t), masks = item if ahead_on else (prepare(item), None)
            videos_s, videos_t = videos_s.to(device, non_blocking=True), videos_t.to(device, non_blocking=True)
            if masks is None:
                videos_t_aug = videos_t_aug.to(device, non_blocking=True)
            labels_s, labels_t = labels_s.to(device, non_blocking=True), labels_t.to(device, non_blocking=True)

            loss, loss_s, loss_t, sel = stage3_step(model, teacher_model, src_classifier, videos_s, labels_s, videos_t, videos_t_aug, labels_t,
                                                    args, mask_ratio, clip_probs_fn, clip_input_resolution, masks=masks)
            optimizer.zero_grad()
            grad_norm = loss_scaler(loss, optimizer, clip_grad=max_norm, parameters=None, create_graph=False, reducer=reducer)
            pending.append((loss, loss_s, loss_t, sel.float().mean(), grad_norm))
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


class _EncoderThenClassifier:
    """what stage 3 evaluates: the student encoder on ALL tokens (an all-visible mask, run_stage3.py:744-749 / :951-957), mean over the tokens
    (``pool_outputs`` without a CLS token), the frozen source classifier -> (B, nb_classes) f32 logits; no activations are kept"""

    def __init__(self, encoder, src_classifier, args):
        if getattr(args, "use_cls_token", False):
            raise NotImplementedError("use_cls_token is not built")
        self.encoder, self.src_classifier = encoder, src_classifier
        self.rt = getattr(encoder, "module", encoder).runtime()
        self.W = src_classifier.weight.detach().float().contiguous()
        self.b = src_classifier.bias.detach().float().contiguous()

    def eval(self):
        self.encoder.eval()
        self.src_classifier.eval()
        return self

    def __call__(self, videos):
        rt = self.rt
        B, T = videos.shape[0], videos.shape[2]
        n = T * rt.frame_tokens
        xv = rt.encode(videos, None, n, "s3val", False, save=False)          # encoder.norm(x) for every token
        pooled = torch.empty(B, rt.D, dtype=F32, device=videos.device)
        ops.token_mean_fwd(xv.view(B, n, rt.D), pooled)
        logits = torch.empty(B, self.W.shape[0], dtype=F32, device=videos.device)
        return ops.linear_f32_fwd(pooled, self.W, self.b, logits)


@torch.no_grad()
def validation_one_epoch(data_loader, encoder, src_classifier, device, fp32=False, args=None, use_wandb=False, save_preds_path=None):
    """run_stage3.py:714-787: the student encoder on ALL tokens, mean-pooled, through the source classifier; loss / top-1 / top-5."""
    from .engine_for_finetuning import accuracy
    criterion = torch.nn.CrossEntropyLoss()
    metric_logger = utils.MetricLogger(delimiter="  ")
    header = 'Val:'
    net = _EncoderThenClassifier(encoder, src_classifier, args).eval()
    for batch in metric_logger.log_every(data_loader, 10, 1, 0, len(data_loader), header):
        videos = batch[0]
        target = batch[2] if getattr(args, "return_aug_for_val", False) else batch[1]
        videos = videos.to(device, non_blocking=True)
        target = target.to(device, non_blocking=True)
        B = videos.shape[0]
        class_logits = net(videos)
        loss = criterion(class_logits, target)
        acc1, acc5 = accuracy(class_logits, target, topk=(1, 5))
        metric_logger.update(loss=loss.item())
        metric_logger.meters['acc1'].update(acc1.item(), n=B)
        metric_logger.meters['acc5'].update(acc5.item(), n=B)
    metric_logger.synchronize_between_processes()
    print('* Acc@1 {top1.global_avg:.3f} Acc@5 {top5.global_avg:.3f} loss {losses.global_avg:.3f}'
          .format(top1=metric_logger.acc1, top5=metric_logger.acc5, losses=metric_logger.loss))
    return {k: meter.global_avg for k, meter in metric_logger.meters.items()}


@torch.no_grad()
def final_test(data_loader, model, src_classifier, device, file, args):
    """run_stage3.py:927-989: every (video, temporal chunk, spatial crop) view of the test set through encoder -> pool_outputs -> src_classifier;
    the views' logits go to ``file`` in the line format ``engine_for_finetuning.merge`` parses (first line: the LAST batch's acc1, acc5, as the
    reference writes it), ECE over this rank's soft-max outputs, loss / top-1 / top-5 meters synchronised over the ranks."""
    from .engine_for_finetuning import _EvalPass, view_line
    lines = []

    def keep_views(batch, logits, target):
        ids, chunk_nb, split_nb = batch[2], batch[3], batch[4]
        rows, tgt = logits.cpu(), target.cpu()
        lines.extend(view_line(ids[i], rows[i], tgt[i], chunk_nb[i], split_nb[i]) for i in range(rows.size(0)))

    ev = _EvalPass(_EncoderThenClassifier(model, src_classifier, args), device, 'Test:').run(data_loader, per_batch=keep_views)
    ev.calibration(gather=False)
    with open(file, 'w') as f:
        f.write("{}, {}\n".format(*ev.last_acc))
        f.writelines(lines)
    return ev.summary()
