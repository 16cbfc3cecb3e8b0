"""Stage-2 (source fine-tuning) training engine -- drop-in for ``train_one_epoch`` / ``train_class_batch`` of the
reference's src/engines/engine_for_finetuning.py:37-171: same signatures, gradient accumulation over ``update_freq``
micro-batches, same returned ``{meter: global_avg}``.  DeepSpeed (loss_scaler is None) and Mixup are outside the built path.
As in stage 1 the per-step host syncs of the reference (:98, :129) are dropped: loss / accuracy / grad-norm stay on the
device and are read back every ``print_freq`` steps, where the finite-loss exit (:100-102) also runs.
"""
from __future__ import annotations

import math
import sys
from typing import Iterable, Optional

import torch

from . import utils


def train_class_batch(model, samples, target, criterion):
    outputs = model(samples)
    loss = criterion(outputs, target)
    return loss, outputs


def train_one_epoch(model: torch.nn.Module, criterion: torch.nn.Module, data_loader: Iterable, optimizer: torch.optim.Optimizer,
                    device: torch.device, epoch: int, loss_scaler, max_norm: float = 0, model_ema=None, mixup_fn=None, log_writer=None,
                    start_steps=None, lr_schedule_values=None, wd_schedule_values=None, num_training_steps_per_epoch=None,
                    update_freq=None, num_epochs=None, train_head_only=False, wandb_run=None, args=None):
    if loss_scaler is None:
        raise NotImplementedError("the DeepSpeed branch (loss_scaler is None) is out of scope (enable_deepspeed: false in every config)")
    if mixup_fn is not None or model_ema is not None:
        raise NotImplementedError("Mixup / ModelEma are disabled by the UNITE stage-2 config (mixup: 0, cutmix: 0) and not built")
    model.train(True)
    update_freq = update_freq or 1
    metric_logger = utils.MetricLogger(delimiter="  ")
    metric_logger.add_meter('lr', utils.SmoothedValue(window_size=1, fmt='{value:.6f}'))
    metric_logger.add_meter('min_lr', utils.SmoothedValue(window_size=1, fmt='{value:.6f}'))
    header = 'Epoch: [{}]'.format(epoch)
    print_freq = 10
    ipe = len(data_loader)
    start_steps = start_steps or 0
    if num_training_steps_per_epoch is None:
        num_training_steps_per_epoch = ipe // update_freq
    reducer = getattr(model, "reducer", None)
    net = getattr(model, "module", model)
    fused = isinstance(criterion, torch.nn.CrossEntropyLoss) and criterion.label_smoothing == 0 and criterion.weight is None \
        and criterion.reduction == "mean" and hasattr(net, "forward_loss")
    optimizer.zero_grad()
    pending = []
    lr_now = [0.0, 0.0, None]          # max lr, min lr, weight decay of the iteration being logged

    def flush():
        if not pending:
            return
        vals = torch.stack([torch.stack([l.detach().float(), a.detach().float(), (g if g is not None else l * 0 - 1).detach().float()])
                            for l, a, g in pending]).tolist()
        pending.clear()
        for lv, av, gv in vals:
            if not math.isfinite(lv):
                print("Loss is {}, stopping training".format(lv))
                sys.exit(1)
            metric_logger.update(loss=lv, class_acc=av)
            if gv >= 0:
                metric_logger.update(grad_norm=gv)
            # engine_for_finetuning.py:134-167: the scalar logger and wandb see every iteration (here: when its scalars are read back)
            if log_writer is not None:
                log_writer.update(loss=lv, head="loss")
                log_writer.update(class_acc=av, head="loss")
                log_writer.update(loss_scale=loss_scaler.state_dict()["scale"], head="opt")
                log_writer.update(lr=lr_now[0], head="opt")
                log_writer.update(min_lr=lr_now[1], head="opt")
                log_writer.update(weight_decay=lr_now[2], head="opt")
                log_writer.update(grad_norm=gv if gv >= 0 else None, head="opt")
                log_writer.set_step()
            if wandb_run is not None:
                wandb_run.log({"train/loss": lv, "train/class_acc": av, "train/lr": lr_now[0], "train/min_lr": lr_now[1],
                               "train/weight_decay": lr_now[2], "train/grad_norm": gv if gv >= 0 else None})

    for data_iter_step, (samples, targets, _, _) in enumerate(metric_logger.log_every(data_loader, print_freq, num_epochs, epoch, ipe, header)):
        step = data_iter_step // update_freq
        if step >= num_training_steps_per_epoch:
            continue
        it = start_steps + step
        if lr_schedule_values is not None or wd_schedule_values is not None and data_iter_step % update_freq == 0:
            for param_group in optimizer.param_groups:
                if lr_schedule_values is not None:
                    param_group["lr"] = lr_schedule_values[it] * param_group["lr_scale"]
                if wd_schedule_values is not None and param_group["weight_decay"] > 0:
                    param_group["weight_decay"] = wd_schedule_values[it]
        samples = samples.to(device, non_blocking=True)
        targets = targets.to(device, non_blocking=True)
        if fused:
            loss, output = net.forward_loss(samples, targets)          # mean CE, run_stage2.py:681
        else:
            loss, output = train_class_batch(model, samples, targets, criterion)
        loss_log = loss
        loss = loss / update_freq
        grad_norm = loss_scaler(loss, optimizer, clip_grad=max_norm, parameters=None, create_graph=False,
                                update_grad=(data_iter_step + 1) % update_freq == 0, reducer=reducer)
        if (data_iter_step + 1) % update_freq == 0:
            optimizer.zero_grad()
        class_acc = (output.detach().max(-1)[-1] == targets).float().mean()
        pending.append((loss_log, class_acc, grad_norm))
        min_lr, max_lr = 10., 0.
        for group in optimizer.param_groups:
            min_lr, max_lr = min(min_lr, group["lr"]), max(max_lr, group["lr"])
        weight_decay_value = None
        for group in optimizer.param_groups:
            if group["weight_decay"] > 0:
                weight_decay_value = group["weight_decay"]
        lr_now[:] = [max_lr, min_lr, weight_decay_value]
        if data_iter_step % print_freq == 0 or data_iter_step == ipe - 1:
            flush()
        metric_logger.update(lr=max_lr, min_lr=min_lr, weight_decay=weight_decay_value, loss_scale=loss_scaler.state_dict()["scale"])
    flush()
    metric_logger.synchronize_between_processes()
    print("Averaged stats:", metric_logger)
    return {k: meter.global_avg for k, meter in metric_logger.meters.items()}


# ----------------------------------------------------------------------------- evaluation (engine_for_finetuning.py:174-352)
def accuracy(output, target, topk=(1,)):
    """timm.utils.accuracy (timm 0.4.12, the reference's import at engine_for_finetuning.py:28): top-k hit rates in percent."""
    maxk = min(max(topk), output.size(1))
    batch_size = target.size(0)
    _, pred = output.topk(maxk, 1, True, True)
    pred = pred.t()
    correct = pred.eq(target.reshape(1, -1).expand_as(pred))
    return [correct[:min(k, maxk)].reshape(-1).float().sum(0) * 100. / batch_size for k in topk]


def compute_ece(softmaxes, labels, n_bins: int = 15):
    """Expected calibration error.  The reference imports it from ``src.knn`` (engine_for_finetuning.py:35), a file that is NOT in
    the repository: parity unpinned.  This is the standard definition (Guo et al. 2017): 15 equal-width confidence bins,
    sum_b |acc_b - conf_b| * n_b / n."""
    conf, pred = softmaxes.max(dim=1)
    hit = pred.eq(labels).float()
    edges = torch.linspace(0, 1, n_bins + 1)
    ece = torch.zeros(())
    for lo, hi in zip(edges[:-1], edges[1:]):
        in_bin = (conf > lo) & (conf <= hi)
        if in_bin.any():
            ece = ece + (hit[in_bin].mean() - conf[in_bin].mean()).abs() * in_bin.float().mean()
    return float(ece)


def _gather_all(t):
    if not utils.is_dist_avail_and_initialized():
        return t
    import torch.distributed as dist
    parts = [torch.zeros_like(t) for _ in range(utils.get_world_size())]
    dist.barrier()
    dist.all_gather(parts, t)
    return torch.cat(parts)


class _EvalPass:
    """One pass of a classifier over a loader without gradients: what ``validation_one_epoch`` (engine_for_finetuning.py:174-232) and
    ``final_test`` (:235-296) have in common -- logits per batch, cross-entropy / top-1 / top-5 meters weighted by batch size, the soft-max
    rows and labels kept for the calibration error -- with a hook per batch for what differs (the per-view lines of final_test)."""

    def __init__(self, model, device, header):
        self.model, self.device, self.header = model, device, header
        self.criterion = torch.nn.CrossEntropyLoss()
        self.meters = utils.MetricLogger(delimiter="  ")
        self.probs, self.labels = [], []
        self.last_acc = (torch.zeros(()), torch.zeros(()))

    def run(self, data_loader, per_batch=None):
        self.model.eval()
        for batch in self.meters.log_every(data_loader, 10, 1, 0, len(data_loader), header=self.header):
            videos = batch[0].to(self.device, non_blocking=True)
            target = batch[1].to(self.device, non_blocking=True)
            logits = self.model(videos)
            n = videos.shape[0]
            self.last_acc = accuracy(logits, target, topk=(1, 5))
            self.meters.update(loss=self.criterion(logits, target).item())
            self.meters.meters['acc1'].update(self.last_acc[0].item(), n=n)
            self.meters.meters['acc5'].update(self.last_acc[1].item(), n=n)
            self.probs.append(torch.softmax(logits, dim=1))
            self.labels.append(target)
            if per_batch is not None:
                per_batch(batch, logits, target)
        return self

    def calibration(self, gather: bool):
        probs, labels = torch.cat(self.probs), torch.cat(self.labels)
        if gather:
            probs, labels = _gather_all(probs), _gather_all(labels)
        ece = compute_ece(probs.cpu(), labels.cpu())
        print(f"Expected Calibration Error (ECE): {ece:.4f}")
        return ece, probs, labels

    def summary(self):
        self.meters.synchronize_between_processes()
        m = self.meters
        print('* Acc@1 {top1.global_avg:.3f} Acc@5 {top5.global_avg:.3f} loss {losses.global_avg:.3f}'.format(top1=m.acc1, top5=m.acc5, losses=m.loss))
        return {k: meter.global_avg for k, meter in m.meters.items()}


@torch.no_grad()
def validation_one_epoch(data_loader, model, device, fp32=False, save_preds_path=None):
    """engine_for_finetuning.py:174-232: loss / top-1 / top-5 over the loader, ECE over the soft-max outputs gathered from all ranks;
    optionally the predictions and labels as .npy files."""
    import os
    import numpy as np
    ev = _EvalPass(model, device, 'Val:').run(data_loader)
    ece, probs, labels = ev.calibration(gather=True)
    if save_preds_path is not None:
        os.makedirs(save_preds_path, exist_ok=True)
        np.save(os.path.join(save_preds_path, 'preds.npy'), torch.argmax(probs, dim=1).cpu().numpy())
        np.save(os.path.join(save_preds_path, 'labels.npy'), labels.cpu().numpy())
        print(f"Saved predictions to {save_preds_path}")
    return ev.summary(), ece


def view_line(video_id, logits_row, label, chunk, split) -> str:
    """one line of a rank's result file: '<id> [l0, l1, ...] <label> <chunk> <split>' (the format ``merge`` parses; :270-275)"""
    return "{} {} {} {} {}\n".format(video_id, str(logits_row.numpy().tolist()), str(int(label)), str(int(chunk)), str(int(split)))


@torch.no_grad()
def final_test(data_loader, model, device, file):
    """engine_for_finetuning.py:235-296: every (video, temporal chunk, spatial crop) view's logits go to ``file`` for ``merge``; its first
    line carries the LAST batch's acc1, acc5 (as the reference writes it)."""
    lines = []

    def keep_views(batch, logits, target):
        ids, chunk_nb, split_nb = batch[2], batch[3], batch[4]
        rows, tgt = logits.cpu(), target.cpu()
        lines.extend(view_line(ids[i], rows[i], tgt[i], chunk_nb[i], split_nb[i]) for i in range(rows.size(0)))

    ev = _EvalPass(model, device, 'Test:').run(data_loader, per_batch=keep_views)
    ece, _, _ = ev.calibration(gather=False)
    with open(file, 'w') as f:
        f.write("{}, {}\n".format(*ev.last_acc))
        f.writelines(lines)
    return ev.summary(), ece


def compute_video(lst):
    """engine_for_finetuning.py:344-352 (``[index, video_id, per-view soft-max vectors, label]``): the video's score is the mean over
    its views -> [prediction, top-1 hit, top-5 hit, label]."""
    import numpy as np
    _, _, views, label = lst
    score = np.mean(np.stack(list(views)), axis=0)
    ranked = np.argsort(-score)
    label = int(label)
    return [ranked[0], float(ranked[0] == label), float(label in ranked[:5]), label]


class ViewResults:
    """The views of every video collected from the ranks' result files (``merge``): a view is identified by its temporal chunk and
    spatial crop, and a view that a second rank wrote again (the distributed sampler pads the last batch) counts once."""

    def __init__(self):
        self.probs, self.seen, self.label = {}, {}, {}

    def add_line(self, line: str) -> None:
        import numpy as np
        from scipy.special import softmax
        line = line.strip()
        name, _, rest = line.rpartition('[')                     # the id may contain spaces or brackets: split at the LAST '['
        vector, _, tail = rest.rpartition(']')
        label, chunk, split = tail.split(' ')[1:4]
        views = self.probs.setdefault(name, [])
        seen = self.seen.setdefault(name, set())
        self.label.setdefault(name, 0)
        key = chunk + split                                      # the reference's key: the two strings joined
        if key in seen:
            return
        seen.add(key)
        views.append(softmax(np.array([float(v) for v in vector.split(',')], dtype=np.float64)))
        self.label[name] = label

    def read(self, path: str) -> "ViewResults":
        with open(path) as f:
            next(f)                                              # first line: the writer's last-batch accuracies
            for line in f:
                self.add_line(line)
        return self

    def videos(self):
        return [compute_video([i, name, views, self.label[name]]) for i, (name, views) in enumerate(self.probs.items())]


def merge(eval_path, num_tasks):
    """engine_for_finetuning.py:299-342: union of the ranks' '<rank>.txt' files, soft-max per view, mean over the views of a video ->
    top-1 / top-5 in percent.  (np.float of the reference is gone from numpy >= 1.24: float64 here; no process pool.)"""
    import os
    import numpy as np
    results = ViewResults()
    print("Reading individual output files")
    for rank in range(num_tasks):
        results.read(os.path.join(eval_path, f"{rank}.txt"))
    print("Computing final results")
    per_video = results.videos()
    return np.mean([v[1] for v in per_video]) * 100, np.mean([v[2] for v in per_video]) * 100
