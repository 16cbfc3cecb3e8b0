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
        if data_iter_step % print_freq == 0 or data_iter_step == ipe - 1:
            flush()
        min_lr, max_lr = 10., 0.
        for group in optimizer.param_groups:
            min_lr, max_lr = min(min_lr, group["lr"]), max(max_lr, group["lr"])
        weight_decay_value = None
        for group in optimizer.param_groups:
            if group["weight_decay"] > 0:
                weight_decay_value = group["weight_decay"]
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


@torch.no_grad()
def validation_one_epoch(data_loader, model, device, fp32=False, save_preds_path=None):
    """engine_for_finetuning.py:174-232: loss / top-1 / top-5 over the loader, ECE over the gathered soft-max outputs."""
    import os
    import numpy as np
    criterion = torch.nn.CrossEntropyLoss()
    metric_logger = utils.MetricLogger(delimiter="  ")
    header = 'Val:'
    ipe = len(data_loader)
    model.eval()
    softmaxes, labels = [], []
    for batch in metric_logger.log_every(data_loader, 10, 1, 0, ipe, header=header):
        videos = batch[0].to(device, non_blocking=True)
        target = batch[1].to(device, non_blocking=True)
        output = model(videos)
        loss = criterion(output, target)
        acc1, acc5 = accuracy(output, target, topk=(1, 5))
        batch_size = videos.shape[0]
        metric_logger.update(loss=loss.item())
        metric_logger.meters['acc1'].update(acc1.item(), n=batch_size)
        metric_logger.meters['acc5'].update(acc5.item(), n=batch_size)
        softmaxes.append(torch.softmax(output, dim=1))
        labels.append(target)
    softmaxes = _gather_all(torch.cat(softmaxes))
    labels = _gather_all(torch.cat(labels))
    ece = compute_ece(softmaxes.cpu(), labels.cpu())
    print(f"Expected Calibration Error (ECE): {ece:.4f}")
    if save_preds_path is not None:
        os.makedirs(save_preds_path, exist_ok=True)
        np.save(os.path.join(save_preds_path, 'preds.npy'), torch.argmax(softmaxes, dim=1).cpu().numpy())
        np.save(os.path.join(save_preds_path, 'labels.npy'), labels.cpu().numpy())
        print(f"Saved predictions to {save_preds_path}")
    metric_logger.synchronize_between_processes()
    print('* Acc@1 {top1.global_avg:.3f} Acc@5 {top5.global_avg:.3f} loss {losses.global_avg:.3f}'
          .format(top1=metric_logger.acc1, top5=metric_logger.acc5, losses=metric_logger.loss))
    return {k: meter.global_avg for k, meter in metric_logger.meters.items()}, ece


@torch.no_grad()
def final_test(data_loader, model, device, file):
    """engine_for_finetuning.py:235-296: per-view logits written as '<id> [logits] <label> <chunk> <split>' lines (first line: the
    last batch's acc1, acc5, as in the reference) for ``merge``."""
    criterion = torch.nn.CrossEntropyLoss()
    metric_logger = utils.MetricLogger(delimiter="  ")
    header = 'Test:'
    model.eval()
    final_result, softmaxes, labels = [], [], []
    acc1 = acc5 = torch.zeros(())
    for batch in metric_logger.log_every(data_loader, 10, 1, 0, len(data_loader), header):
        videos, target, ids, chunk_nb, split_nb = batch[0], batch[1], batch[2], batch[3], batch[4]
        videos = videos.to(device, non_blocking=True)
        target = target.to(device, non_blocking=True)
        output = model(videos)
        loss = criterion(output, target)
        out_cpu, tgt_cpu = output.cpu(), target.cpu()
        for i in range(output.size(0)):
            final_result.append("{} {} {} {} {}\n".format(ids[i], str(out_cpu[i].numpy().tolist()), str(int(tgt_cpu[i])),
                                                          str(int(chunk_nb[i])), str(int(split_nb[i]))))
        acc1, acc5 = accuracy(output, target, topk=(1, 5))
        batch_size = videos.shape[0]
        metric_logger.update(loss=loss.item())
        metric_logger.meters['acc1'].update(acc1.item(), n=batch_size)
        metric_logger.meters['acc5'].update(acc5.item(), n=batch_size)
        softmaxes.append(torch.softmax(output, dim=1).cpu())
        labels.append(tgt_cpu)
    ece = compute_ece(torch.cat(softmaxes), torch.cat(labels))
    print(f"Expected Calibration Error (ECE): {ece:.4f}")
    with open(file, 'w') as f:
        f.write("{}, {}\n".format(acc1, acc5))
        for line in final_result:
            f.write(line)
    metric_logger.synchronize_between_processes()
    print('* Acc@1 {top1.global_avg:.3f} Acc@5 {top5.global_avg:.3f} loss {losses.global_avg:.3f}'
          .format(top1=metric_logger.acc1, top5=metric_logger.acc5, losses=metric_logger.loss))
    return {k: meter.global_avg for k, meter in metric_logger.meters.items()}, ece


def compute_video(lst):
    """engine_for_finetuning.py:344-352: mean of the per-view soft-max vectors -> prediction, top-1 / top-5 hits."""
    import numpy as np
    i, video_id, data, label = lst
    feat = np.mean([x for x in data], axis=0)
    pred = np.argmax(feat)
    top1 = (int(pred) == int(label)) * 1.0
    top5 = (int(label) in np.argsort(-feat)[:5]) * 1.0
    return [pred, top1, top5, int(label)]


def merge(eval_path, num_tasks):
    """engine_for_finetuning.py:299-342: union of the ranks' '<rank>.txt' files, duplicate (chunk, split) views dropped, soft-max per
    view, mean over views per video.  (np.float of the reference is gone from numpy >= 1.24: float64 here; no process pool.)"""
    import os
    import numpy as np
    from scipy.special import softmax
    dict_feats, dict_label, dict_pos = {}, {}, {}
    print("Reading individual output files")
    for x in range(num_tasks):
        file = os.path.join(eval_path, str(x) + '.txt')
        for line in open(file, 'r').readlines()[1:]:
            line = line.strip()
            name = line.rsplit('[', maxsplit=1)[0]
            tail = line.rsplit(']', maxsplit=1)[1].split(' ')
            label, chunk_nb, split_nb = tail[1], tail[2], tail[3]
            data = np.array([float(v) for v in line.rsplit('[', maxsplit=1)[1].rsplit(']', maxsplit=1)[0].split(',')], dtype=np.float64)
            data = softmax(data)
            if name not in dict_feats:
                dict_feats[name], dict_label[name], dict_pos[name] = [], 0, []
            if chunk_nb + split_nb in dict_pos[name]:
                continue
            dict_feats[name].append(data)
            dict_pos[name].append(chunk_nb + split_nb)
            dict_label[name] = label
    print("Computing final results")
    ans = [compute_video([i, item, dict_feats[item], dict_label[item]]) for i, item in enumerate(dict_feats)]
    top1, top5 = [x[1] for x in ans], [x[2] for x in ans]
    return np.mean(top1) * 100, np.mean(top5) * 100
