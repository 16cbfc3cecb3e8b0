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
