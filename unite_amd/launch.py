"""What the three drivers (run_stage1 / run_stage2 / run_stage3) share around their ``main``: run set-up (distributed init, per-rank
seed, output directory + config dump: run_stage1.py:609-628 and the same lines of the other two), the seeded synthetic loaders that stand in
for the reference's decord / PIL dataset stack, an in-memory scalar logger with the TensorboardLogger interface, and the per-epoch
checkpoint / log.txt tail (run_stage1.py:880-898)."""
from __future__ import annotations

import json
import os

import numpy as np
import torch
import yaml

from . import utils


def start_run(args):
    """-> (device, per-rank seed)"""
    utils.init_distributed_mode(args)
    device = torch.device(args.device)
    seed = args.seed + utils.get_rank()
    torch.manual_seed(seed)
    np.random.seed(seed)
    if utils.is_main_process() and args.output_dir:
        os.makedirs(args.output_dir, exist_ok=True)
        with open(os.path.join(args.output_dir, "config.yaml"), "w") as f:
            yaml.dump(vars(args), f, default_flow_style=False)
    return device, seed


def require_synthetic(args, what):
    if not getattr(args, "synthetic", False):
        raise NotImplementedError(f"no input: run with --synthetic, give annotation lists (ann_file_train, ...: unite_amd/datasets_cls.py), or hand "
                                  f"your own loaders to {what}")


def cls_loaders(args, device, num_tasks, global_rank, batch_sizes, *, with_val=True, dist_eval=True, train_repetitions=1, target_annotation=None):
    """The stage-2 / stage-3 loaders over real video lists (run_stage2.py:492-563, run_stage3.py:1042-1145): ``build_dataset`` for the train /
    validation / test modes (+ stage 3's target-domain list in validation mode with its augmented second view), the reference's samplers --
    its own ``DistributedSampler`` with repetitions for training, torch's un-shuffled one for evaluation -- and ``DeviceLoader``s: workers decode
    and draw, the GPU does the pixel arithmetic (datasets_cls.py).  ``batch_sizes`` = (train, validation, test).  Returns a dict of loaders."""
    import numpy as np
    from . import utils
    from .data import DistributedSampler
    from .datasets import DeviceLoader
    from .datasets_cls import build_dataset
    kw = dict(persistent_workers=True) if args.num_workers > 0 else {}

    def loader(ds, bs, sampler, drop_last, seeded=False):
        return DeviceLoader(ds, bs, device, sampler=sampler, num_workers=args.num_workers, drop_last=drop_last,
                            worker_init_fn=utils.seed_worker if seeded else None, **kw)

    def eval_sampler(ds):
        if dist_eval:
            return torch.utils.data.DistributedSampler(ds, num_replicas=num_tasks, rank=global_rank, shuffle=False)
        return torch.utils.data.SequentialSampler(ds)

    dataset_train, args.nb_classes = build_dataset(is_train=True, test_mode=False, args=args)
    dataset_val = build_dataset(is_train=False, test_mode=False, args=args)[0] if with_val else None
    dataset_test = build_dataset(is_train=False, test_mode=True, args=args)[0]
    out = {}
    if target_annotation:
        dataset_target, _ = build_dataset(is_train=False, test_mode=False, args=args, annotation_file=target_annotation)
        if len(dataset_target) < len(dataset_train):
            target_rep = int(np.ceil(len(dataset_train) / len(dataset_target)))
            print("Repeating target dataset %d times" % target_rep)
        else:
            target_rep = 1
            train_repetitions = train_repetitions if train_repetitions > 0 else int(np.ceil(len(dataset_target) / len(dataset_train)))
            print("Repeating source dataset %d times" % train_repetitions)
        out["target"] = loader(dataset_target, batch_sizes[0], DistributedSampler(dataset_target, num_replicas=num_tasks, rank=global_rank,
                                                                                  shuffle=True, repetitions=target_rep), True, seeded=True)
    sampler_train = DistributedSampler(dataset_train, num_replicas=num_tasks, rank=global_rank, shuffle=True, repetitions=max(1, train_repetitions))
    print("Sampler_train = %s" % str(sampler_train))
    out["train"] = loader(dataset_train, batch_sizes[0], sampler_train, True, seeded=True)
    out["val"] = loader(dataset_val, batch_sizes[1], eval_sampler(dataset_val), False) if dataset_val is not None else None
    out["test"] = loader(dataset_test, batch_sizes[2], eval_sampler(dataset_test), False)
    return out


class SyntheticLoader:
    """``steps`` batches per epoch, generated on the device from a per-rank, per-epoch seed.  ``make(generator, batch_size)`` returns
    the batch tuple in the layout of the loader it replaces."""

    def __init__(self, steps, batch_size, device, seed, make):
        self.steps, self.batch_size, self.device, self.seed, self.make = steps, batch_size, device, seed, make
        self.sampler = self                    # the drivers call loader.sampler.set_epoch(epoch)
        self.dataset = range(steps * batch_size)
        self.epoch = 0

    def set_epoch(self, epoch):
        self.epoch = epoch

    def __len__(self):
        return self.steps

    def __iter__(self):
        g = torch.Generator(device=self.device).manual_seed(self.seed + 7919 * self.epoch)
        for _ in range(self.steps):
            yield self.make(g, self.batch_size)


def clips(g, B, T, size, device):
    return torch.randn((B, 3, T, size, size), generator=g, device=device)       # stands for ImageNet-normalised pixels


def labels(g, B, nb_classes, device):
    return torch.randint(0, nb_classes, (B,), generator=g, device=device)


class ScalarLog:
    """TensorboardLogger's interface (set_step / update(head=..., step=..., **scalars) / flush) keeping the scalars in memory and, if
    given a directory, appending them to ``scalars.jsonl`` on flush: the drivers log through it when tensorboardX / torch's
    tensorboard writer is not installed (the reference requires tensorboardX)."""

    def __init__(self, log_dir=None):
        self.step, self.rows, self.log_dir = 0, [], log_dir

    def set_step(self, step=None):
        self.step = step if step is not None else self.step + 1

    def update(self, head='scalar', step=None, **kwargs):
        for k, v in kwargs.items():
            if v is None:
                continue
            if isinstance(v, torch.Tensor):
                v = v.item()
            self.rows.append((self.step if step is None else step, f"{head}/{k}", float(v)))

    def flush(self):
        if self.log_dir:
            with open(os.path.join(self.log_dir, "scalars.jsonl"), "a") as f:
                for step, key, v in self.rows:
                    f.write(json.dumps({"step": step, "key": key, "value": v}) + "\n")
            self.rows = []


def make_log_writer(args):
    """rank 0 with an output directory logs scalars (run_stage2.py:514-518): tensorboard if a writer is installed, ScalarLog otherwise"""
    if not (utils.get_rank() == 0 and args.output_dir):
        return None
    try:
        return utils.TensorboardLogger(log_dir=args.output_dir)
    except ImportError:
        return ScalarLog(args.output_dir)


def end_of_epoch(args, epoch, model, model_without_ddp, optimizer, loss_scaler, stats: dict, n_parameters, save_enabled, log_writer=None):
    """checkpoint-{epoch}.pth every save_ckpt_freq epochs and at the end, checkpoint-latest.pth every epoch, one JSON line in log.txt"""
    if args.output_dir and save_enabled and utils.is_main_process():
        if (epoch + 1) % args.save_ckpt_freq == 0 or epoch + 1 == args.epochs:
            utils.save_model(args=args, model=model, model_without_ddp=model_without_ddp, optimizer=optimizer, loss_scaler=loss_scaler, epoch=epoch)
        utils.save_latest_model(args=args, model=model, model_without_ddp=model_without_ddp, optimizer=optimizer, loss_scaler=loss_scaler, epoch=epoch)
    if args.output_dir and utils.is_main_process():
        if log_writer is not None:
            log_writer.flush()
        with open(os.path.join(args.output_dir, "log.txt"), mode="a", encoding="utf-8") as f:
            f.write(json.dumps({**stats, 'epoch': epoch, 'n_parameters': n_parameters}) + "\n")


def layer_decay_assigner(layer_decay, num_layers):
    from .optim_factory import LayerDecayValueAssigner
    if layer_decay >= 1.0:
        return None
    a = LayerDecayValueAssigner([layer_decay ** (num_layers + 1 - i) for i in range(num_layers + 2)])
    print("Assigned values = %s" % str(a.values))
    return a
