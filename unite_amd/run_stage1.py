"""Stage-1 driver: ``torchrun --nproc_per_node=N -m unite_amd.run_stage1 --config configs/stage1_config.yaml [--synthetic]``
(reference: stage1.sh:15-17 -> run_stage1.py:604-908 ``main``).  Same sequence as the reference's main(): distributed init, seeds,
student + frozen CLIP teacher, linear lr scaling by the global batch (:798-800), DistributedDataParallel, create_optimizer with the
layer-decay assigner, cosine lr / weight-decay schedules, auto-resume, ``train_one_epoch`` per epoch, checkpoints and log.txt.

What is NOT here is the reference's dataset stack (decord / PIL workers, SURVEY.md 8f-1: next row): without ``--synthetic`` the
driver stops with that message.  ``--synthetic`` feeds seeded random clips of the configured shape, ``--synthetic_steps`` per epoch,
which is how the path is exercised offline (no datasets, no CLIP weights: the teacher is random-init unless
``--clip_teacher_weights`` / UNITE_CLIP_PATH names a checkpoint, Appendix A-3)."""
from __future__ import annotations

import datetime
import json
import os
import time

import numpy as np
import torch
import yaml

from . import cli, clip, utils
from .checkpoint import load_student_from_ckpt
from .engine_stage1 import train_one_epoch
from .optim_factory import LayerDecayValueAssigner, create_optimizer
from .registry import create_model
from .utils import NativeScalerWithGradNormCount as NativeScaler


class SyntheticClips:
    """An epoch of ``steps`` batches shaped like the reference's pre-training loader output (videos (B,3,T,H,W) f32 standing for
    ImageNet-normalised pixels, bool_masked_pos placeholder, labels): seeded per rank and epoch, generated on the device."""

    def __init__(self, steps, batch_size, num_frames, size, nb_classes, device, seed):
        self.steps, self.shape, self.nb, self.device, self.seed = steps, (batch_size, 3, num_frames, size, size), nb_classes, device, seed
        self.sampler = self
        self.epoch = 0

    def set_epoch(self, epoch):
        self.epoch = epoch

    def __len__(self):
        return self.steps

    def __iter__(self):
        g = torch.Generator(device=self.device).manual_seed(self.seed + 7919 * self.epoch)
        for _ in range(self.steps):
            videos = torch.randn(self.shape, generator=g, device=self.device)
            labels = torch.randint(0, self.nb, (self.shape[0],), generator=g, device=self.device)
            yield videos, torch.full((self.shape[0],), -1), labels


def get_model(args):
    """run_stage1.py:273-291"""
    print(f"Creating model: {args.model}")
    return create_model(
        args.model, pretrained=False, drop_path_rate=args.drop_path, drop_block_rate=None, use_learnable_pos_emb=args.use_learnable_pos_emb,
        use_checkpoint=args.use_checkpoint, checkpoint_num=args.checkpoint_num, clip_decoder_embed_dim=args.clip_decoder_embed_dim,
        clip_output_dim=args.clip_output_dim, clip_norm_type=args.clip_norm_type, num_frames=args.num_frames, tubelet_size=args.tubelet_size,
        clip_return_layers=args.clip_return_layers, clip_student_return_interval=args.clip_student_return_interval, use_cls_token=args.use_cls_token)


def real_loaders(args, device):
    """run_stage1.py:654-745: source (and target) training clips through unite_amd.datasets -- frame numbers, crop boxes, flips and masks are
    drawn in the workers like the reference draws them, the pixels are cropped / resized / normalised on the GPU.  Source and target are
    repeated so that both loaders have the same number of steps.  (The reference also builds validation / test sets here, :658-659, which
    stage 1 only hands to the missing src/knn.py: not built.)"""
    from .data import DistributedSampler
    from .datasets import build_pretraining_dataset, DeviceLoader
    args.window_size = (args.num_frames // args.tubelet_size, args.input_size // 16, args.input_size // 16)       # (:769, before the model exists)
    world, rank = utils.get_world_size(), utils.get_rank()

    def loader(dataset, repetitions):
        sampler = DistributedSampler(dataset, num_replicas=world, rank=rank, shuffle=True, repetitions=repetitions)
        kw = dict(persistent_workers=True) if args.num_workers > 0 else {}
        return DeviceLoader(dataset, args.batch_size, device, sampler=sampler, num_workers=args.num_workers, drop_last=True,
                            worker_init_fn=utils.seed_worker, **kw)

    dataset_train = build_pretraining_dataset(args, args.ann_file_train, fraction=args.train_fraction)
    train_rep, target_loader = args.train_repetitions, None
    if args.ann_file_train_target:
        dataset_target = build_pretraining_dataset(args, args.ann_file_train_target)
        if len(dataset_target) < len(dataset_train):
            target_rep = int(np.ceil(len(dataset_train) / len(dataset_target)))
            print("Repeating target dataset %d times" % target_rep)
        else:
            target_rep, train_rep = 1, int(np.ceil(len(dataset_target) / len(dataset_train)))
            print("Repeating source dataset %d times" % train_rep)
        target_loader = loader(dataset_target, target_rep)
    return loader(dataset_train, train_rep), target_loader


def main(args):
    utils.init_distributed_mode(args)
    device = torch.device(args.device)
    seed = args.seed + utils.get_rank()                                    # :613
    torch.manual_seed(seed)
    np.random.seed(seed)
    if utils.is_main_process() and args.output_dir:
        os.makedirs(args.output_dir, exist_ok=True)
        with open(os.path.join(args.output_dir, "config.yaml"), "w") as f:
            yaml.dump(vars(args), f, default_flow_style=False)
    if args.synthetic:
        data_loader_train = SyntheticClips(args.synthetic_steps, args.batch_size, args.num_frames, args.input_size, args.nb_classes, device, seed)
        data_loader_train_target = None                                    # a second domain would double the batch (:796)
    else:
        data_loader_train, data_loader_train_target = real_loaders(args, device)
    num_training_steps_per_epoch = len(data_loader_train)

    model = get_model(args)
    if args.student_init:
        model = load_student_from_ckpt(args, model)
        print("Loaded student model!")
    patch_size = model.encoder.patch_embed.patch_size
    args.window_size = (args.num_frames // args.tubelet_size, args.input_size // patch_size[0], args.input_size // patch_size[1])
    args.patch_size = patch_size
    model.to(device)
    model_without_ddp = model
    n_parameters = utils.count_parameters(model)
    print('Student Params: {} M'.format(n_parameters / 1e6))

    weights = args.clip_teacher_weights or os.environ.get("UNITE_CLIP_PATH", "")
    if weights:
        os.environ["UNITE_CLIP_PATH"] = weights
    teacher_model = getattr(clip, args.clip_teacher)(                      # the reference resolves the name with eval(), :782
        pretrained=bool(weights), clip_norm_type=args.clip_norm_type, input_resolution=args.clip_input_resolution,
        return_attn=args.clip_return_attn or args.mask_type == 'attention', clip_return_layers=args.clip_return_layers,
        clip_return_interval=args.clip_return_interval)
    teacher_model.to(device)
    print(f'Teacher model: {args.clip_teacher}')

    total_batch_size = args.batch_size * utils.get_world_size() * (2 if data_loader_train_target is not None else 1)
    scale = total_batch_size * args.num_sample / 256                      # :798-800
    args.lr, args.min_lr, args.warmup_lr = args.lr * scale, args.min_lr * scale, args.warmup_lr * scale
    print("LR = %.8f" % args.lr)
    print("Batch size = %d" % total_batch_size)
    print("Number of training steps per epoch = %d" % num_training_steps_per_epoch)

    num_layers = 12                                                        # hard-coded in the reference (:807)
    if args.distributed:
        from .ddp import DistributedDataParallel
        model = DistributedDataParallel(model, device_ids=[args.gpu], find_unused_parameters=False)     # the frozen teacher stays a plain replica (A-18)
        model_without_ddp = model.module
    assigner = LayerDecayValueAssigner([args.layer_decay ** (num_layers + 1 - i) for i in range(num_layers + 2)]) if args.layer_decay < 1.0 else None
    optimizer = create_optimizer(args, model_without_ddp, skip_list=model_without_ddp.no_weight_decay(),
                                 get_num_layer=assigner.get_layer_id if assigner is not None else None,
                                 get_layer_scale=assigner.get_scale if assigner is not None else None)
    loss_scaler = NativeScaler()
    lr_schedule_values = utils.cosine_scheduler(args.lr, args.min_lr, args.epochs, num_training_steps_per_epoch,
                                                warmup_epochs=args.warmup_epochs, warmup_steps=args.warmup_steps)
    if args.weight_decay_end is None:
        args.weight_decay_end = args.weight_decay
    wd_schedule_values = utils.cosine_scheduler(args.weight_decay, args.weight_decay_end, args.epochs, num_training_steps_per_epoch)
    if args.auto_resume and args.output_dir:
        utils.auto_load_model(args=args, model=model, model_without_ddp=model_without_ddp, optimizer=optimizer, loss_scaler=loss_scaler)
    print(f"Start training for {args.epochs} epochs")
    start_time = time.time()
    train_stats = {}
    for epoch in range(args.start_epoch, args.epochs):
        data_loader_train.sampler.set_epoch(epoch)
        train_stats = train_one_epoch(
            model, data_loader_train, data_loader_train_target, optimizer, device, epoch, loss_scaler, args.clip_grad,
            start_steps=epoch * num_training_steps_per_epoch, lr_schedule_values=lr_schedule_values, wd_schedule_values=wd_schedule_values,
            src_classifier=None, teacher_model=teacher_model, clip_input_resolution=args.clip_input_resolution,
            clip_loss_type=args.clip_loss_type, clip_loss_ratio=args.clip_loss_ratio, mask_type=args.mask_type, mask_ratio=args.mask_ratio,
            use_wandb=False, args=args)
        if args.output_dir and args.checkpoints_enabled and utils.is_main_process():
            if (epoch + 1) % args.save_ckpt_freq == 0 or epoch + 1 == args.epochs:
                utils.save_model(args=args, model=model, model_without_ddp=model_without_ddp, optimizer=optimizer, loss_scaler=loss_scaler, epoch=epoch)
            utils.save_latest_model(args=args, model=model, model_without_ddp=model_without_ddp, optimizer=optimizer, loss_scaler=loss_scaler, epoch=epoch)
        if args.output_dir and utils.is_main_process():
            with open(os.path.join(args.output_dir, "log.txt"), mode="a", encoding="utf-8") as f:
                f.write(json.dumps({**{f'train_{k}': v for k, v in train_stats.items()}, 'epoch': epoch, 'n_parameters': n_parameters}) + "\n")
    print('Training time {}'.format(str(datetime.timedelta(seconds=int(time.time() - start_time)))))
    if utils.is_dist_avail_and_initialized():
        torch.distributed.destroy_process_group()
    return train_stats


if __name__ == '__main__':
    main(cli.get_args("stage1"))
