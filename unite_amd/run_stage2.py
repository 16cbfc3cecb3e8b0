"""Stage-2 driver (source fine-tuning of the video classifier): ``torchrun --nproc_per_node=N -m unite_amd.run_stage2 --config
configs/stage2_config.yaml [--synthetic]`` -- reference stage2.sh -> run_stage2.py:455-848 ``main``.

Sequence of the reference's main(): run set-up, loaders, ``get_model`` (:326-347) and ``load_from_ckpt`` (--finetune), DistributedDataParallel,
layer-wise lr decay from ``model.get_num_layers()`` (live here: stage-2 parameter names start with ``blocks.``), create_optimizer, the lr
schedule the config names (cosine / constant / step) and the cosine weight-decay schedule, auto-reload, the trainable-parameter
configuration (:711-746: --train_head_only, --frozen_layers [+ --freeze_patch_embedding], --lp_ft_epochs: blocks 0-8 + patch embedding frozen
for the first epochs, everything trainable from then on), per epoch ``train_one_epoch`` with ``update_freq`` gradient accumulation and the
scalar logger, validation every ``eval_freq`` epochs (best checkpoint kept), then ``final_test`` + ``merge``.
Input: --synthetic feeds seeded clips / labels; with ``ann_file_train / ann_file_val / ann_file_test`` lists the loaders come from
unite_amd/datasets_cls.py (``VideoClsDataset_sparse``: draws in the workers, pixel arithmetic on the GPU).  Not here: DeepSpeed, Mixup, ModelEma (off in the shipped config; the engine
refuses them)."""
from __future__ import annotations

import datetime
import os
import time

import torch

from . import cli, launch, utils
from .checkpoint import load_from_ckpt
from .engine_for_finetuning import final_test, merge, train_one_epoch, validation_one_epoch
from .optim_factory import create_optimizer
from .registry import create_model
from .utils import NativeScalerWithGradNormCount as NativeScaler


def get_model(args):
    """run_stage2.py:326-347"""
    print(f"Creating model: {args.model}")
    return create_model(
        args.model, pretrained=False, num_classes=args.nb_classes, all_frames=args.num_frames * args.num_segments, tubelet_size=args.tubelet_size,
        use_learnable_pos_emb=args.use_learnable_pos_emb, fc_drop_rate=args.fc_drop_rate, drop_rate=args.drop, drop_path_rate=args.drop_path,
        attn_drop_rate=args.attn_drop_rate, drop_block_rate=None, use_checkpoint=args.use_checkpoint, checkpoint_num=args.checkpoint_num,
        use_mean_pooling=args.use_mean_pooling, init_scale=args.init_scale, classifier_type=args.head_type,
        classifier_hidden_dim=args.head_hidden_dim)


def set_trainable(model, frozen_substrings):
    """requires_grad by name: what run_stage2.py's ``freeze_params`` does (:715-727); an empty list makes everything trainable"""
    frozen, trainable = [], []
    for name, param in model.named_parameters():
        param.requires_grad = not any(s in name for s in frozen_substrings)
        (trainable if param.requires_grad else frozen).append(name)
    print("Trainable parameters:\n{}".format(trainable))
    print("Frozen parameters:\n{}".format(frozen))
    return model


def configure_trainable(args, model):
    """run_stage2.py:729-746"""
    if args.train_head_only:
        for name, param in model.named_parameters():
            param.requires_grad = "head" in name or "norm.weight" in name or "norm.bias" in name
            if param.requires_grad:
                print("Training {}".format(name))
    elif args.frozen_layers:
        names = ['blocks.%d.' % int(n) for n in str(args.frozen_layers).split(",")]
        if args.freeze_patch_embedding:
            names.append('patch_embed')
        set_trainable(model, names)
    if args.lp_ft_epochs > 0:      # linear-probe-then-fine-tune: the lower blocks and the patch embedding wait for epoch lp_ft_epochs
        set_trainable(model, ['blocks.%d.' % n for n in range(9)] + ['patch_embed'])


def lr_schedule(args, steps_per_epoch):
    common = dict(warmup_epochs=args.warmup_epochs, start_warmup_value=args.warmup_lr, warmup_steps=args.warmup_steps)
    if args.lr_schedule == 'cosine':
        return utils.cosine_scheduler(args.lr, args.min_lr, args.epochs, steps_per_epoch, **common)
    if args.lr_schedule == 'constant':
        return utils.step_scheduler(args.lr, args.step_fraction, args.epochs, steps_per_epoch, **common)
    if args.lr_schedule == 'step':
        assert args.lr_step_epochs is not None
        return utils.step_scheduler(args.lr, args.step_fraction, args.epochs, steps_per_epoch, steps=args.lr_step_epochs, **common)
    raise ValueError(f"lr_schedule {args.lr_schedule!r}")


def main(args, ds_init=None):
    if ds_init is not None or getattr(args, "enable_deepspeed", False):
        raise NotImplementedError("DeepSpeed is out of scope (enable_deepspeed: false in the shipped config)")
    device, seed = launch.start_run(args)
    T, size, nb = args.num_frames * args.num_segments, args.input_size, args.nb_classes
    num_tasks, global_rank = utils.get_world_size(), utils.get_rank()
    if args.synthetic:
        def train_batch(g, B):       # (samples, targets, ids, extra) as the training dataset yields them (engine_for_finetuning.py:70)
            return launch.clips(g, B, T, size, device), launch.labels(g, B, nb, device), None, None

        def eval_batch(g, B):        # (videos, label, id, chunk, split)
            ids = [f"video_{global_rank}_{int(torch.randint(0, 1 << 30, (1,), generator=g, device=device))}" for _ in range(B)]
            return launch.clips(g, B, T, size, device), launch.labels(g, B, nb, device), ids, [0] * B, [0] * B

        steps_per_rank = args.synthetic_steps * args.update_freq
        data_loader_train = launch.SyntheticLoader(steps_per_rank, args.batch_size, device, seed, train_batch)
        data_loader_val = None if args.disable_eval_during_finetuning else launch.SyntheticLoader(2, 2 * args.batch_size, device, seed + 1, eval_batch)
        data_loader_test = launch.SyntheticLoader(2, 2 * args.batch_size, device, seed + 2, eval_batch)
    elif getattr(args, "ann_file_train", None):
        # real video lists (run_stage2.py:492-563): validation batches of 2 B, test batches of 4 B, evaluation spread over the ranks
        ld = launch.cls_loaders(args, device, num_tasks, global_rank, (args.batch_size, int(2 * args.batch_size), int(4 * args.batch_size)),
                                with_val=not args.disable_eval_during_finetuning, dist_eval=bool(getattr(args, "dist_eval", True)),
                                train_repetitions=getattr(args, "train_repetitions", 1))
        data_loader_train, data_loader_val, data_loader_test = ld["train"], ld["val"], ld["test"]
    else:
        launch.require_synthetic(args, "unite_amd.engine_for_finetuning.train_one_epoch")
    log_writer = launch.make_log_writer(args)

    model = get_model(args)
    patch_size = model.patch_embed.patch_size
    args.window_size = (args.num_frames // args.tubelet_size, args.input_size // patch_size[0], args.input_size // patch_size[1])
    args.patch_size = patch_size
    if args.finetune:
        model = load_from_ckpt(args, model)
    model.to(device)
    model_without_ddp = model
    n_parameters = sum(p.numel() for p in model.parameters() if p.requires_grad)
    print('number of params:', n_parameters)
    total_batch_size = args.batch_size * args.update_freq * num_tasks
    num_training_steps_per_epoch = len(data_loader_train) // args.update_freq
    print("LR = %.8f" % args.lr)                                             # no batch-size scaling in stage 2 (:603-605)
    print("Batch size = %d" % total_batch_size)
    print("Update frequent = %d" % args.update_freq)
    print("Number of training steps per epoch = %d" % num_training_steps_per_epoch)

    assigner = launch.layer_decay_assigner(args.layer_decay, model_without_ddp.get_num_layers())
    skip = model.no_weight_decay()
    if args.distributed:
        from .ddp import DistributedDataParallel
        model = DistributedDataParallel(model, device_ids=[args.gpu], find_unused_parameters=False)
        model_without_ddp = model.module
    optimizer = create_optimizer(args, model_without_ddp, skip_list=skip,
                                 get_num_layer=assigner.get_layer_id if assigner is not None else None,
                                 get_layer_scale=assigner.get_scale if assigner is not None else None)
    loss_scaler = NativeScaler()
    lr_schedule_values = lr_schedule(args, num_training_steps_per_epoch)
    if args.weight_decay_end is None:
        args.weight_decay_end = args.weight_decay
    wd_schedule_values = utils.cosine_scheduler(args.weight_decay, args.weight_decay_end, args.epochs, num_training_steps_per_epoch)
    print("Max WD = %.7f, Min WD = %.7f" % (max(wd_schedule_values), min(wd_schedule_values)))
    criterion = torch.nn.CrossEntropyLoss(label_smoothing=args.smoothing) if args.smoothing > 0. else torch.nn.CrossEntropyLoss()
    print("criterion = %s" % str(criterion))

    def test_and_merge():
        preds_file = os.path.join(args.output_dir, str(global_rank) + '.txt')
        final_test(data_loader_test, model, device, preds_file)
        if utils.is_dist_avail_and_initialized():
            torch.distributed.barrier()
        if global_rank == 0:
            print("Start merging results...")
            top1, top5 = merge(args.output_dir, num_tasks)
            print(f"Accuracy of the network on the test videos: Top-1: {top1:.2f}%, Top-5: {top5:.2f}%")
            return {'Final top-1': top1, 'Final Top-5': top5}
        return {}

    if args.eval:
        return test_and_merge()
    if args.auto_reload:
        utils.auto_load_model(args=args, model=model, model_without_ddp=model_without_ddp, optimizer=optimizer, loss_scaler=loss_scaler)
    configure_trainable(args, model_without_ddp)

    print(f"Start training for {args.epochs} epochs")
    start_time = time.time()
    max_accuracy, train_stats = 0.0, {}
    for epoch in range(args.start_epoch, args.epochs):
        data_loader_train.sampler.set_epoch(epoch)
        if log_writer is not None:
            log_writer.set_step(epoch * num_training_steps_per_epoch * args.update_freq)
        if args.lp_ft_epochs > 0 and epoch == args.lp_ft_epochs:
            set_trainable(model_without_ddp, [])
        train_stats = train_one_epoch(
            model, criterion, data_loader_train, optimizer, device, epoch, loss_scaler, args.clip_grad, None, None, log_writer=log_writer,
            start_steps=epoch * num_training_steps_per_epoch, num_epochs=args.epochs, lr_schedule_values=lr_schedule_values,
            wd_schedule_values=wd_schedule_values, num_training_steps_per_epoch=num_training_steps_per_epoch, update_freq=args.update_freq,
            train_head_only=args.train_head_only, wandb_run=None, args=args)
        stats = {f'train_{k}': v for k, v in train_stats.items()}
        if data_loader_val is not None and (epoch + 1) % args.eval_freq == 0:
            test_stats, ece = validation_one_epoch(data_loader_val, model, device)
            print(f"[{time.strftime('%Y-%m-%d %H:%M:%S', time.localtime())}] Accuracy of the network on the val videos: {test_stats['acc1']:.1f}%")
            if max_accuracy < test_stats["acc1"]:
                max_accuracy = test_stats["acc1"]
                if args.output_dir and args.save_ckpt and utils.is_main_process():
                    utils.save_latest_model(args=args, model=model, model_without_ddp=model_without_ddp, optimizer=optimizer,
                                            loss_scaler=loss_scaler, epoch=epoch, model_name='best')
            print(f'Max accuracy: {max_accuracy:.2f}%')
            if log_writer is not None:
                log_writer.update(val_acc1=test_stats['acc1'], head="perf", step=epoch)
                log_writer.update(val_acc5=test_stats['acc5'], head="perf", step=epoch)
                log_writer.update(val_loss=test_stats['loss'], head="perf", step=epoch)
            stats.update({f'val_{k}': v for k, v in test_stats.items()})
        launch.end_of_epoch(args, epoch, model, model_without_ddp, optimizer, loss_scaler, stats, n_parameters, args.save_ckpt, log_writer)

    final = {}
    if args.output_dir:
        if args.test_best:
            # every rank reads the checkpoint rank 0 wrote in the last end_of_epoch: wait until it is complete (the reference sleeps 10 s here,
            # run_stage2.py:825-829; the file itself appears atomically, utils.save_on_master)
            if utils.is_dist_avail_and_initialized():
                torch.distributed.barrier()
            utils.auto_load_model(args=args, model=model, model_without_ddp=model_without_ddp, optimizer=optimizer, loss_scaler=loss_scaler)
        final = test_and_merge()
        if final and utils.is_main_process():
            import json
            with open(os.path.join(args.output_dir, "log.txt"), mode="a", encoding="utf-8") as f:
                f.write(json.dumps(final) + "\n")
    print('Training time {}'.format(str(datetime.timedelta(seconds=int(time.time() - start_time)))))
    if utils.is_dist_avail_and_initialized():
        torch.distributed.destroy_process_group()
    return {**train_stats, **final}


if __name__ == '__main__':
    main(cli.get_args("stage2"))
