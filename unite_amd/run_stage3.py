"""Stage-3 driver (collaborative self-training): ``torchrun --nproc_per_node=N -m unite_amd.run_stage3 --config
configs/stage3_config.yaml [--synthetic]`` -- reference stage3.sh -> run_stage3.py:992-1414 ``main``.

Sequence of the reference's main(): run set-up, source / target loaders, student (``get_model`` :311-331, ``load_student_from_ckpt``), frozen
CLIP mask teacher, the source classifier -- nn.Linear or a two-layer MLP on the pooled encoder output, initialised from the stage-2
``head.*`` rows of --student_init (:1203-1211) -- which is kept OUTSIDE the optimizer and outside DistributedDataParallel exactly as the
reference leaves it (:1264 builds the optimizer from the student only: SURVEY Appendix A-8), linear lr scaling by the global batch x 2
domains, layer-decay assigner (inert on 'encoder.*' names, A-7), cosine schedules, optional initial validation, per epoch
``train_one_epoch`` + validation every ``val_interval`` epochs, checkpoints + ``src_classifier_latest.pth`` (:1371-1372), log.txt, and at the
end ``final_test`` over the test views + ``merge`` -> 'Final top-1 / Final Top-5' in log.txt (:1393-1409).
Input: --synthetic, or ``ann_file_train`` + ``ann_file_train_target`` (+ val / test lists) through unite_amd/datasets_cls.py.  Not here: OpenAI CLIP's text tower -- the zero-shot probabilities of the clip_* selection strategies come
from the frozen CLIP IMAGE tower of this build (random-init offline, or --clip_teacher_weights) against class text features loaded from
``--clip_text_features`` (a .pt / .npy of shape (nb_classes, C)) or, under --synthetic, seeded random ones."""
from __future__ import annotations

import datetime
import os
import time
from functools import partial
from pathlib import Path

import torch
import torch.nn as nn

from . import cli, clip, launch, utils
from .checkpoint import load_student_from_ckpt_stage3, read_checkpoint
from .engine_for_finetuning import merge
from .engine_stage3 import final_test, train_one_epoch, validation_one_epoch
from .optim_factory import create_optimizer
from .registry import create_model
from .utils import NativeScalerWithGradNormCount as NativeScaler


def get_model(args):
    """run_stage3.py:311-331"""
    print(f"Creating model: {args.model}")
    return create_model(
        args.model, pretrained=False, drop_path_rate=args.drop_path, drop_block_rate=None, use_learnable_pos_emb=args.use_learnable_pos_emb,
        use_checkpoint=args.use_checkpoint, checkpoint_num=args.checkpoint_num, clip_decoder_embed_dim=args.clip_decoder_embed_dim,
        clip_output_dim=args.clip_output_dim, clip_norm_type=args.clip_norm_type, num_frames=args.num_frames, tubelet_size=args.tubelet_size,
        clip_return_layers=args.clip_return_layers, clip_student_return_interval=args.clip_student_return_interval, use_cls_token=args.use_cls_token)


def build_src_classifier(args, device):
    """run_stage3.py:1184-1227"""
    if args.class_loss_src_ratio < 0:          # negative: no classifier at all
        return None
    D = args.clip_decoder_embed_dim
    if args.src_classifier_type == 'linear':
        head = nn.Linear(D, args.nb_classes)
    elif args.src_classifier_type == 'mlp':
        head = nn.Sequential(nn.Linear(D, D), nn.ReLU(), nn.Linear(D, args.nb_classes))
    else:
        raise NotImplementedError('Unknown source classifier type!')
    if args.student_init:
        print("Loading source classifier head from %s" % args.student_init)
        sd = read_checkpoint(args.student_init)
        sd = sd.get('model', sd)
        utils.load_state_dict(head, {k.split('.')[1]: v for k, v in sd.items() if k.startswith('head')}, prefix='')
    head.to(device)
    print(f'Source classifier: {args.src_classifier_type}')
    print('Source classifier Params: {} M'.format(utils.count_parameters(head) / 1e6))
    return head


def zero_shot_probs_fn(args, device, seed):
    """``clip_probs_fn(videos_t) -> (B_t, nb_classes)`` for the clip_only / clip_matchORconf strategies (run_stage3.py:376-377, 551-560:
    utils.setup_clip + utils.clip_infer), or None for the strategies that do not use CLIP."""
    if args.selection_strategy not in ('clip_matchORconf', 'clip_only'):
        return None
    if args.synthetic and not (getattr(args, "clip_text_features", "") or getattr(args, "clip_text_weights", "")):
        weights = getattr(args, "clip_teacher_weights", "") or os.environ.get("UNITE_CLIP_PATH", "")
        tower = clip.clip_b16(pretrained=bool(weights), return_attn=False, clip_return_layers=[11]).to(device)      # utils.setup_clip loads "ViT-B/16"
        text = torch.randn(args.nb_classes, tower.output_dim, generator=torch.Generator().manual_seed(seed + 99)).to(device)
    else:
        tower, text = utils.setup_clip(args, device)          # run_stage3.py:376-377
    if text.shape != (args.nb_classes, tower.output_dim):
        raise ValueError(f"text features must be ({args.nb_classes}, {tower.output_dim}), got {tuple(text.shape)}")
    return partial(lambda videos, m, t: utils.clip_infer(m, videos, t), m=tower, t=text)


def main(args):
    device, seed = launch.start_run(args)
    T, size, nb = args.num_frames, args.input_size, args.nb_classes
    if args.synthetic:
        def source_batch(g, B):      # (videos, labels, ...) -- run_stage3.py:399-401
            return launch.clips(g, B, T, size, device), launch.labels(g, B, nb, device)

        def target_batch(g, B):      # (videos, augmented videos, labels) with return_aug_for_val (:404-411)
            v = launch.clips(g, B, T, size, device)
            return v, v + 0.1 * launch.clips(g, B, T, size, device), launch.labels(g, B, nb, device)

        def val_batch(g, B):         # validation_one_epoch reads batch[0] and batch[2] (or [1]) (:736-742)
            return launch.clips(g, B, T, size, device), launch.labels(g, B, nb, device), launch.labels(g, B, nb, device)

        def test_batch(g, B):        # (videos, label, id, chunk, split) -- the test-mode dataset's tuple, final_test reads all five (:939-944)
            ids = [f"video_{utils.get_rank()}_{int(torch.randint(0, 1 << 30, (1,), generator=g, device=device))}" for _ in range(B)]
            return launch.clips(g, B, T, size, device), launch.labels(g, B, nb, device), ids, [0] * B, [0] * B
        data_loader_train = launch.SyntheticLoader(args.synthetic_steps, args.batch_size, device, seed, source_batch)
        data_loader_train_target = launch.SyntheticLoader(args.synthetic_steps, args.batch_size, device, seed + 1, target_batch)
        data_loader_val = launch.SyntheticLoader(2, 2 * args.batch_size, device, seed + 2, val_batch)
        data_loader_test = launch.SyntheticLoader(2, 2 * args.batch_size, device, seed + 3, test_batch)
    elif getattr(args, "ann_file_train", None) and getattr(args, "ann_file_train_target", None):
        # real video lists (run_stage3.py:1042-1145): the source list in train mode, the target list in VALIDATION mode with its augmented second
        # view (return_aug_for_val), the shorter of the two repeated; validation / test batches of batch_size_val
        bv = int(getattr(args, "batch_size_val", None) or 2 * args.batch_size)
        ld = launch.cls_loaders(args, device, utils.get_world_size(), utils.get_rank(), (args.batch_size, bv, bv), with_val=True, dist_eval=True,
                                train_repetitions=getattr(args, "train_repetitions", 0), target_annotation=args.ann_file_train_target)
        data_loader_train, data_loader_train_target, data_loader_val, data_loader_test = ld["train"], ld["target"], ld["val"], ld["test"]
    else:
        launch.require_synthetic(args, "unite_amd.engine_stage3.train_one_epoch")
    log_writer = launch.make_log_writer(args)
    num_training_steps_per_epoch = len(data_loader_train)

    model = get_model(args)
    if args.student_init:
        model = load_student_from_ckpt_stage3(args, model)
        print("Loaded student model!")
    patch_size = model.encoder.patch_embed.patch_size
    args.window_size = (args.num_frames // args.tubelet_size, args.input_size // patch_size[0], args.input_size // patch_size[1])
    args.patch_size = patch_size
    model.to(device)
    model_without_ddp = model
    n_parameters = utils.count_parameters(model)
    print('Student Params: {} M'.format(n_parameters / 1e6))

    weights = getattr(args, "clip_teacher_weights", "") or os.environ.get("UNITE_CLIP_PATH", "")
    if weights:
        os.environ["UNITE_CLIP_PATH"] = weights
    teacher_model = getattr(clip, args.clip_teacher)(                      # the reference resolves the name with eval(), :1176
        pretrained=bool(weights), clip_norm_type=args.clip_norm_type, input_resolution=args.clip_input_resolution,
        return_attn=True, clip_return_layers=args.clip_return_layers, clip_return_interval=args.clip_return_interval)
    teacher_model.to(device)
    print(f'Teacher model: {args.clip_teacher}')
    src_classifier = build_src_classifier(args, device)
    clip_probs_fn = zero_shot_probs_fn(args, device, seed)

    total_batch_size = args.batch_size * utils.get_world_size() * 2        # source + target loaders (:1231)
    scale = total_batch_size * args.num_sample / 256
    args.lr, args.min_lr, args.warmup_lr = args.lr * scale, args.min_lr * scale, args.warmup_lr * scale
    print("LR = %.8f" % args.lr)
    print("Batch size = %d" % total_batch_size)
    print("Number of training steps per epoch = %d" % num_training_steps_per_epoch)

    if args.distributed:
        from .ddp import DistributedDataParallel
        # clip_decoder.* receives no gradient in stage 3 (the reference passes find_unused_parameters=True, :1246); the source classifier
        # and the frozen teacher stay plain replicas (A-8, A-18)
        model = DistributedDataParallel(model, device_ids=[args.gpu], find_unused_parameters=True)
        model_without_ddp = model.module
    assigner = launch.layer_decay_assigner(args.layer_decay, 12)           # num_layers = 12 hard-coded in the reference (:1244)
    optimizer = create_optimizer(args, model_without_ddp, skip_list=model_without_ddp.no_weight_decay(),
                                 get_num_layer=assigner.get_layer_id if assigner is not None else None,
                                 get_layer_scale=assigner.get_scale if assigner is not None else None)
    loss_scaler = NativeScaler()
    lr_schedule_values = utils.cosine_scheduler(args.lr, args.min_lr, args.epochs, num_training_steps_per_epoch,
                                                warmup_epochs=args.warmup_epochs, warmup_steps=args.warmup_steps)
    if args.weight_decay_end is None:
        args.weight_decay_end = args.weight_decay
    wd_schedule_values = utils.cosine_scheduler(args.weight_decay, args.weight_decay_end, args.epochs, num_training_steps_per_epoch)

    val_stats = {}
    if args.initial_validation and src_classifier is not None:
        print("Performing initial validation with source only model...")
        val_stats = validation_one_epoch(data_loader_val, model_without_ddp, src_classifier, device, args=args)
    if args.auto_resume and args.output_dir:
        utils.auto_load_model(args=args, model=model, model_without_ddp=model_without_ddp, optimizer=optimizer, loss_scaler=loss_scaler)
    print(f"Start training for {args.epochs} epochs")
    print(f"Mask ratio: {args.mask_ratio}")
    start_time = time.time()
    train_stats = {}
    for epoch in range(args.start_epoch, args.epochs):
        data_loader_train.sampler.set_epoch(epoch)
        data_loader_train_target.sampler.set_epoch(epoch)
        if log_writer is not None:
            log_writer.set_step(epoch * num_training_steps_per_epoch)
        train_stats = train_one_epoch(
            model, data_loader_train, data_loader_train_target, optimizer, device, epoch, loss_scaler, args.clip_grad, log_writer=log_writer,
            start_steps=epoch * num_training_steps_per_epoch, lr_schedule_values=lr_schedule_values, wd_schedule_values=wd_schedule_values,
            src_classifier=src_classifier, teacher_model=teacher_model, clip_input_resolution=args.clip_input_resolution,
            clip_loss_type=args.clip_loss_type, clip_loss_ratio=args.clip_loss_ratio, mask_type=args.mask_type, mask_ratio=args.mask_ratio,
            use_wandb=False, args=args, classwise_thresholds=[0] * args.nb_classes, global_threshold=0, clip_probs_fn=clip_probs_fn)
        stats = {f'train_{k}': v for k, v in train_stats.items()}
        if (epoch + 1) % args.val_interval == 0 and src_classifier is not None:
            val_stats = validation_one_epoch(data_loader_val, model_without_ddp, src_classifier, device, args=args)
            stats.update({f'val_{k}': v for k, v in val_stats.items()})
        launch.end_of_epoch(args, epoch, model, model_without_ddp, optimizer, loss_scaler, stats, n_parameters, args.checkpoints_enabled, log_writer)
        if args.output_dir and args.checkpoints_enabled and utils.is_main_process() and src_classifier is not None:
            utils.save_on_master(src_classifier.state_dict(), Path(args.output_dir) / "src_classifier_latest.pth")
    print('Training time {}'.format(str(datetime.timedelta(seconds=int(time.time() - start_time)))))

    # ---- testing (run_stage3.py:1393-1409): per-view logits of every rank -> '<rank>.txt', merged on rank 0 into the run's headline accuracy
    final = {}
    if args.output_dir and src_classifier is not None:
        global_rank, num_tasks = utils.get_rank(), utils.get_world_size()
        final_test(data_loader_test, model, src_classifier, device, os.path.join(args.output_dir, str(global_rank) + '.txt'), args)
        if utils.is_dist_avail_and_initialized():
            torch.distributed.barrier()
        if global_rank == 0:
            print("Start merging results...")
            top1, top5 = merge(args.output_dir, num_tasks)
            print(f"Accuracy of the network on the test videos: Top-1: {top1:.2f}%, Top-5: {top5:.2f}%")
            final = {'Final top-1': top1, 'Final Top-5': top5}
            import json
            with open(os.path.join(args.output_dir, "log.txt"), mode="a", encoding="utf-8") as f:
                f.write(json.dumps(final) + "\n")
    if utils.is_dist_avail_and_initialized():
        torch.distributed.destroy_process_group()
    return {**train_stats, **{f'val_{k}': v for k, v in val_stats.items()}, **final}


if __name__ == '__main__':
    main(cli.get_args("stage3"))
