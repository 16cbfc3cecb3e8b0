"""Command-line / YAML surface of the three training drivers -- drop-in for ``get_args()`` of the reference's run_stage1.py:53-247,
run_stage2.py:54-303 and run_stage3.py:62-288: every flag with its option strings, dest, type, action and default, and the same
precedence: explicit command-line flag > ``--config`` YAML > parser default, then the ``--dataset`` entry of dataset_mappings.yaml
written over everything (run_stage1.py:231-270).  The flag tables are interface data; tests/test_cli.py checks them against the tables
read from the reference's source (oracle/make_golden_cli.py -> tests/golden/cli_flags.json) and checks the precedence rules.

Extra flags of this build (absent from the reference, all optional): ``--synthetic`` (seeded random clips instead of a dataset: the
reference has no offline data path), ``--synthetic_steps``, ``--dist_backend``, ``--clip_teacher_weights``, ``--clip_text_features`` / ``--clip_text_weights`` + ``--clip_bpe_vocab`` (stage 3).
"""
from __future__ import annotations

import argparse
import os
from typing import Optional, Sequence

import yaml


def str2bool(v):
    """reference src/utils.py:84-87"""
    if isinstance(v, bool):
        return v
    return v.lower() in ("yes", "true", "t", "1")


def A(*opts, **kw):
    return opts, kw


# flags grouped by the drivers that define them identically: ALL three, stages 1 and 3, stages 1 and 2, one stage only
ALL = [
    A('--batch_size', default=64, type=int),
    A('--model_key', default='model|module', type=str),
    A('--input_size', default=224, type=int),
    A('--tubelet_size', default=2, type=int),
    A('--use_learnable_pos_emb', action='store_true'),
    A('--opt', default='adamw', type=str),
    A('--opt_eps', default=1e-08, type=float),
    A('--opt_betas', default=None, type=float, nargs='+'),
    A('--clip_grad', default=None, type=float),
    A('--momentum', default=0.9, type=float),
    A('--weight_decay', default=0.05, type=float),
    A('--weight_decay_end', default=None, type=float),
    A('--warmup_lr', default=1e-06, type=float),
    A('--warmup_steps', default=-1, type=int),
    A('--use_checkpoint', action='store_true'),
    A('--checkpoint_num', default=0, type=int),
    A('--train_interpolation', default='bicubic', type=str),
    A('--dataset', default='', type=str),
    A('--prefix', default='', type=str),
    A('--split', default=' ', type=str),
    A('--train_fraction', default=1.0, type=float),
    A('--ann_file_train', default=None, type=str),
    A('--nb_classes', default=400, type=int),
    A('--ann_file_val', default=None, type=str),
    A('--ann_file_test', default=None, type=str),
    A('--imagenet_default_mean_and_std', default=True, action='store_true'),
    A('--num_segments', default=1, type=int),
    A('--num_frames', default=16, type=int),
    A('--sampling_rate', default=4, type=int),
    A('--device', default='cuda'),
    A('--seed', default=0, type=int),
    A('--resume', default=''),
    A('--auto_resume', action='store_true'),
    A('--no_auto_resume', action='store_false', dest='auto_resume'),
    A('--start_epoch', default=0, type=int),
    A('--test_best', action='store_true'),
    A('--num_workers', default=10, type=int),
    A('--pin_mem', action='store_true'),
    A('--no_pin_mem', action='store_false', dest='pin_mem'),
    A('--world_size', default=1, type=int),
    A('--local_rank', default=-1, type=int),
    A('--dist_on_itp', action='store_true'),
    A('--dist_url', default='env://'),
    A('--disable_wandb', default=False, action='store_true'),
    A('--output_dir', default=''),
    A('--wandb_group', default=None, type=str),
    A('--crop_pct', default=None, type=float),
    A('--short_side_size', default=224, type=int),
    A('--test_num_segment', default=5, type=int),
    A('--test_num_crop', default=3, type=int),
    A('--config', default='', type=str),
]
S13 = [
    A('--batch_size_val', default=64, type=int),
    A('--epochs', default=800, type=int),
    A('--save_ckpt_freq', default=50, type=int),
    A('--checkpoints_enabled', action='store_true'),
    A('--checkpoints_disabled', action='store_false', dest='checkpoints_enabled'),
    A('--model', default='pretrain_umt_base_patch16_224', type=str),
    A('--student_init', default='', type=str),
    A('--student_prefix', default='', type=str),
    A('--decoder_depth', default=4, type=int),
    A('--mask_type', default='attention', type=str, choices=['random', 'tube', 'attention']),
    A('--mask_ratio', default=0.75, type=float),
    A('--drop_path', default=0.0, type=float),
    A('--normlize_target', default=True, type=bool),
    A('--use_mean_pooling', action='store_false', dest='use_cls_token'),
    A('--use_cls_token', action='store_true', dest='use_cls_token'),
    A('--clip_teacher', default='clip_b16', type=str),
    A('--clip_input_resolution', default=224, type=int),
    A('--clip_loss_ratio', default=1.0, type=float),
    A('--clip_loss_type', default='l2', type=str),
    A('--clip_loss_data', default='mixed', type=str),
    A('--clip_decoder_type', default='SA_Decoder', type=str),
    A('--clip_decoder_embed_dim', default=512, type=int),
    A('--clip_output_dim', default=768, type=int),
    A('--clip_norm_type', default='l2', type=str),
    A('--clip_return_attn', default=False, type=bool),
    A('--clip_return_layers', default=[6, 7, 8, 9, 10, 11], type=int, nargs='+'),
    A('--clip_return_interval', default=1, type=float),
    A('--clip_student_return_interval', default=1, type=float),
    A('--freeze_clip_decoders', default=False, action='store_true'),
    A('--no_freeze_clip_decoders', action='store_false', dest='freeze_clip_decoders'),
    A('--lr', default=0.00015, type=float),
    A('--min_lr', default=1e-05, type=float),
    A('--layer_decay', default=1.0, type=float),
    A('--warmup_epochs', default=40, type=int),
    A('--num_sample', default=1, type=int),
    A('--color_jitter', default=0.0, type=float),
    A('--flip', default=False),
    A('--data_set', default='Kinetics_sparse', type=str),
    A('--ann_file_train_target', default=None, type=str),
    A('--use_decord', default=True),
    A('--umt_step', default=1, type=int),
    A('--log_freq', default=10, type=int),
    A('--val_interval', default=1, type=int),
    A('--initial_validation', default=False, action='store_true'),
]
S1 = [
    A('--clip_decoder_init'),
    A('--ann_file_train_knn', default=None, type=str),
]
S2 = [
    A('--epochs', default=30, type=int),
    A('--save_ckpt_freq', default=100, type=int),
    A('--model', default='vit_base_patch16_224', type=str),
    A('--drop_path', default=0.1, type=float),
    A('--use_mean_pooling', action='store_true'),
    A('--lr', default=0.001, type=float),
    A('--min_lr', default=1e-06, type=float),
    A('--layer_decay', default=0.75, type=float),
    A('--warmup_epochs', default=5, type=int),
    A('--num_sample', default=2, type=int),
    A('--color_jitter', default=0.4, type=float),
    A('--data_set', default='Kinetics', type=str, choices=['Kinetics', 'Kinetics_sparse', 'SSV2', 'UCF101', 'HMDB51', 'image_folder', 'mitv1_sparse']),
    A('--use_decord', default=False, action='store_true'),
    A('--reprob', default=0.25, type=float),
    A('--eval', default=False, type=str2bool, nargs='?', const=True),
    A('--update_freq', default=1, type=int),
    A('--train_head_only', default=False, action='store_true'),
    A('--frozen_layers', default='', type=str),
    A('--freeze_patch_embedding', default=False, type=str2bool, nargs='?', const=True),
    A('--head_type', default='linear', type=str, choices=['linear', 'mlp']),
    A('--head_hidden_dim', default=256, type=int),
    A('--fc_drop_rate', default=0.0, type=float),
    A('--drop', default=0.0, type=float),
    A('--attn_drop_rate', default=0.0, type=float),
    A('--disable_eval_during_finetuning', default=False, action='store_true'),
    A('--model_ema', default=False, action='store_true'),
    A('--model_ema_decay', default=0.9999, type=float),
    A('--model_ema_force_cpu', default=False, action='store_true'),
    A('--lr_schedule', default='cosine', type=str, choices=['constant', 'cosine', 'step']),
    A('--step_fraction', default=0.1, type=float),
    A('--lr_step_epochs', default=None, type=int, nargs='+'),
    A('--aa', default='rand-m7-n4-mstd0.5-inc1', type=str),
    A('--smoothing', default=0.1, type=float),
    A('--remode', default='pixel', type=str),
    A('--recount', default=1, type=int),
    A('--resplit', default=False, action='store_true'),
    A('--mixup', default=0.8, type=float),
    A('--cutmix', default=1.0, type=float),
    A('--cutmix_minmax', default=None, type=float, nargs='+'),
    A('--mixup_prob', default=1.0, type=float),
    A('--mixup_switch_prob', default=0.5, type=float),
    A('--mixup_mode', default='batch', type=str),
    A('--finetune', default=''),
    A('--delete_head', action='store_true'),
    A('--no_delete_head', action='store_false', dest='delete_head'),
    A('--model_prefix', default='', type=str),
    A('--init_scale', default=0.001, type=float),
    A('--use_cls', action='store_false', dest='use_mean_pooling'),
    A('--data_path', default='you_data_path', type=str),
    A('--eval_data_path', default=None, type=str),
    A('--reset_train_dataset', action='store_true'),
    A('--no_reset_train_dataset', action='store_false', dest='reset_train_data'),
    A('--save_ckpt', action='store_true'),
    A('--no_save_ckpt', action='store_false', dest='save_ckpt'),
    A('--dist_eval', default=False, action='store_true'),
    A('--auto_reload', action='store_true'),
    A('--no_auto_reload', action='store_false', dest='auto_reload'),
    A('--eval_freq', default=1, type=int),
    A('--lp_ft_epochs', default=0, type=int),
    A('--distributed', default=False, action='store_true'),
    A('--enable_deepspeed', default=False, action='store_true'),
]
S3 = [
    A('--clip_decoder_init', default='/cis/home/areddy/unmasked_teacher/checkpoints/b16_ptk710_f8_res224.pth'),
    A('--train_repetitions', default=0, type=int),
    A('--wandb_entity', default='targeted-ssda2', type=str),
    A('--wandb_project', default='umt', type=str),
    A('--class_loss_src_ratio', default=0.0, type=float),
    A('--src_classifier_type', default='linear', type=str),
    A('--unmasked_classification', default=False, action='store_true'),
    A('--pseudolabel_threshold', default=0.0, type=float),
    A('--target_only_classification', default=False, action='store_true'),
    A('--reprob', default=0.0, type=float),
    A('--eval', default=False, action='store_true'),
    A('--return_aug_for_val', default=False, action='store_true'),
    A('--full_oracle', default=False, type=str2bool),
    A('--conf_weighted_loss', default=False, type=str2bool),
    A('--class_loss_tgt_ratio', default=0.1, type=float),
    A('--class_loss_src_ratio_pl', default=1.0, type=float),
    A('--clip_threshold', default=0.5, type=float),
    A('--train_masked', default=True, type=str2bool),
    A('--selection_strategy', default='conf', type=str),
    A('--masking_type', default='clip_attention', type=str),
    A('--add_cons_constraint', default=False, type=str2bool),
]
S23 = [

]
S12 = [
    A('--train_repetitions', default=1, type=int),
    A('--wandb_entity', type=str),
    A('--wandb_project', type=str),
]

SET_DEFAULTS = {
    "stage1": dict(auto_resume=True, checkpoints_enabled=True, pin_mem=True, use_checkpoint=False, use_cls_token=True, use_learnable_pos_emb=False),
    "stage2": dict(auto_reload=True, auto_resume=True, pin_mem=True, reset_train_data=False, save_ckpt=True, use_checkpoint=False,
                   use_learnable_pos_emb=False, use_mean_pooling=False),
    "stage3": dict(auto_resume=True, checkpoints_enabled=True, pin_mem=True, use_checkpoint=False, use_cls_token=True, use_learnable_pos_emb=False),
}
STAGE_GROUPS = {"stage1": (ALL, S13, S12, S1), "stage2": (ALL, S12, S2), "stage3": (ALL, S13, S3)}
TITLES = {"stage1": "UMT Adaptation Script", "stage2": "VideoMAE fine-tuning and evaluation script for video classification",
          "stage3": "UMT Adaptation Script"}
EXTRA = [
    A("--synthetic", action="store_true", default=False),
    A("--synthetic_steps", default=20, type=int),
    A("--dist_backend", default="nccl", type=str),
    A("--clip_teacher_weights", default="", type=str),
    A("--clip_text_features", default="", type=str),      # stage 3: class text embeddings (nb_classes, C) for the zero-shot CLIP side (.pt / .npy)
    A("--clip_text_weights", default="", type=str),       # ... or OpenAI CLIP's text-side weights (state dict) and
    A("--clip_bpe_vocab", default="", type=str),          # its BPE merge table (bpe_simple_vocab_16e6.txt.gz): unite_amd/clip_text.py
]


def flag_table(stage: str):
    """[(option strings, add_argument kwargs)] of one driver, reference flags only"""
    return [f for grp in STAGE_GROUPS[stage] for f in grp]


def build_parser(stage: str, extras: bool = True) -> argparse.ArgumentParser:
    parser = argparse.ArgumentParser(TITLES[stage], add_help=False)
    for opts, kw in flag_table(stage) + (EXTRA if extras else []):
        parser.add_argument(*opts, **kw)
    parser.set_defaults(**SET_DEFAULTS[stage])
    return parser


def update_dataset_args_from_yaml(args, mappings_path: Optional[str] = None):
    """run_stage1.py:250-270: the entry of dataset_mappings.yaml named by --dataset is written over the parsed values."""
    path = mappings_path or os.environ.get("UNITE_DATASET_MAPPINGS") or os.path.join(os.getcwd(), "dataset_mappings.yaml")
    if not os.path.exists(path):
        print("No dataset_mappings.yaml file found, skipping update_dataset_args_from_yaml!")
        raise FileNotFoundError(path)
    with open(path, "r") as f:
        mappings = yaml.safe_load(f)
    try:
        entry = mappings[args.dataset]
    except KeyError:
        print(f"Dataset <{args.dataset}> not found in dataset_mappings.yaml")
        raise
    for k, v in entry.items():
        setattr(args, k, v)
        print("Updated %s to %s" % (k, v))
    return args


def get_args(stage: str, argv: Optional[Sequence[str]] = None, mappings_path: Optional[str] = None):
    """Namespace of one driver.  As in the reference the YAML values are loaded into the namespace FIRST and the command line is parsed
    on top of it: argparse then fills defaults only for what the YAML did not name, and flags given explicitly win."""
    parser = build_parser(stage)
    cmd = parser.parse_args(argv)
    if cmd.config:
        ns = argparse.Namespace()
        with open(cmd.config, "r") as f:
            ns.__dict__ = yaml.safe_load(f) or {}
        args = parser.parse_args(argv, namespace=ns)
    else:
        args = cmd
    if args.dataset:
        args = update_dataset_args_from_yaml(args, mappings_path)
    return args
