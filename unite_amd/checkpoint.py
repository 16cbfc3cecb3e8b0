"""Checkpoint hand-off between the stages -- drop-in for the loaders of the reference drivers:

  run_stage1.py:518-602  load_student_from_ckpt   (UMT K710 encoder  -> stage-1 student, + optional decoders)
  run_stage2.py:349-438  load_from_ckpt           (stage-1 student   -> stage-2 classifier: 'encoder.' / 'backbone.' stripped,
                                                   head handling, position-table interpolation)
  run_stage3.py:829-924  load_student_from_ckpt   (stage-1/2 weights -> stage-3 student)

Same ``args`` fields, same key rewrites in the same order, same ``utils.load_state_dict`` (non-strict, prefix-aware) at the end.
Host logic only (torch CPU tensors); nothing here touches the GPU path.  The reference reads checkpoints with a plain
``torch.load`` (they carry an argparse Namespace): ``read_checkpoint`` uses the tensor-only loader with that class allow-listed and
unpickles only on an explicit opt-in.
"""
from __future__ import annotations

import json
from collections import OrderedDict

import torch

from . import utils


def read_checkpoint(path: str, trusted: bool = False):
    """Tensor-only load (``weights_only=True``: nothing in the file is executed) with the one non-tensor class the reference's
    checkpoints carry allow-listed -- an ``argparse.Namespace`` under 'args' (utils.py:700-704 saves ``args`` itself).  Files that
    still refuse (arbitrary pickled objects) are loaded with the executing unpickler only when the caller says the file is its own
    (``trusted``) or the user opts in with UNITE_UNSAFE_CHECKPOINT_LOAD=1: ``student_init`` / ``finetune`` / ``clip_decoder_init``
    usually point at third-party downloads."""
    import argparse
    import os
    if str(path).startswith("https"):
        raise RuntimeError("remote checkpoints are not fetched (no network in this build); download the file and pass its path")
    import pickle
    try:
        with torch.serialization.safe_globals([argparse.Namespace]):
            return torch.load(path, map_location="cpu", weights_only=True)
    except pickle.UnpicklingError as e:      # the tensor-only loader's refusal; a missing / unreadable / corrupt file raises what it raises
        if not (trusted or os.environ.get("UNITE_UNSAFE_CHECKPOINT_LOAD", "0") == "1"):
            raise RuntimeError(
                f"{path}: refused by the tensor-only loader ({type(e).__name__}: {str(e).splitlines()[0][:200]}).  If you trust the file, "
                "set UNITE_UNSAFE_CHECKPOINT_LOAD=1 to unpickle it as the reference's torch.load does (this can execute code from the file).") from e
        print(f"WARNING: unpickling {path} with weights_only=False (it can execute code from the file)")
        return torch.load(path, map_location="cpu", weights_only=False)


def _select(checkpoint, model_key: str):
    """first of the '|'-separated keys present in the file, else the file itself (run_stage1.py:523-533)"""
    for key in model_key.split('|'):
        if key in checkpoint:
            print("Load state_dict by model_key = %s" % key)
            return checkpoint[key], True
    return checkpoint, False


def _resize_time(grid: torch.Tensor, frames: int) -> torch.Tensor:
    """(t, s, s, C) position grid -> (frames, s, s, C): linear along the frame axis, one 1-D signal per (cell, channel)"""
    t, s, _, C = grid.shape
    lines = grid.reshape(t, s * s * C).t().unsqueeze(0)                                   # (1, cells x channels, t)
    lines = torch.nn.functional.interpolate(lines, size=frames, mode='linear')
    return lines.squeeze(0).t().reshape(frames, s, s, C)


def _resize_space(grid: torch.Tensor, side: int) -> torch.Tensor:
    """(t, s, s, C) position grid -> (t, side, side, C): bicubic inside every frame"""
    planes = torch.nn.functional.interpolate(grid.permute(0, 3, 1, 2), size=(side, side), mode='bicubic', align_corners=False)
    return planes.permute(0, 2, 3, 1)


def interpolate_pos_embed(checkpoint_model, model, num_frames: int, patch_embed=None, pretrain_frames: int = 8):
    """The position table of a checkpoint fitted to ``model``'s token grid, as the reference does it in run_stage1.py:553-591 ==
    run_stage2.py:396-434 == run_stage3.py:875-913: the checkpoints were pre-trained on ``pretrain_frames`` frames, so the table is a
    (frames, side, side) grid behind the class / dist tokens; it is resized linearly along the frame axis first and bicubically
    inside every frame second, the extra tokens are kept.  In place on checkpoint_model['pos_embed'] (tests/golden/stage2_ckpt.npz:
    the reference's own load_from_ckpt on the same files).  With extra tokens the reference's frame resize reshapes them INTO the
    grid and fails; here they are set aside first."""
    table = checkpoint_model.get('pos_embed')
    if table is None:
        return checkpoint_model
    pe = patch_embed if patch_embed is not None else model.patch_embed
    C = table.shape[-1]
    n_extra = model.pos_embed.shape[-2] - pe.num_patches
    t_old, t_new = pretrain_frames // pe.tubelet_size, num_frames // pe.tubelet_size
    s_old = int(((table.shape[-2] - n_extra) // t_old) ** 0.5)
    s_new = int((pe.num_patches // t_new) ** 0.5)
    if (t_old, s_old) == (t_new, s_new):
        return checkpoint_model
    extra, grid = table[0, :n_extra], table[0, n_extra:].reshape(t_old, s_old, s_old, C)
    if t_old != t_new:
        print(f"Temporal interpolate from {t_old} to {t_new}")
        grid = _resize_time(grid, t_new)
    if s_old != s_new:
        print("Position interpolate from %dx%d to %dx%d" % (s_old, s_old, s_new, s_new))
        grid = _resize_space(grid, s_new)
    checkpoint_model['pos_embed'] = torch.cat((extra, grid.reshape(-1, C)), dim=0).unsqueeze(0)
    return checkpoint_model


def _strip_backbone(checkpoint_model, strip_encoder: bool):
    new_dict = OrderedDict()
    for key in list(checkpoint_model.keys()):
        if key.startswith('backbone.'):
            new_dict[key[9:]] = checkpoint_model[key]
        elif strip_encoder and key.startswith('encoder.'):
            new_dict[key[8:]] = checkpoint_model[key]
        else:
            new_dict[key] = checkpoint_model[key]
    return new_dict


def _finish_student(args, model, checkpoint_model):
    if getattr(args, "clip_decoder_init", None):
        decoder_ckpt = read_checkpoint(args.clip_decoder_init)
        checkpoint_model.update({k: v for k, v in decoder_ckpt.items() if k.startswith('clip_decoder.')})
        print("Loaded decoder params from %s" % args.clip_decoder_init)
    # the reference indexes model.patch_embed / model.pos_embed here, which only exist on the encoder of the student wrapper:
    # a learnable 'pos_embed' in the file therefore crashes it; this build interpolates against the encoder (sinusoid students
    # simply report the key as unused, as utils.load_state_dict does)
    enc = getattr(model, "encoder", model)
    if 'pos_embed' in checkpoint_model and hasattr(enc, "pos_embed") and hasattr(enc, "patch_embed"):
        interpolate_pos_embed(checkpoint_model, enc, args.num_frames)
    utils.load_state_dict(model, checkpoint_model, prefix=getattr(args, "student_prefix", ''))
    if getattr(args, "freeze_clip_decoders", False):
        for name, param in model.named_parameters():
            if name.startswith('clip_decoder.'):
                param.requires_grad = False
                print("Freezing %s" % name)
    return model


def load_student_from_ckpt(args, model):
    """run_stage1.py:518-602: every key of the selected state_dict gets the 'encoder.' prefix (UMT encoder-only checkpoints)."""
    print("Loading student model from %s" % args.student_init)
    checkpoint = read_checkpoint(args.student_init)
    checkpoint_model, found = _select(checkpoint, args.model_key)
    if found:
        checkpoint_model = {f'encoder.{k}': v for k, v in checkpoint_model.items()}
    return _finish_student(args, model, _strip_backbone(checkpoint_model, strip_encoder=False))


def load_student_from_ckpt_stage3(args, model):
    """run_stage3.py:829-924: as stage 1, but a state_dict that already starts with 'encoder.' (a stage-1 / stage-2 output of this
    pipeline) is taken as it is.  (The reference's two hard-wired /cis/home paths for 'umt_k710*' are site-specific and not kept.)"""
    print("Loading student model from %s" % args.student_init)
    checkpoint = read_checkpoint(args.student_init)
    checkpoint_model, found = _select(checkpoint, args.model_key)
    if found and not list(checkpoint_model.keys())[0].startswith('encoder.'):
        checkpoint_model = {f'encoder.{k}': v for k, v in checkpoint_model.items()}
    return _finish_student(args, model, _strip_backbone(checkpoint_model, strip_encoder=False))


def _fit_head(sd, args):
    """the classifier rows of a pre-trained checkpoint against this run's classes (run_stage2.py:366-382): dropped on --delete_head; a
    Kinetics-710 head keeps its first 400 rows for K400 or the rows a label map names for K600 / K700; anything else is left alone (a
    shape mismatch is then reported and skipped by utils.load_state_dict)"""
    if 'head.weight' not in sd:
        return
    if getattr(args, "delete_head", False):
        print("Removing head from pretrained checkpoint")
        rows = None
    elif sd['head.weight'].shape[0] != 710:
        return
    elif args.nb_classes == 400:
        rows = slice(0, 400)
    elif args.nb_classes in (600, 700):
        map_path = f'k710/label_mixto{args.nb_classes}.json'
        print(f'Load label map from {map_path}')
        with open(map_path) as f:
            rows = json.load(f)
    else:
        return
    for key in ('head.weight', 'head.bias'):
        if rows is None:
            del sd[key]
        else:
            sd[key] = sd[key][rows]


def load_from_ckpt(args, model):
    """run_stage2.py:349-438: classifier initialisation from a pre-trained / stage-1 checkpoint."""
    checkpoint = read_checkpoint(args.finetune)
    print("Load ckpt from %s" % args.finetune)
    checkpoint_model, _ = _select(checkpoint, args.model_key)
    _fit_head(checkpoint_model, args)
    checkpoint_model = _strip_backbone(checkpoint_model, strip_encoder=True)
    if 'pos_embed' in checkpoint_model and hasattr(model, "pos_embed"):
        interpolate_pos_embed(checkpoint_model, model, args.num_frames)
    utils.load_state_dict(model, checkpoint_model, prefix=getattr(args, "model_prefix", ''))
    return model
