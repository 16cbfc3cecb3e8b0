"""Optimizer construction -- drop-in for reference src/optim_factory.py (AdamW path, the only one any UNITE config
selects: ``opt: adamw`` in configs/stage{1,2,3}_config.yaml).

``create_optimizer`` keeps the reference's signature, parameter grouping (no_decay for 1-D tensors / ``.bias`` /
``no_weight_decay()`` names, optional ``layer_{id}_`` split with ``lr_scale``) and returns a ``torch.optim.Optimizer``
subclass whose ``param_groups`` the engines edit exactly as before (run_stage1.py:326-338).  ``step()`` is ONE launch
of ``unite_adamw_flat`` over the model's flat parameter buffer, which also refreshes the bf16 weights the GEMMs read.
"""
from __future__ import annotations

import math

from typing import Dict, List, Optional

import torch

from . import ops
from .flat_params import FlatParams


def get_num_layer_for_vit(var_name, num_max_layer):
    """reference optim_factory.py:44-62 (note: 'encoder.blocks.*' names fall through to the last id, SURVEY A-7)."""
    if var_name in ("cls_token", "mask_token", "pos_embed"):
        return 0
    elif var_name.startswith("patch_embed"):
        return 0
    elif var_name.startswith("rel_pos_bias"):
        return num_max_layer - 1
    elif var_name.startswith("blocks"):
        return int(var_name.split('.')[1]) + 1
    elif var_name.startswith("transformer.resblocks"):
        return int(var_name.split('.')[2]) + 1
    elif var_name in ("class_embedding", "positional_embedding", "temporal_positional_embedding"):
        return 0
    elif var_name.startswith("conv1"):
        return 0
    else:
        return num_max_layer - 1


class LayerDecayValueAssigner(object):
    def __init__(self, values):
        self.values = values

    def get_scale(self, layer_id):
        return self.values[layer_id]

    def get_layer_id(self, var_name):
        return get_num_layer_for_vit(var_name, len(self.values))


def get_parameter_groups(model, weight_decay=1e-5, skip_list=(), get_num_layer=None, get_layer_scale=None, with_names=False):
    """reference optim_factory.py:76-118; group order = first appearance in named_parameters()."""
    names: Dict[str, dict] = {}
    groups: Dict[str, dict] = {}
    for name, param in model.named_parameters():
        if not param.requires_grad:
            continue
        if len(param.shape) == 1 or name.endswith(".bias") or name in skip_list:
            group_name, this_wd = "no_decay", 0.
        else:
            group_name, this_wd = "decay", weight_decay
        layer_id = None
        if get_num_layer is not None:
            layer_id = get_num_layer(name)
            group_name = "layer_%d_%s" % (layer_id, group_name)
        if group_name not in groups:
            scale = get_layer_scale(layer_id) if get_layer_scale is not None else 1.
            groups[group_name] = {"weight_decay": this_wd, "params": [], "lr_scale": scale}
            names[group_name] = {"weight_decay": this_wd, "params": [], "lr_scale": scale}
        groups[group_name]["params"].append(param)
        names[group_name]["params"].append(name)
    if with_names:
        return list(groups.values()), list(names.values())
    return list(groups.values())


class FusedAdamW(torch.optim.Optimizer):
    """torch.optim.AdamW semantics (decoupled weight decay, bias correction) in one kernel launch per step."""

    def __init__(self, params, flat: Optional[FlatParams], lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        for g in self.param_groups:
            g.setdefault("lr_scale", 1.0)
        if len(self.param_groups) > 64:
            raise ValueError("unite_adamw_flat supports at most 64 parameter groups")
        self._flat = flat
        self._step = 0
        self._ready = False
        self._unused = ()
        self._step_params = None

    def set_unused(self, prefixes):
        """parameters (by name prefix) that receive no gradient in this training stage: torch.optim.AdamW skips p.grad is None
        entirely (no weight decay, no moment update) -- stage 3 never back-propagates through clip_decoder.*"""
        prefixes = tuple(prefixes)
        if prefixes != self._unused:
            self._unused = prefixes
            if self._ready:
                self._chunk_group = self._build_chunk_table()

    def _effective_unused(self):
        """parameters without a gradient this step, as (name PREFIXES, exact NAMES): prefixes are what the engine declared (set_unused) and what the
        model's runtime reports (layers its last forward did not execute, e.g. blocks above the highest tap under clip_only); exact names are the
        parameters whose requires_grad was switched off AFTER the optimizer was built (run_stage2.py:711-746 freezes layers behind
        create_optimizer: autograd then leaves their p.grad at None and torch.optim.AdamW skips them -- no moment update, no weight decay;
        --lp_ft_epochs switches them back on).  Frozen names are matched exactly, never as prefixes: freezing `head.weight` must not freeze a
        `head.weight_g` beside it."""
        frozen = frozenset(n for n, p in zip(self._flat.names, self._flat.params) if not p.requires_grad) if self._flat is not None else frozenset()
        return tuple(self._unused) + tuple(getattr(self._flat, "unused_prefixes", ()) or ()), frozen

    def no_grad_chunks(self):
        """(chunk -> group table, id of the group without gradients) for the masked gradient norm; (None, -1) if every parameter has one"""
        if not self._ready:
            self._prepare()
        self._refresh_unused()
        return (self._chunk_group, self._frozen_group) if self._frozen_group is not None else (None, -1)

    def _refresh_unused(self):
        eff = self._effective_unused()
        if eff != getattr(self, "_table_unused", None):
            self._chunk_group = self._build_chunk_table()

    def _build_chunk_table(self):
        fp = self._flat
        self._table_unused = self._effective_unused()
        name_of = {id(p): n for n, p in zip(fp.names, fp.params)}
        group_of = {}
        for gi, g in enumerate(self.param_groups):
            for p in g["params"]:
                group_of[name_of[id(p)]] = gi
        prefixes, frozen = self._table_unused
        missing = [n for n in fp.names if n not in group_of or n in frozen or n.startswith(prefixes or ("\0",))]
        if missing:
            # parameters outside the optimizer (frozen / filtered / without gradient): a group the kernel skips (lr < 0)
            if len(self.param_groups) >= 64:
                raise ValueError("no spare group for parameters that are not optimised")
            self._frozen_group = len(self.param_groups)
            for n in missing:
                group_of[n] = self._frozen_group
        else:
            self._frozen_group = None
        return fp.chunk_groups(group_of)

    def attach(self, flat: FlatParams):
        self._flat = flat

    def _prepare(self):
        fp = self._flat
        if fp is None:
            raise RuntimeError("FusedAdamW is not attached to a flat parameter buffer (call model.runtime() first)")
        self._chunk_group = self._build_chunk_table()
        self.exp_avg = torch.zeros_like(fp.param)
        self.exp_avg_sq = torch.zeros_like(fp.param)
        self._ready = True

    def use_step_params(self, params):
        """graph_step.StepParams: lr / wd / bias corrections are then read from device memory by the kernel (a captured step() launch stays
        valid while the schedule moves); the caller stages them with stage_hparams() before every launch"""
        self._step_params = params

    def _hparams(self):
        lrs = [float(g["lr"]) for g in self.param_groups]
        wds = [float(g["weight_decay"]) for g in self.param_groups]
        if self._frozen_group is not None:
            lrs.append(-1.0)
            wds.append(0.0)
        return lrs, wds

    def stage_hparams(self, params):
        """advance the step count and put this step's group table and Adam bias corrections into `params` (host side)"""
        if not self._ready:
            self._prepare()
        self._refresh_unused()
        self._step += 1
        params.lrs, params.wds = self._hparams()
        # exactly what unite_adamw_flat computes on the host: the betas rounded to f32 first, the powers in double
        import numpy as np
        b1, b2 = (float(np.float32(b)) for b in self.param_groups[0]["betas"])
        params.inv_bc = (1.0 / (1.0 - b1 ** self._step), 1.0 / math.sqrt(1.0 - b2 ** self._step))

    @torch.no_grad()
    def step(self, closure=None, grad_scale: Optional[torch.Tensor] = None, found_inf: Optional[torch.Tensor] = None):
        self._check_usable()
        if not self._ready:
            self._prepare()
        if getattr(self, "_open", None) is not None:      # begin_step() ... step_range() ...: update what is left, close the step
            self._finish_open_step(grad_scale)
            return
        fp = self._flat
        b1, b2 = self.param_groups[0]["betas"]
        if self._step_params is not None:
            ops.adamw_flat_dev(fp.param, fp.grad, self.exp_avg, self.exp_avg_sq, fp.shadow, self._chunk_group, self._step_params.hp,
                               float(b1), float(b2), float(self.param_groups[0]["eps"]), grad_scale=grad_scale, found_inf=found_inf)
            return
        self._refresh_unused()
        self._step += 1
        lrs, wds = self._hparams()
        ops.adamw_flat(fp.param, fp.grad, self.exp_avg, self.exp_avg_sq, fp.shadow, self._chunk_group, lrs, wds,
                       float(b1), float(b2), float(self.param_groups[0]["eps"]), self._step, grad_scale=grad_scale, found_inf=found_inf)

    # ---- AdamW per gradient bucket (opt-in: UNITE_BUCKET_ADAMW=1 / NativeScalerWithGradNormCount.bucket_adamw)
    # No shipped config clips gradients, so a layer's update needs nothing but its own (reduced) gradient: ``begin_step()`` fixes this
    # step's count and hyper-parameters, ``step_range(lo, hi)`` updates one 1024-aligned range of the flat buffer (the reducer calls it
    # behind each bucket's all-reduce, on the reducer's stream: the optimizer then overlaps the remaining backward and communication),
    # and the closing ``step()`` covers whatever no bucket touched.  Element for element the arithmetic is that of the single launch.
    @torch.no_grad()
    def begin_step(self):
        if self._step_params is not None:
            raise RuntimeError("per-bucket AdamW and the captured (device-parameter) step are mutually exclusive")
        self._check_usable()
        if not self._ready:
            self._prepare()
        self._refresh_unused()
        self._step += 1
        self._open = dict(hp=self._hparams(), done=[])

    @torch.no_grad()
    def step_range(self, lo: int, hi: int, grad_scale: Optional[torch.Tensor] = None):
        from .flat_params import CHUNK
        if getattr(self, "_open", None) is None:
            raise RuntimeError("step_range() outside begin_step() ... step()")
        if lo % CHUNK or hi % CHUNK or not 0 <= lo < hi <= self._flat.total:
            raise ValueError("ranges are whole 1024-element chunks of the flat buffer")
        fp = self._flat
        b1, b2 = self.param_groups[0]["betas"]
        lrs, wds = self._open["hp"]
        ops.adamw_flat(fp.param[lo:hi], fp.grad[lo:hi], self.exp_avg[lo:hi], self.exp_avg_sq[lo:hi], fp.shadow[lo:hi],
                       self._chunk_group[lo // CHUNK:hi // CHUNK], lrs, wds, float(b1), float(b2), float(self.param_groups[0]["eps"]),
                       self._step, grad_scale=grad_scale)
        self._open["done"].append((lo, hi))

    def abort_step(self):
        """drop a step opened by begin_step() whose backward did not complete.  If no range had been updated yet the optimizer is as it was
        (the step count goes back).  If some ranges HAD been updated, parameters of different layers are now one optimisation step apart -- and
        differ from the other ranks' -- and nothing here can undo that: the optimizer refuses every further step (RuntimeError from begin_step /
        step), so a caller that swallows the backward's exception and carries on cannot train an inconsistent model; resume from a checkpoint."""
        if getattr(self, "_open", None) is not None:
            done = len(self._open["done"])
            self._open = None
            self._step -= 1
            if done:
                self._broken = f"a per-bucket AdamW step was aborted after {done} of its ranges had been updated"

    def _check_usable(self):
        if getattr(self, "_broken", None):
            raise RuntimeError(self._broken + ": the parameters are inconsistent (and differ between ranks); restore a checkpoint")

    def _finish_open_step(self, grad_scale=None):
        done = sorted(self._open["done"])
        pos = 0
        for lo, hi in done + [(self._flat.total, self._flat.total)]:
            if lo > pos:
                self.step_range(pos, lo, grad_scale)
            pos = max(pos, hi)
        self._open = None

    def zero_grad(self, set_to_none: bool = True):
        """The next backward overwrites the flat gradient buffer instead of adding to it: no 352 MB memset."""
        if self._flat is not None:
            self._flat.accumulate = False
            self._flat.ensure_grad_views()

    # checkpoint layout compatible with torch.optim.AdamW ({'state': {idx: {step, exp_avg, exp_avg_sq}}, 'param_groups'})
    def state_dict(self):
        if not self._ready:
            self._prepare()
        fp = self._flat
        pid, state, groups = {}, {}, []
        i = 0
        for g in self.param_groups:
            ids = []
            for p in g["params"]:
                pid[id(p)] = i
                ids.append(i)
                i += 1
            groups.append({**{k: v for k, v in g.items() if k != "params"}, "params": ids})
        name_of = {id(p): n for n, p in zip(fp.names, fp.params)}
        for g in self.param_groups:
            for p in g["params"]:
                o, k = fp.offsets[name_of[id(p)]]
                state[pid[id(p)]] = {"step": torch.tensor(float(self._step)),
                                     "exp_avg": self.exp_avg[o:o + k].view(p.shape).clone(),
                                     "exp_avg_sq": self.exp_avg_sq[o:o + k].view(p.shape).clone()}
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        if not self._ready:
            self._prepare()
        fp = self._flat
        name_of = {id(p): n for n, p in zip(fp.names, fp.params)}
        i = 0
        for g, sg in zip(self.param_groups, sd["param_groups"]):
            for k, v in sg.items():
                if k != "params":
                    g[k] = v
            for p in g["params"]:
                st = sd["state"].get(i)
                if st is not None:
                    o, k = fp.offsets[name_of[id(p)]]
                    self.exp_avg[o:o + k].view(p.shape).copy_(st["exp_avg"])
                    self.exp_avg_sq[o:o + k].view(p.shape).copy_(st["exp_avg_sq"])
                    self._step = int(float(st["step"]))
                i += 1


def create_optimizer(args, model, get_num_layer=None, get_layer_scale=None, filter_bias_and_bn=True, skip_list=None):
    """reference optim_factory.py:121-211, AdamW/Adam branches."""
    opt_lower = args.opt.lower()
    weight_decay = args.weight_decay
    if weight_decay and filter_bias_and_bn:
        skip = {}
        if skip_list is not None:
            skip = skip_list
        elif hasattr(model, 'no_weight_decay'):
            skip = model.no_weight_decay()
        parameters = get_parameter_groups(model, weight_decay, skip, get_num_layer, get_layer_scale)
        weight_decay = 0.
    else:
        parameters = [p for p in model.parameters() if p.requires_grad]
    opt_args = dict(lr=args.lr, weight_decay=weight_decay)
    if getattr(args, 'opt_eps', None) is not None:
        opt_args['eps'] = args.opt_eps
    if getattr(args, 'opt_betas', None) is not None:
        opt_args['betas'] = args.opt_betas
    print("optimizer settings:", opt_args)
    name = opt_lower.split('_')[-1]
    if name not in ("adamw", "fusedadamw"):
        raise NotImplementedError(f"optimizer '{args.opt}': only the AdamW branch of the reference factory is built "
                                  "(all UNITE configs use opt: adamw)")
    flat = model.runtime().fp if hasattr(model, "runtime") else None
    return FusedAdamW(parameters, flat, **opt_args)
