"""HIP-graph capture of the stage-1 training step.

Why: the eager step is ~600 kernel launches, events and stream waits issued from Python through ctypes -- 20 ms of host time per
23-ms step on MI355X (bench.py: host_enqueue_ms_per_step).  Shapes are static in training, so the whole step (teacher on three streams,
mask, targets, student forward, loss, hand-scheduled backward with the weight-gradient stream, gradient norm, AdamW) is captured ONCE
and replayed with a single hipGraphLaunch.  What changes from step to step lives in device memory and is rewritten by the host between
replays (``StepParams``): the learning rates / weight decays of the parameter groups (run_stage1.py:326-338 writes the schedule into
``optimizer.param_groups`` every step), Adam's bias corrections, and the seeds of the mask sampler and of stochastic depth.

Opt-in: ``GraphedStage1Step(model, teacher, optimizer, scaler, ...)`` is called directly (``bench.py --graph 1``, tests/test_model_gpu.py);
``train_one_epoch`` runs eager launches (the replay measured slower than eager on this ROCm) and data-parallel
runs keep the eager step (the bucket all-reduces are issued through torch.distributed, whose capture on RCCL is untested here).
"""
from __future__ import annotations

import math
from typing import Optional

import torch

from . import ops


class StepParams:
    """Per-step scalars in device memory: hp f32 [130] = lr[64] | wd[64] | 1/(1-b1^t) | 1/sqrt(1-b2^t); seeds int64 [2] = mask, drop-path.
    ``publish`` stages them in a ring of pinned host slots and enqueues two small asynchronous copies on the current stream."""
    RING = 32

    def __init__(self, device):
        self.hp = torch.zeros(130, dtype=torch.float32, device=device)
        self.seeds = torch.zeros(2, dtype=torch.int64, device=device)
        self._h_hp = torch.zeros(self.RING, 130, dtype=torch.float32).pin_memory()
        self._h_seeds = torch.zeros(self.RING, 2, dtype=torch.int64).pin_memory()
        self._ev = [None] * self.RING
        self._i = 0
        self.lrs, self.wds, self.inv_bc = [0.0], [0.0], (1.0, 1.0)
        self.seed_mask, self.seed_dp = 0, 0

    @property
    def seed_mask_dev(self):
        return self.seeds[0:1]

    @property
    def seed_dp_dev(self):
        return self.seeds[1:2]

    def publish(self):
        k = self._i % self.RING
        self._i += 1
        if self._ev[k] is not None:
            self._ev[k].synchronize()             # the copy that last read this slot (RING steps ago) has run: practically never waits
        h = self._h_hp[k]
        h.zero_()
        n = len(self.lrs)
        h[:n] = torch.tensor(self.lrs, dtype=torch.float32)
        h[64:64 + n] = torch.tensor(self.wds, dtype=torch.float32)
        h[128], h[129] = self.inv_bc
        self._h_seeds[k, 0] = _i64(self.seed_mask)
        self._h_seeds[k, 1] = _i64(self.seed_dp)
        self.hp.copy_(h, non_blocking=True)
        self.seeds.copy_(self._h_seeds[k], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._ev[k] = ev


def _i64(u: int) -> int:
    u &= 0xFFFFFFFFFFFFFFFF
    return u - (1 << 64) if u >= (1 << 63) else u


class GraphedStage1Step:
    """callable(videos) -> (loss, grad_norm) device scalars; same arithmetic as engine_stage1's eager step (mask_type 'attention',
    clip_loss_data 'mixed').  The first ``warmup`` calls run eagerly (buffers get allocated, kernels get their attributes set), the next
    one is captured, later ones are replays."""

    def __init__(self, model, teacher_model, optimizer, loss_scaler, batch_shape, mask_ratio: float, clip_grad: Optional[float] = None,
                 clip_input_resolution: int = 224, state=None, warmup: int = 2):
        from .engine_stage1 import StepState
        self.model, self.teacher, self.opt, self.scaler = model, teacher_model, optimizer, loss_scaler
        self.mask_ratio, self.clip_grad, self.res = mask_ratio, clip_grad, clip_input_resolution
        student = getattr(model, "module", model)
        if getattr(model, "reducer", None) is not None and getattr(model.reducer, "world", 1) > 1:
            raise NotImplementedError("the captured step is single-rank; data-parallel runs use the eager step")
        dev = next(student.parameters()).device
        self.params = StepParams(dev)
        self.state = state or StepState()
        self.state.step_params = self.params
        student.runtime().runner.step_params = self.params
        optimizer.use_step_params(self.params)
        self.videos = torch.empty(batch_shape, dtype=torch.float32, device=dev)
        self.graph = None
        self.calls, self.warmup = 0, warmup
        self._out = None
        self._ring = torch.zeros(256, 2, device=dev)
        self._slot = 0

    def _body(self):
        from .engine_stage1 import stage1_step
        B = self.videos.shape[0]
        loss = stage1_step(self.model, self.teacher, self.videos, B, self.mask_ratio, 'attention', None, 'mixed', self.state, self.res)
        self.opt.zero_grad()
        gn = self.scaler(loss, self.opt, clip_grad=self.clip_grad, parameters=None, create_graph=False, reducer=None)
        return loss, gn

    def _advance(self):
        """host side of a step: next seeds, optimizer step count, this step's lr / wd table -> device"""
        p, st = self.params, self.state
        st.seed += 1
        p.seed_mask = st.seed
        runner = getattr(self.model, "module", self.model).runtime().runner
        p.seed_dp = runner.next_drop_path_seed()
        self.opt.stage_hparams(p)
        p.publish()

    def __call__(self, videos):
        if videos.data_ptr() != self.videos.data_ptr():
            self.videos.copy_(videos, non_blocking=True)
        self._advance()
        if self.graph is None and self.calls < self.warmup:
            loss, gn = self._body()
        elif self.graph is None:
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._out = self._body()
            self.graph = g                        # capture does not execute: run the captured step once now
            g.replay()
            loss, gn = self._out
        else:
            self.graph.replay()
            loss, gn = self._out
        self.calls += 1
        # the graph writes the same two scalars every replay: hand out copies from a ring (the engines read them back later)
        self._slot = (self._slot + 1) % self._ring.shape[0]
        row = self._ring[self._slot]
        row[0].copy_(loss.detach())
        row[1].copy_(gn.detach())
        return row[0], row[1]
