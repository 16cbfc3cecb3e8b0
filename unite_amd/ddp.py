"""Data parallelism for the flat-gradient models: one process per GPU, gradient all-reduce (mean) over RCCL / xGMI.

Replaces torch.nn.parallel.DistributedDataParallel as the reference uses it (run_stage1.py:809): parameters are
broadcast from rank 0 once, and during backward each *bucket* -- a contiguous slice of the flat fp32 gradient
buffer covering whole layers -- is all-reduced on a side HIP stream as soon as the last layer in it has written its
gradients, so communication overlaps the remaining backward.  There are no packing copies (the buckets ARE the
gradient storage) and no autograd hooks (the hand-scheduled backward reports layer completion itself).
xGMI is point-to-point (7 links x ~153 GB/s per GPU): a ring all-reduce is per-link bound, so buckets are large
(default 64 MiB over whole layers -> 5 collectives per step for ViT-B) rather than NCCL's 25 MiB default; the layers that complete
last (block 0 + patch embed, 29 MiB) form a bucket of their own, the only all-reduce that has no backward work left to hide under.

``GradReducer`` is device-agnostic (it only needs a flat tensor, tag -> range table and a process group), which is
how the N > 1 path is tested with gloo on CPU.
"""
from __future__ import annotations

import contextlib
import ctypes
import os
import sys
import time
from typing import Dict, Hashable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist
import torch.nn as nn

_COMM_TIMING = os.environ.get("UNITE_COMM_TIMING", "0") == "1"
_COMM_TIMES: List[float] = []


def native_comm(group=None):
    """The process-wide RCCL communicator behind include/unite_comm.h, created on first use: rank 0 draws the rendezvous id and the
    launcher's process group carries it to the other ranks (one object broadcast).  Returns the bound library."""
    from . import _lib
    lib = _lib.load_comm()
    if lib.unite_comm_world() == 0:
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        buf = ctypes.create_string_buffer(_lib.COMM_ID_BYTES)
        if rank == 0:
            _lib.check_comm(lib.unite_comm_unique_id(buf, _lib.COMM_ID_BYTES), "unite_comm_unique_id")
        box = [buf.raw]
        dist.broadcast_object_list(box, src=0, group=group)
        _lib.check_comm(lib.unite_comm_init(rank, world, box[0], _lib.COMM_ID_BYTES), "unite_comm_init")
    return lib


class GradReducer:
    def __init__(self, flat_grad: torch.Tensor, tag_ranges: Sequence[Tuple[Hashable, int, int]], bucket_bytes: int = 64 << 20,
                 group=None, tail_bytes: int = 32 << 20):
        """tag_ranges: (tag, start, end) element ranges in BACKWARD COMPLETION order; consecutive tags must be adjacent in
        memory (descending addresses) for them to share a bucket."""
        self.grad = flat_grad
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # UNITE_DDP_FORCE_COLLECTIVES=1: issue the bucket collectives even in a one-rank group (a one-GPU box can then run the whole RCCL path --
        # side stream, completion events, ncclAvg all-reduce per bucket, join -- as a rehearsal; the values are unchanged by a one-rank mean)
        self.multi = self.world > 1 or (dist.is_initialized() and os.environ.get("UNITE_DDP_FORCE_COLLECTIVES", "0") == "1")
        # The all-reduce of whatever completes LAST cannot hide under any backward work (nothing is left to compute): the layers
        # that finish last (at most `tail_bytes` of gradients, at least one layer) get a bucket of their own, so that everything
        # before them is already being reduced while they are still in their backward.
        parts = list(tag_ranges)
        n_tail, size = 0, 0
        if tail_bytes and len(parts) > 1:
            for tag, lo, hi in reversed(parts):
                if n_tail and size + (hi - lo) * 4 > tail_bytes:
                    break
                n_tail, size = n_tail + 1, size + (hi - lo) * 4
            if n_tail == len(parts):
                n_tail = 0
        head, tail = parts[:len(parts) - n_tail], parts[len(parts) - n_tail:]
        self.buckets: List[dict] = []

        def pack(seq, cap):
            cur = None
            for tag, lo, hi in seq:
                if cur is not None and (lo == cur["hi"] or hi == cur["lo"]) and (cur["hi"] - cur["lo"]) * 4 < cap:
                    cur["lo"], cur["hi"] = min(cur["lo"], lo), max(cur["hi"], hi)
                    cur["tags"].add(tag)
                    cur["parts"].append((tag, lo, hi))
                else:
                    cur = dict(lo=lo, hi=hi, tags={tag}, parts=[(tag, lo, hi)])
                    self.buckets.append(cur)
        pack(head, bucket_bytes)
        pack(tail, 1 << 62)
        self.tag_bucket: Dict[Hashable, int] = {t: i for i, b in enumerate(self.buckets) for t in b["tags"]}
        self.use_stream = flat_grad.is_cuda
        self.stream = torch.cuda.Stream(device=flat_grad.device) if self.use_stream else None
        # RCCL ('nccl') averages in the collective; gloo (CPU tests, single-GPU rehearsal) has no AVG: sum, then scale
        self.native_avg = dist.is_initialized() and dist.get_backend(group) == "nccl"
        # UNITE_COMM_NATIVE=1: the buckets go through libunite_comm.so (unite_comm_allreduce_bucket, include/unite_comm.h) instead of
        # torch.distributed's nccl backend -- the same RCCL underneath, without the process-group layer per collective.  Opt-in: its
        # multi-rank path has not run on hardware yet (one GPU per box here; tests/test_comm_gpu.py covers the one-rank communicator).
        self.comm = None
        if self.use_stream and self.multi and self.native_avg and os.environ.get("UNITE_COMM_NATIVE", "0") == "1":
            self.comm = native_comm(group)
        self._pending: List[set] = []
        self._events: List[list] = []
        self._works = []
        # False inside no_sync(): a micro-batch of a gradient-accumulation step only ADDS to the flat buffer; nothing is reduced
        # until the backward of the last micro-batch (reference: update_freq, engine_for_finetuning.py:83-84 -- there DDP
        # all-reduces every micro-batch; reducing the accumulated sum once is the same mean at 1 / update_freq of the traffic)
        self.enabled = True
        self.launched = 0                # collectives issued since construction (tests / diagnostics)
        # optional ``after_bucket(lo, hi)``: called once per bucket when its reduced (averaged) gradients are in place -- on the reducer's
        # stream right behind the collective (RCCL), or at finish() (gloo).  FusedAdamW.step_range hangs here (UNITE_BUCKET_ADAMW=1).
        self.after_bucket = None
        self.reset()

    def reset(self):
        self._pending = [set(b["tags"]) for b in self.buckets]
        self._events = [[] for _ in self.buckets]
        self._works = []

    @contextlib.contextmanager
    def no_sync(self):
        old, self.enabled = self.enabled, False
        try:
            yield
        finally:
            self.enabled = old

    def layer_done(self, tag, events=None):
        """called by the backward when every gradient of `tag` has been written: by default on the current stream; `events`
        (HIP events, one per stream that wrote gradients of the layer) when its weight gradients ran on a side stream -- the
        collective then waits for exactly those events instead of for whatever the main stream does next."""
        if not self.enabled:
            return
        i = self.tag_bucket.get(tag)
        if i is None:
            return
        if self.use_stream and self.multi:
            if events is None:
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream())
                events = (ev,)
            self._events[i].extend(events)
        self._pending[i].discard(tag)
        if not self._pending[i]:
            self._launch(i)

    def _launch(self, i):
        if not self.multi:
            if self.after_bucket is not None:      # one rank: nothing to reduce, the bucket is final as it stands
                self.after_bucket(self.buckets[i]["lo"], self.buckets[i]["hi"])
            return
        b = self.buckets[i]
        view = self.grad[b["lo"]:b["hi"]]
        self.launched += 1
        if self.use_stream and self.native_avg:
            for ev in self._events[i]:
                self.stream.wait_event(ev)
            if self.comm is not None:
                from . import _lib
                t0 = time.perf_counter() if _COMM_TIMING else 0.0
                _lib.check_comm(self.comm.unite_comm_allreduce_bucket(view.data_ptr(), view.numel(), 0, 1, self.stream.cuda_stream),
                                "unite_comm_allreduce_bucket")
                if _COMM_TIMING:                   # diagnostic: how long the host thread (autograd's backward thread) sits inside the RCCL call
                    _COMM_TIMES.append((time.perf_counter() - t0) * 1e3)
                    if len(_COMM_TIMES) % 100 == 0:
                        last = _COMM_TIMES[-100:]
                        print(f"[unite_amd.ddp] native all-reduce enqueue: mean {sum(last) / 100:.3f} ms, max {max(last):.3f} ms over the last 100 calls",
                              file=sys.stderr, flush=True)
            else:
                with torch.cuda.stream(self.stream):
                    dist.all_reduce(view, op=dist.ReduceOp.AVG, group=self.group)
            if self.after_bucket is not None:
                with torch.cuda.stream(self.stream):
                    self.after_bucket(b["lo"], b["hi"])
        else:
            if self.use_stream:          # gloo on device tensors stages through the host from the CURRENT stream
                for ev in self._events[i]:
                    torch.cuda.current_stream().wait_event(ev)
            w = dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True)   # gloo has no AVG
            self._works.append((w, view, (b["lo"], b["hi"])))
        self._events[i] = []

    def finish(self):
        """join: every bucket reduced and visible to the current stream (call before grad-norm / optimizer)."""
        if not self.enabled:
            self.reset()
            return
        for i, p in enumerate(self._pending):
            if p:            # a layer never reported (e.g. unused parameters): reduce what is there
                self._pending[i] = set()
                if self.use_stream and self.multi:
                    ev = torch.cuda.Event()
                    ev.record(torch.cuda.current_stream())
                    self._events[i].append(ev)
                self._launch(i)
        if self.use_stream and self.native_avg:
            torch.cuda.current_stream().wait_stream(self.stream)
        else:
            for w, view, rng in self._works:
                w.wait()
                view.div_(self.world)
                if self.after_bucket is not None:
                    self.after_bucket(*rng)
        self.reset()


def student_tag_ranges(rt) -> List[Tuple[Hashable, int, int]]:
    """Backward completion order of the stage-1/3 student's parameter layers with their flat ranges
    (= _StudentRuntime.tag_ranges(); kept as a function for callers that hold a runtime)."""
    fp = rt.fp
    tags: List[Tuple[Hashable, str]] = [("clip_decoder", "clip_decoder.")]
    depth = rt.depth
    lo_tap = min(rt.taps)
    for i in reversed(range(depth)):
        tags.append((i, f"encoder.blocks.{i}."))
    tags.append(("patch_embed", "encoder.patch_embed."))
    ranges = fp.layer_ranges([p for _, p in tags])
    out = [(t, lo, hi) for (t, _), (lo, hi) in zip(tags, ranges)]
    # encoder.norm sits between blocks.{depth-1} and clip_decoder in memory but only completes with the lowest tap:
    # it rides with the bucket of block `lo_tap + 1` by being reported then (see _StudentRuntime.backward_from_dy).
    (nlo, nhi), = fp.layer_ranges(["encoder.norm."])
    out.insert(1, ("norm", nlo, nhi))
    return out


class DistributedDataParallel(nn.Module):
    """model wrapper with the reference's attribute surface (.module, forward passthrough)."""

    def __init__(self, module: nn.Module, device_ids=None, find_unused_parameters: bool = False, bucket_cap_mb: int = 64,
                 process_group=None):
        super().__init__()
        self.module = module
        rt = module.runtime()
        self.rt = rt
        forced = os.environ.get("UNITE_DDP_FORCE_COLLECTIVES", "0") == "1"         # one-rank rehearsal of the collective path (GradReducer)
        if dist.is_initialized() and (dist.get_world_size(process_group) > 1 or forced):
            dist.broadcast(rt.fp.param, src=0, group=process_group)      # run_stage1.py:809 broadcasts rank 0's weights
            rt.fp.sync_shadow()
        # every runtime lists its own layers in backward-completion order (stage 1/3 student: decoders, norm, blocks, patch
        # embed; stage-2 classifier: head + fc_norm, blocks, patch embed)
        self.reducer = GradReducer(rt.fp.grad, rt.tag_ranges(), bucket_cap_mb << 20, process_group)
        rt.layer_done_hook = self.reducer.layer_done

    def no_sync(self):
        """torch DDP's context manager: backward passes inside only accumulate into the flat gradient buffer."""
        return self.reducer.no_sync()

    def forward(self, *a, **k):
        return self.module(*a, **k)

    def forward_loss(self, *a, **k):
        return self.module.forward_loss(*a, **k)
