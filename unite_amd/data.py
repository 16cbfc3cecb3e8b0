"""GPU side of the input path: decoded uint8 frames -> the normalised (B,3,T,H,W) clip tensor of the engines.

The reference does flip + HWC->CHW transpose + /255 + normalise per clip in the DataLoader workers
(src/datasets/build.py:34-54 with transforms.py:68-96,209-245, mae.py:218-219); at > 1 000 clips/s per GPU that is
~60 M pixels/s per worker-second it cannot keep up with.  Workers here only decode / crop / resize to uint8 (T,H,W,3); the rest is
one HBM-bound kernel on the device, bit-identical to the CPU pipeline.
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch

from . import ops

IMAGENET_DEFAULT_MEAN = (0.485, 0.456, 0.406)        # build.py:34-35
IMAGENET_DEFAULT_STD = (0.229, 0.224, 0.225)


class ClipToTensor:
    """callable: uint8 (B,T,H,W,3) on the device (+ optional per-clip flip flags) -> f32 (B,3,T,H,W); output buffer reused."""

    def __init__(self, mean: Sequence[float] = IMAGENET_DEFAULT_MEAN, std: Sequence[float] = IMAGENET_DEFAULT_STD, flip_prob: float = 0.0,
                 seed: int = 0):
        self.mean, self.std, self.flip_prob = tuple(mean), tuple(std), float(flip_prob)
        self._gen: Optional[torch.Generator] = None
        self._seed = seed
        self._out = None

    def __call__(self, frames: torch.Tensor, flip: Optional[torch.Tensor] = None) -> torch.Tensor:
        if frames.device.type != "cuda":
            raise RuntimeError("ClipToTensor runs on the MI355X only (no CPU path): move the uint8 frames to 'cuda' first")
        B, T, H, W, _ = frames.shape
        if flip is None and self.flip_prob > 0:              # GroupRandomHorizontalFlip: one draw per clip, v < 0.5 flips
            if self._gen is None:
                self._gen = torch.Generator(device=frames.device)
                self._gen.manual_seed(self._seed)
            flip = (torch.rand(B, device=frames.device, generator=self._gen) < self.flip_prob).to(torch.uint8)
        if self._out is None or tuple(self._out.shape) != (B, 3, T, H, W):
            self._out = torch.empty(B, 3, T, H, W, dtype=torch.float32, device=frames.device)
        return ops.clip_u8_to_f32(frames.contiguous(), self._out, self.mean, self.std, flip)


class DistributedSampler(torch.utils.data.Sampler):
    """Which samples a rank sees in an epoch: the reference's sampler (src/datasets/distributed.py:81-163), i.e. torch's
    ``DistributedSampler`` plus ``repetitions`` -- the epoch is `repetitions` independent permutations of the dataset laid end to
    end (all drawn from one generator seeded ``seed + epoch``), padded by wrapping around (or truncated with ``drop_last``) to a
    multiple of the world size and dealt out round-robin: rank r takes positions r, r + world, ...

    Same constructor arguments, ``set_epoch`` / ``__len__`` / ``__iter__`` behaviour and index streams as the reference
    (tests/golden/sampler.json holds the reference's own output)."""

    def __init__(self, dataset, num_replicas: Optional[int] = None, rank: Optional[int] = None, shuffle: bool = True, seed: int = 0,
                 drop_last: bool = False, repetitions: int = 1) -> None:
        import torch.distributed as dist
        if num_replicas is None or rank is None:
            if not (dist.is_available() and dist.is_initialized()):
                raise RuntimeError("DistributedSampler needs num_replicas and rank, or an initialised process group to read them from")
            num_replicas = dist.get_world_size() if num_replicas is None else num_replicas
            rank = dist.get_rank() if rank is None else rank
        if not 0 <= rank < num_replicas:
            raise ValueError(f"Invalid rank {rank}, rank should be in the interval [0, {num_replicas - 1}]")
        self.dataset, self.num_replicas, self.rank = dataset, int(num_replicas), int(rank)
        self.shuffle, self.seed, self.drop_last, self.num_repetitions = shuffle, seed, drop_last, int(repetitions)
        self.epoch = 0
        total = len(dataset) * self.num_repetitions
        if drop_last and total % self.num_replicas:
            self.num_samples = total // self.num_replicas          # = ceil((total - world) / world) when world does not divide total
        else:
            self.num_samples = -(-total // self.num_replicas)
        self.total_size = self.num_samples * self.num_replicas

    def _epoch_order(self):
        n = len(self.dataset)
        if not self.shuffle:
            return list(range(n)) * self.num_repetitions
        g = torch.Generator()
        g.manual_seed(self.seed + self.epoch)
        order = []
        for _ in range(self.num_repetitions):
            order += torch.randperm(n, generator=g).tolist()
        return order

    def __iter__(self):
        order = self._epoch_order()
        if self.drop_last:
            order = order[:self.total_size]
        else:
            short = self.total_size - len(order)
            if short > 0:
                order = (order * (1 + -(-short // len(order))))[:self.total_size]      # wrap around as often as needed
        assert len(order) == self.total_size
        mine = order[self.rank:self.total_size:self.num_replicas]
        assert len(mine) == self.num_samples
        return iter(mine)

    def __len__(self) -> int:
        return self.num_samples

    def set_epoch(self, epoch: int) -> None:
        """call before building each epoch's DataLoader iterator: the permutation is a function of seed + epoch"""
        self.epoch = epoch


def sample_train_indices(num_frames: int, num_segments: int, skip_length: int = 1, new_step: int = 1, temporal_jitter: bool = False,
                         rng=None):
    """Sparse (TSN-style) frame sampling of a training clip: one random frame per equal-length segment of the video, 1-based
    segment offsets + per-step jitter offsets (reference src/datasets/mae.py:253-273, drawing from numpy's global generator in
    the same order: segment offsets first, then the jitter).  `rng`: a numpy RandomState / module with ``randint`` (default
    ``numpy.random``).  PARITY UNPINNED by the reference (its module imports decord and cv2, absent here): pinned by the properties
    in tests/test_host_logic.py only."""
    import numpy as np
    rng = np.random if rng is None else rng
    seg_len = (num_frames - skip_length + 1) // num_segments
    if seg_len > 0:
        offsets = np.arange(num_segments) * seg_len + rng.randint(seg_len, size=num_segments)
    elif num_frames > max(num_segments, skip_length):
        offsets = np.sort(rng.randint(num_frames - skip_length + 1, size=num_segments))
    else:
        offsets = np.zeros((num_segments,))
    n_steps = skip_length // new_step
    skip_offsets = rng.randint(new_step, size=n_steps) if temporal_jitter else np.zeros(n_steps, dtype=int)
    return offsets + 1, skip_offsets


def frame_id_list(duration: int, indices, skip_offsets, skip_length: int = 1, new_step: int = 1):
    """0-based frame numbers to decode for the sampled segment offsets (reference src/datasets/mae.py:275-287): `skip_length /
    new_step` frames per segment, `new_step` apart, clamped so that no frame lies beyond the video.  PARITY UNPINNED, as above."""
    out = []
    for seg in indices:
        offset = int(seg)
        for i in range(len(range(0, skip_length, new_step))):
            out.append(offset + int(skip_offsets[i]) - 1 if offset + skip_offsets[i] <= duration else offset - 1)
            if offset + new_step < duration:
                offset += new_step
    return out
