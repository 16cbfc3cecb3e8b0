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
    """callable: uint8 (B,T,H,W,3) on the device (+ optional per-clip flip flags) -> f32 (B,3,T,H,W).  Every call returns a NEW tensor
    unless ``reuse_output``: with the teacher one batch ahead the following batch is transformed (on the teacher's stream) while the
    student may still be reading this one, so a loader must not hand out the same buffer twice."""

    def __init__(self, mean: Sequence[float] = IMAGENET_DEFAULT_MEAN, std: Sequence[float] = IMAGENET_DEFAULT_STD, flip_prob: float = 0.0,
                 seed: int = 0, reuse_output: bool = False):
        self.reuse_output = reuse_output
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
        if not self.reuse_output:
            return ops.clip_u8_to_f32(frames.contiguous(), torch.empty(B, 3, T, H, W, dtype=torch.float32, device=frames.device), self.mean,
                                      self.std, flip)
        if self._out is None or tuple(self._out.shape) != (B, 3, T, H, W):
            self._out = torch.empty(B, 3, T, H, W, dtype=torch.float32, device=frames.device)
        return ops.clip_u8_to_f32(frames.contiguous(), self._out, self.mean, self.std, flip)


class MultiScaleCrop:
    """Which box of a clip the training transform keeps (reference GroupMultiScaleCrop._sample_crop_size, src/datasets/transforms.py:154-205;
    tests/golden/sampling.json holds that method's own draws): side lengths are the short side of the frame times one of ``scales``
    (snapped to the network's input size when within 3 pixels), width and height at most ``max_distort`` scale steps apart, the position
    one of 5 (13 with ``more_fix_crop``) anchor points of a 4 x 4 grid of the slack, or uniform without ``fix_crop``.  Returns
    (x0, y0, w, h) -- the order unite_crop_resize_u8 takes; ``rng`` is a ``random.Random``-like object (default: the ``random`` module)."""

    def __init__(self, input_size, scales=(1, .875, .75, .66), max_distort=1, fix_crop=True, more_fix_crop=True):
        self.input_size = (input_size, input_size) if isinstance(input_size, int) else tuple(input_size)
        self.scales, self.max_distort, self.fix_crop, self.more_fix_crop = tuple(scales), max_distort, fix_crop, more_fix_crop

    def anchors(self, slack_w: int, slack_h: int):
        qw, qh = slack_w // 4, slack_h // 4
        grid = [(0, 0), (4, 0), (0, 4), (4, 4), (2, 2)]                               # corners, centre
        if self.more_fix_crop:
            grid += [(0, 2), (4, 2), (2, 4), (2, 0), (1, 1), (3, 1), (1, 3), (3, 3)]  # edge centres, quarter points
        return [(i * qw, j * qh) for i, j in grid]

    def __call__(self, im_w: int, im_h: int, rng=None):
        import random as _random
        rng = _random if rng is None else rng
        short = min(im_w, im_h)
        sides = [int(short * s) for s in self.scales]
        ws = [self.input_size[0] if abs(v - self.input_size[0]) < 3 else v for v in sides]
        hs = [self.input_size[1] if abs(v - self.input_size[1]) < 3 else v for v in sides]
        w, h = rng.choice([(ws[j], hs[i]) for i in range(len(hs)) for j in range(len(ws)) if abs(i - j) <= self.max_distort])
        if self.fix_crop:
            x0, y0 = rng.choice(self.anchors(im_w - w, im_h - h))
        else:
            x0 = rng.randint(0, im_w - w)
            y0 = rng.randint(0, im_h - h)
        return x0, y0, w, h


class GpuTrainTransform:
    """The training transform of build.py:34-54 (GroupMultiScaleCrop -> GroupRandomHorizontalFlip -> Stack -> ToTorchFormatTensor ->
    GroupNormalize) on the device: decoded uint8 frames (B,T,H,W,3) in, the engines' f32 (B,3,T,S,S) clip out.  One crop box per clip
    (drawn on the host like the reference does), unite_crop_resize_u8 (Pillow-bilinear arithmetic, bit for bit), then ClipToTensor."""

    def __init__(self, input_size: int, mean=IMAGENET_DEFAULT_MEAN, std=IMAGENET_DEFAULT_STD, flip_prob: float = 0.0, seed: int = 0):
        self.size, self.crop = int(input_size), MultiScaleCrop(input_size)
        self.to_tensor = ClipToTensor(mean, std, flip_prob, seed)
        self._u8, self._ws = None, None

    def __call__(self, frames: torch.Tensor, boxes=None, flip: Optional[torch.Tensor] = None, rng=None) -> torch.Tensor:
        if frames.device.type != "cuda":
            raise RuntimeError("GpuTrainTransform runs on the MI355X only (no CPU path): move the uint8 frames to 'cuda' first")
        B, T, H, W, _ = frames.shape
        if boxes is None:
            boxes = [self.crop(W, H, rng) for _ in range(B)]
        if self._u8 is None or tuple(self._u8.shape) != (B, T, self.size, self.size, 3):
            self._u8 = torch.empty(B, T, self.size, self.size, 3, dtype=torch.uint8, device=frames.device)
        need = ops.crop_resize_workspace(B, T, H, self.size, self.size)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=frames.device)
        ops.crop_resize_u8(frames.contiguous(), boxes, self._u8, self._ws)
        return self.to_tensor(self._u8, flip)


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


def get_seq_frames(video_size: int, num_frames: int, clip_idx: int = -1, skip_frames: int = 0, mode: str = "train",
                   test_num_segment: int = 1, rng=None):
    """Frame numbers of one clip of a video, as the reference's sparse dataset draws them (src/datasets/kinetics_sparse.py:283-312
    ``VideoClsDataset_sparse._get_seq_frames``; tests/golden/sampling.json holds that method's own output on seeded streams).
    Sparse strategy (skip_frames <= 0): the video is cut into ``num_frames`` equal segments; training (clip_idx == -1) takes one uniformly
    drawn frame of each segment (both ends included, so neighbours can share a frame), evaluation takes the frame a fixed fraction
    (clip_idx + 1) / (segments_of_the_view + 1) into each segment.  Skip strategy: ``num_frames`` frames ``skip_frames`` apart from a random
    start.  Indices never pass the last frame.  ``rng``: a ``random.Random``-like object (default: the ``random`` module, as the
    reference uses)."""
    import random as _random
    import numpy as np
    rng = _random if rng is None else rng
    last = int(video_size) - 1
    if skip_frames > 0:
        first = rng.randint(0, max(0, last - num_frames * skip_frames))
        return [min(first + k * skip_frames, last) for k in range(num_frames)]
    seg = max(0., float(video_size - 1) / num_frames)
    edges = [int(np.round(seg * i)) for i in range(num_frames + 1)]
    if clip_idx == -1:
        return [min(rng.randint(edges[i], edges[i + 1]), last) for i in range(num_frames)]
    views = test_num_segment if mode == 'test' else 1
    into = int(seg / (views + 1) * (clip_idx + 1))
    return [min(edges[i] + into, last) for i in range(num_frames)]


def sample_train_indices(num_frames: int, num_segments: int, skip_length: int = 1, new_step: int = 1, temporal_jitter: bool = False,
                         rng=None):
    """Sparse (TSN-style) frame sampling of a training clip: one random frame per equal-length segment of the video, 1-based
    segment offsets + per-step jitter offsets (reference src/datasets/mae.py:253-273, drawing from numpy's global generator in
    the same order: segment offsets first, then the jitter).  `rng`: a numpy RandomState / module with ``randint`` (default
    ``numpy.random``).  Pinned on the reference method's own output (tests/golden/sampling.json, oracle/make_golden_sampling.py)."""
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
    new_step` frames per segment, `new_step` apart, clamped so that no frame lies beyond the video.  Pinned as above."""
    out = []
    for seg in indices:
        offset = int(seg)
        for i in range(len(range(0, skip_length, new_step))):
            out.append(offset + int(skip_offsets[i]) - 1 if offset + skip_offsets[i] <= duration else offset - 1)
            if offset + new_step < duration:
                offset += new_step
    return out
