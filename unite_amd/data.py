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
