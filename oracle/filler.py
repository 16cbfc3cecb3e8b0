"""Deterministic weight / input filler shared by the golden generator and the tests.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Values depend only on (seed, key name, shape),
so the reference model (in oracle/make_golden.py) and the build's model (in tests/) can be given
identical full-size weights without committing a 350 MB state_dict.
"""
from __future__ import annotations

import zlib
from typing import Dict, Iterable, Tuple

import torch


def _gen(seed: int, name: str) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed((seed * 1000003 + zlib.crc32(name.encode())) & 0x7FFFFFFF)
    return g


def fill_tensor(name: str, shape: Tuple[int, ...], seed: int = 0) -> torch.Tensor:
    """uniform(-1,1) * scale(name, shape) (+1 for normalisation gains)."""
    u = torch.rand(tuple(shape), generator=_gen(seed, name), dtype=torch.float32) * 2.0 - 1.0
    leaf = name.split(".")[-1]
    parent = name.split(".")[-2] if "." in name else ""
    is_norm = parent.startswith("norm") or parent.startswith("ln_") or parent in ("fc_norm",)
    if is_norm and leaf == "weight":
        return 1.0 + 0.2 * u
    if is_norm and leaf == "bias":
        return 0.1 * u
    if name in ("class_embedding", "positional_embedding"):
        return u * (shape[-1] ** -0.5) * 1.7
    if name == "proj":                                  # clip.py:143, (width, output_dim)
        return u * (3.0 / shape[0]) ** 0.5
    if len(shape) == 1:                                 # biases, q_bias, v_bias
        return 0.1 * u
    fan_in = 1
    for s in shape[1:]:
        fan_in *= s
    return u * (3.0 / fan_in) ** 0.5 * 1.5              # var ~ 2.25/fan_in: keeps activations O(1)


def fill_state_dict(shapes: Iterable[Tuple[str, Tuple[int, ...]]], seed: int = 0) -> Dict[str, torch.Tensor]:
    return {k: fill_tensor(k, tuple(s), seed) for k, s in shapes}


def make_videos(B: int, T: int, H: int, W: int, seed: int = 0) -> torch.Tensor:
    """randn stands for ImageNet-normalised pixels."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    return torch.randn(B, 3, T, H, W, generator=g, dtype=torch.float32)


def make_importance(BT: int, N: int, seed: int = 0) -> torch.Tensor:
    """A per-row permutation standing in for torch.multinomial(attn, N) (run_stage1.py:382)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed + 17)
    return torch.stack([torch.randperm(N, generator=g) for _ in range(BT)])
