#!/usr/bin/env python
"""CLIP checkpoint hand-off fixture: the REFERENCE's own ``inflate_weight`` and ``load_state_dict`` (src/models/clip.py:191-231,
importable as is) applied to a small OpenAI-style ``visual.*`` state_dict -- a 2-D conv1 weight that has to be inflated along time
(center and mean forms) and a position table that has to be bicubic-resized from a 3 x 3 to a 2 x 2 grid -> tests/golden/clip_ckpt.npz.
TEST INFRASTRUCTURE; runs only where /root/reference exists."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import make_golden as G  # noqa: E402


def main():
    clip_ref = G._load("ref_clip_standalone", "src/models/clip.py")
    g = torch.Generator().manual_seed(5)
    w2d = torch.randn(8, 3, 16, 16, generator=g)
    out = {"in.w2d": w2d, "out.inflate_center_t3": clip_ref.inflate_weight(w2d, 3, center=True),
           "out.inflate_mean_t2": clip_ref.inflate_weight(w2d, 2, center=False)}
    # model at 32 x 32 / patch 16 (2 x 2 grid) loaded from a checkpoint made at 48 x 48 (3 x 3 grid), conv1 stored 2-D (kernel_size 1: the T = 1 inflation the UNITE configs use; T = 2, 3 are covered by the direct calls above)
    kw = dict(input_resolution=32, patch_size=16, width=64, layers=2, heads=1, output_dim=32, kernel_size=1, return_attn=True, clip_return_layers=[1])
    for center in (True, False):
        model = clip_ref.VisionTransformer(**kw)
        sd = {k: torch.randn(v.shape, generator=g) for k, v in model.state_dict().items()}
        sd["conv1.weight"] = torch.randn(64, 3, 16, 16, generator=g)
        sd["positional_embedding"] = torch.randn(1 + 9, 64, generator=g)
        tag = "c" if center else "m"
        out.update({f"in.{tag}.{k}": v.clone() for k, v in sd.items() if k in ("conv1.weight", "positional_embedding", "proj", "ln_pre.weight")})
        out[f"in.{tag}.seed_rest"] = 0
        full = {k: v.clone() for k, v in sd.items()}
        clip_ref.load_state_dict(model, sd, input_resolution=32, patch_size=16, center=center)
        got = model.state_dict()
        out[f"out.{tag}.conv1.weight"] = got["conv1.weight"]
        out[f"out.{tag}.positional_embedding"] = got["positional_embedding"]
        out[f"out.{tag}.proj"] = got["proj"]
        torch.save(full, os.path.join(G.OUT, f"_clip_ckpt_{tag}.tmp"))
    np.savez_compressed(os.path.join(G.OUT, "clip_ckpt.npz"), **G._np(out))
    for tag in ("c", "m"):
        os.remove(os.path.join(G.OUT, f"_clip_ckpt_{tag}.tmp"))
    print("clip_ckpt", {k: tuple(v.shape) for k, v in out.items() if hasattr(v, "shape")})


if __name__ == "__main__":
    main()
