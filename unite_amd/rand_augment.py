"""RandAugment for clips (host side, PIL): the policy ``create_random_augment`` builds for the stage-2 / stage-3 training clips.

Reference: src/datasets/video_transforms.py:640-673 (``create_random_augment``) -> src/datasets/rand_augment.py (the timm auto-augment code
adapted to lists of frames: one draw per operation, applied to every frame of the clip).  This file restates that policy as a table -- every
operation is (how a magnitude becomes its argument, how the argument is applied to one PIL image) -- and keeps the reference's draws in the
reference's order, because the clips a seeded run sees depend on them:

  per clip   : ``numpy.random.choice`` picks ``n`` operations (with replacement; without when a weight set ``w`` is given),
  per picked operation, in order:
               ``random.random()``  -- skipped when it exceeds 0.5 (nothing else is drawn then),
               ``random.gauss(m, mstd)`` when mstd > 0, clipped to [0, 10],
               ``random.random()``  -- a sign, for the operations that have one (rotate, shear, translate, the "increasing" enhancements).

The per-pixel work is Pillow's (ImageOps / ImageEnhance / Image.transform): it stays in the loader workers.  These are byte-exact library
calls on PIL images -- a device version would have to reproduce Pillow's affine resampler, histogram equalisation and enhancement blends bit
for bit for no gain, since the workers run beside the GPU; what IS on the device is everything behind it (unite_amd/datasets_cls.py).
Pinned bit for bit against the reference module on seeded clips: tests/golden/dataset_cls.npz, tests/test_host_logic.py.
"""
from __future__ import annotations

import math
import random
import re
from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
from PIL import Image, ImageEnhance, ImageOps

MAX_LEVEL = 10.0
FILL = (128, 128, 128)
INTERPOLATIONS = {"bilinear": Image.BILINEAR, "bicubic": Image.BICUBIC, "lanczos": Image.LANCZOS, "hamming": Image.HAMMING}


def _signed(v: float) -> float:
    """the reference's coin for a direction: negative when random.random() > 0.5"""
    return -v if random.random() > 0.5 else v


# ---- magnitude (0 .. 10) -> argument of the operation --------------------------------------------------------------------------------
def _none(level, hp):
    return ()


def _scaled_signed(top):
    return lambda level, hp: (_signed(level / MAX_LEVEL * top),)


def _enhance(level, hp):                     # 0.1 .. 1.9, growing with the level
    return (level / MAX_LEVEL * 1.8 + 0.1,)


def _enhance_away_from_one(level, hp):       # 1 +- up to 0.9: the "increasing" form (stronger with the level in both directions)
    return (1.0 + _signed(level / MAX_LEVEL * 0.9),)


def _translate_abs(level, hp):
    return (_signed(level / MAX_LEVEL * float(hp["translate_const"])),)


def _translate_rel(level, hp):
    return (_signed(level / MAX_LEVEL * hp.get("translate_pct", 0.45)),)


def _posterize_bits(level, hp):
    return (int(level / MAX_LEVEL * 4),)


def _solarize_threshold(level, hp):
    return (int(level / MAX_LEVEL * 256),)


# ---- the operations on ONE image (kw: fillcolor, resample) ----------------------------------------------------------------------------
def _affine(matrix_of):
    def apply(img, v, **kw):
        return img.transform(img.size, Image.AFFINE, matrix_of(img, v), **kw)
    return apply


def _solarize_add(img, add, thresh=128, **_):
    if img.mode not in ("L", "RGB"):
        return img
    lut = [min(255, i + add) if i < thresh else i for i in range(256)]
    return img.point(lut * 3 if img.mode == "RGB" else lut)


def _posterize(img, bits, **_):
    return img if bits >= 8 else ImageOps.posterize(img, bits)


@dataclass(frozen=True)
class _Kind:
    to_arg: Callable
    apply: Callable
    geometric: bool = False          # takes fillcolor / resample


_KINDS: Dict[str, _Kind] = {
    "AutoContrast": _Kind(_none, lambda img, **_: ImageOps.autocontrast(img)),
    "Equalize": _Kind(_none, lambda img, **_: ImageOps.equalize(img)),
    "Invert": _Kind(_none, lambda img, **_: ImageOps.invert(img)),
    "Rotate": _Kind(_scaled_signed(30.0), lambda img, deg, **kw: img.rotate(deg, **kw), True),
    "Posterize": _Kind(_posterize_bits, _posterize),
    "PosterizeIncreasing": _Kind(lambda l, hp: (4 - _posterize_bits(l, hp)[0],), _posterize),
    "PosterizeOriginal": _Kind(lambda l, hp: (_posterize_bits(l, hp)[0] + 4,), _posterize),
    "Solarize": _Kind(_solarize_threshold, lambda img, t, **_: ImageOps.solarize(img, t)),
    "SolarizeIncreasing": _Kind(lambda l, hp: (256 - _solarize_threshold(l, hp)[0],), lambda img, t, **_: ImageOps.solarize(img, t)),
    "SolarizeAdd": _Kind(lambda l, hp: (int(l / MAX_LEVEL * 110),), _solarize_add),
    "ShearX": _Kind(_scaled_signed(0.3), _affine(lambda img, v: (1, v, 0, 0, 1, 0)), True),
    "ShearY": _Kind(_scaled_signed(0.3), _affine(lambda img, v: (1, 0, 0, v, 1, 0)), True),
    "TranslateX": _Kind(_translate_abs, _affine(lambda img, v: (1, 0, v, 0, 1, 0)), True),
    "TranslateY": _Kind(_translate_abs, _affine(lambda img, v: (1, 0, 0, 0, 1, v)), True),
    "TranslateXRel": _Kind(_translate_rel, _affine(lambda img, v: (1, 0, v * img.size[0], 0, 1, 0)), True),
    "TranslateYRel": _Kind(_translate_rel, _affine(lambda img, v: (1, 0, 0, 0, 1, v * img.size[1])), True),
}
for _name, _enh in (("Color", ImageEnhance.Color), ("Contrast", ImageEnhance.Contrast), ("Brightness", ImageEnhance.Brightness),
                    ("Sharpness", ImageEnhance.Sharpness)):
    _KINDS[_name] = _Kind(_enhance, (lambda E: lambda img, f, **_: E(img).enhance(f))(_enh))
    _KINDS[_name + "Increasing"] = _Kind(_enhance_away_from_one, (lambda E: lambda img, f, **_: E(img).enhance(f))(_enh))

# the two operation lists of the reference, in its order (numpy.random.choice indexes into them)
PLAIN = ["AutoContrast", "Equalize", "Invert", "Rotate", "Posterize", "Solarize", "SolarizeAdd", "Color", "Contrast", "Brightness",
         "Sharpness", "ShearX", "ShearY", "TranslateXRel", "TranslateYRel"]
INCREASING = [n + "Increasing" if n in ("Posterize", "Solarize", "Color", "Contrast", "Brightness", "Sharpness") else n for n in PLAIN]
WEIGHTS_0 = {"Rotate": 0.3, "ShearX": 0.2, "ShearY": 0.2, "TranslateXRel": 0.1, "TranslateYRel": 0.1, "Color": 0.025, "Sharpness": 0.025,
             "AutoContrast": 0.025, "Solarize": 0.005, "SolarizeAdd": 0.005, "Contrast": 0.005, "Brightness": 0.005, "Equalize": 0.005,
             "Posterize": 0, "Invert": 0}


class ClipOp:
    """one operation of the policy, applied to every frame of a clip with ONE set of draws"""

    def __init__(self, name: str, magnitude: float, hparams: dict, prob: float = 0.5):
        self.name, self.kind, self.magnitude, self.prob = name, _KINDS[name], magnitude, prob
        self.hparams = dict(hparams)
        self.std = self.hparams.get("magnitude_std", 0)
        self.resample = self.hparams.get("interpolation", (Image.BILINEAR, Image.BICUBIC))
        self.fill = self.hparams.get("img_mean", FILL)

    def __call__(self, frames: List[Image.Image]) -> List[Image.Image]:
        if self.prob < 1.0 and random.random() > self.prob:
            return frames
        level = random.gauss(self.magnitude, self.std) if self.std and self.std > 0 else self.magnitude
        args = self.kind.to_arg(min(MAX_LEVEL, max(0, level)), self.hparams)
        out = []
        for img in frames:
            if self.kind.geometric:           # (the reference resolves a random interpolation per frame, when one was left open)
                rs = random.choice(self.resample) if isinstance(self.resample, (list, tuple)) else self.resample
                out.append(self.kind.apply(img, *args, fillcolor=self.fill, resample=rs))
            else:
                out.append(self.kind.apply(img, *args))
        return out

    def __repr__(self):
        return f"ClipOp({self.name}, m={self.magnitude}, mstd={self.std})"


class ClipRandAugment:
    def __init__(self, ops: Sequence[ClipOp], num_layers: int, choice_weights: Optional[np.ndarray]):
        self.ops, self.num_layers, self.choice_weights = list(ops), num_layers, choice_weights

    def __call__(self, frames: List[Image.Image]) -> List[Image.Image]:
        picked = np.random.choice(np.arange(len(self.ops)), self.num_layers, replace=self.choice_weights is None, p=self.choice_weights)
        for k in picked:
            frames = self.ops[int(k)](frames)
        return frames


def parse_policy(config: str) -> Tuple[float, int, Optional[int], Optional[float], bool]:
    """'rand-m7-n4-mstd0.5-inc1' -> (magnitude 7, layers 4, weight set None, magnitude std 0.5, increasing True); defaults m 10, n 2"""
    parts = config.split("-")
    if parts[0] != "rand":
        raise ValueError(f"only 'rand-...' policies are built, got {config!r}")
    m, n, w, std, inc = MAX_LEVEL, 2, None, None, False
    for part in parts[1:]:
        found = re.split(r"(\d.*)", part)
        if len(found) < 2:
            continue
        key, val = found[0], found[1]
        if key == "mstd":
            std = float(val)
        elif key == "inc":
            inc = inc or bool(val)               # (sic: any non-empty value switches it on, 'inc0' included -- rand_augment.py:530)
        elif key == "m":
            m = int(val)
        elif key == "n":
            n = int(val)
        elif key == "w":
            w = int(val)
    return m, n, w, std, inc


def create_random_augment(input_size, auto_augment: Optional[str] = None, interpolation: str = "bilinear") -> ClipRandAugment:
    """video_transforms.py:640-673: the RandAugment policy for clips of ``input_size``; translations are limited to 45 % of its short side"""
    if not auto_augment or not auto_augment.startswith("rand"):
        raise NotImplementedError(f"auto_augment {auto_augment!r}")
    size = input_size[-2:] if isinstance(input_size, tuple) else input_size
    hp = {"translate_const": int((min(size) if isinstance(size, tuple) else size) * 0.45)}
    if interpolation and interpolation != "random":
        hp["interpolation"] = INTERPOLATIONS.get(interpolation, Image.BILINEAR)       # (anything unknown is bilinear, video_transforms.py:52-60)
    m, n, w, std, inc = parse_policy(auto_augment)
    if std is not None:
        hp.setdefault("magnitude_std", std)
    names = INCREASING if inc else PLAIN
    weights = None
    if w is not None:
        if w != 0:
            raise ValueError("only weight set 0 exists")
        p = np.array([WEIGHTS_0[k] for k in PLAIN], dtype=np.float64)       # (the reference looks the weights up by the PLAIN names)
        weights = p / p.sum()
    return ClipRandAugment([ClipOp(name, m, hp) for name in names], n, weights)
