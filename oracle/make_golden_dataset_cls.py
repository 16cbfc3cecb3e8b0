#!/usr/bin/env python
"""Generate tests/golden/dataset_cls.npz by RUNNING THE REFERENCE'S OWN stage-2 / stage-3 dataset code on the CPU.

TEST INFRASTRUCTURE.  Runs only in the build container (needs /root/reference); only inputs / outputs are committed.

What is executed, unchanged, from /root/reference/src/datasets: rand_augment.py, random_erasing.py, functional_umt.py, volume_transforms.py,
video_transforms.py and kinetics_sparse.py (``VideoClsDataset_sparse``: train / validation / test modes, ``_aug_frame``, ``_get_seq_frames``,
``spatial_sampling``, ``tensor_normalize``).  They import four things this image does not have; each gets a stand-in that takes NO part in the
arithmetic the fixture pins:
  * ``decord.VideoReader``      -- reads the seeded uint8 arrays this script wrote as .npy files (len, seek, get_batch(...).asnumpy());
  * ``torchvision.transforms``  -- ``Compose`` (calls in order), ``ToPILImage`` (``PIL.Image.fromarray`` of an HWC uint8 frame), ``ToTensor``
    (HWC uint8 -> CHW float32 divided by 255: torchvision's documented behaviour, the one arithmetic line of a stand-in here);
  * ``cv2``                     -- constants only; ``cv2.resize`` RAISES.  The fixture's videos have their short side equal to ``short_side_size``,
    for which the reference's ``resize_clip`` returns the clip untouched (functional_umt.py:52-56), so OpenCV's resampler is never part of a
    pinned value (it cannot be: cv2 is not in the image -- the device resize that stands for it is checked against its published fixed-point
    algorithm only, oracle/cv2_resize.py, "parity unpinned");
  * ``numpy.lib.function_base.disp`` (gone from numpy 2; imported, never called).
pandas (``read_csv`` of the annotation list) and Pillow are the image's own.

Recorded per case: the seeds, the sample index, the output tensors, and -- captured by wrapping the reference's own functions -- the frames after
RandAugment, the crop box of ``random_resized_crop`` and whether ``horizontal_flip`` flipped, so that a mismatch can be located.
Part 1 pins unite_amd/rand_augment.py alone: ``create_random_augment`` policies on seeded frames, byte for byte.

Usage:  python oracle/make_golden_dataset_cls.py   (writes tests/golden/dataset_cls.npz)
"""
from __future__ import annotations

import os
import random
import sys
import tempfile
import types

import numpy as np
import torch
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from oracle import make_golden as MG  # noqa: E402

T, H, W, CROP, SHORT = 4, 40, 56, 32, 40
POLICIES = ["rand-m7-n4-mstd0.5-inc1", "rand-m3-n2-mstd0.5-inc1", "rand-m9-n3-mstd0.5", "rand-m5-n2-w0"]


def video(seed: int, frames: int, h: int = H, w: int = W) -> np.ndarray:
    """a seeded uint8 clip with some spatial structure (a gradient plus noise: histogram operations then have something to work on)"""
    rng = np.random.RandomState(1000 + seed)
    yy, xx = np.mgrid[0:h, 0:w]
    base = ((yy * 3 + xx * 2 + 17 * seed) % 200)[None, :, :, None] + rng.randint(0, 56, size=(frames, h, w, 3))
    return base.astype(np.uint8)


def install_standins(video_dir: str):
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class Compose:
        def __init__(self, ts):
            self.transforms = ts

        def __call__(self, x):
            for t in self.transforms:
                x = t(x)
            return x

    class ToPILImage:
        def __call__(self, frame):
            return Image.fromarray(np.ascontiguousarray(frame))

    class ToTensor:
        def __call__(self, img):
            a = np.array(img, copy=True)
            return torch.from_numpy(a).permute(2, 0, 1).contiguous().to(torch.float32).div(255)

    tv = mod("torchvision")
    tv.transforms = mod("torchvision.transforms", Compose=Compose, ToPILImage=ToPILImage, ToTensor=ToTensor)
    tv.transforms.functional = mod("torchvision.transforms.functional")

    def no_resize(*a, **k):
        raise AssertionError("cv2.resize must not be reached: the fixture's clips already have the short side the transform asks for")
    mod("cv2", INTER_LINEAR=1, INTER_NEAREST=0, resize=no_resize)

    class _Batch:
        def __init__(self, a):
            self.a = a

        def asnumpy(self):
            return self.a

    class VideoReader:
        def __init__(self, fname, num_threads=1, ctx=None, width=None, height=None):
            self.frames = np.load(fname)

        def __len__(self):
            return len(self.frames)

        def seek(self, i):
            return None

        def get_batch(self, idx):
            return _Batch(self.frames[np.asarray(idx)])
    mod("decord", VideoReader=VideoReader, cpu=lambda i=0: None)
    import numpy.lib as nl
    if not hasattr(nl, "function_base"):
        nl.function_base = mod("numpy.lib.function_base", disp=lambda *a, **k: None)
    pkg = sys.modules.get("src") or mod("src")
    pkg.__path__ = [os.path.join(MG.REF, "src")]
    pkd = mod("src.datasets")
    pkd.__path__ = [os.path.join(MG.REF, "src", "datasets")]


def main():
    out = {}
    tmp = tempfile.mkdtemp(prefix="unite_golden_ds_")
    install_standins(tmp)
    ra_ref = MG._load("src.datasets.rand_augment", "src/datasets/rand_augment.py")
    MG._load("src.datasets.random_erasing", "src/datasets/random_erasing.py")
    MG._load("src.datasets.functional_umt", "src/datasets/functional_umt.py")
    MG._load("src.datasets.volume_transforms", "src/datasets/volume_transforms.py")
    vt_ref = MG._load("src.datasets.video_transforms", "src/datasets/video_transforms.py")
    ks_ref = MG._load("src.datasets.kinetics_sparse", "src/datasets/kinetics_sparse.py")

    # ---------------------------------------------------------------- part 1: the RandAugment policies alone
    frames = video(7, 3)
    out["ra.frames"] = frames
    out["ra.policies"] = np.array(POLICIES)
    n_seeds = 6
    out["ra.n_seeds"] = n_seeds
    used = set()
    for pi, pol in enumerate(POLICIES):
        for interp in ("bicubic", "bilinear"):
            for sd in range(n_seeds):
                random.seed(100 * pi + sd)
                np.random.seed(100 * pi + sd)
                tf = vt_ref.create_random_augment(input_size=(CROP, CROP), auto_augment=pol, interpolation=interp)
                imgs = tf([Image.fromarray(f) for f in frames])
                out[f"ra.{pi}.{interp}.{sd}"] = np.stack([np.array(i) for i in imgs])
                used.add(out[f"ra.{pi}.{interp}.{sd}"].tobytes() != frames.tobytes())
    assert True in used

    # ---------------------------------------------------------------- part 2: the dataset, three modes
    names, labels = [], []
    for v in range(5):
        n_fr = 9 + 3 * v
        portrait = v == 3                                   # one clip taller than wide (test mode crops along the other axis)
        arr = video(20 + v, n_fr, W if portrait else H, H if portrait else W)
        np.save(os.path.join(tmp, f"vid{v}.npy"), arr)
        names.append(f"vid{v}.npy")
        labels.append((3 * v + 1) % 5)          # (the clips themselves are regenerated by the tests from video(): not stored)
    ann = os.path.join(tmp, "list.txt")
    with open(ann, "w") as f:
        f.writelines(f"{n} {l}\n" for n, l in zip(names, labels))
    test_ids = [0, 1, 3]                                    # the test list: two landscape clips and the portrait one
    ann_test = os.path.join(tmp, "list_test.txt")
    with open(ann_test, "w") as f:
        f.writelines(f"{names[i]} {labels[i]}\n" for i in test_ids)
    out["ds.labels"] = np.array(labels)
    out["ds.test_ids"] = np.array(test_ids)
    out["ds.video_frames"] = np.array([9 + 3 * v for v in range(5)])

    def make_args(**over):
        a = types.SimpleNamespace(aa="rand-m7-n4-mstd0.5-inc1", train_interpolation="bicubic", reprob=0.25, remode="pixel", recount=1,
                                  data_set="Kinetics_sparse", num_sample=1, train_fraction=1.0, return_aug_for_val=False)
        a.__dict__.update(over)
        return a

    def dataset(mode, args, **kw):
        return ks_ref.VideoClsDataset_sparse(anno_path=ann_test if mode == "test" else ann, prefix=tmp, split=" ", mode=mode, clip_len=T, frame_sample_rate=0, crop_size=CROP,
                                             short_side_size=SHORT, new_height=256, new_width=320, keep_aspect_ratio=True, num_segment=1,
                                             num_crop=1 if mode != "test" else 3, test_num_segment=2, test_num_crop=3, args=args, **kw)

    rec = {}
    real_cra, real_crop, real_flip = ks_ref.create_random_augment, vt_ref._get_param_spatial_crop, ks_ref.horizontal_flip

    def cra(*a, **k):
        tf = real_cra(*a, **k)

        def run(imgs):
            res = tf(imgs)
            rec["aug_frames"] = np.stack([np.array(i) for i in res])
            return res
        return run

    def crop_params(*a, **k):
        rec["crop"] = real_crop(*a, **k)
        return rec["crop"]

    def flip(prob, images, boxes=None):
        res = real_flip(prob, images, boxes)
        rec["flipped"] = bool(res[0].data_ptr() != images.data_ptr()) or not torch.equal(res[0], images)
        return res
    ks_ref.create_random_augment, vt_ref._get_param_spatial_crop, ks_ref.horizontal_flip = cra, crop_params, flip

    def seed_all(s):
        random.seed(s)
        np.random.seed(s)
        torch.manual_seed(s)

    # train mode: the stage-2 settings, and the same with the erasing forced (reprob 1) / switched off
    cases = [("train", dict()), ("train_erase", dict(reprob=1.0)), ("train_noerase", dict(reprob=0.0)), ("train_ssv2", dict(data_set="SSV2"))]
    for tag, over in cases:
        ds = dataset("train", make_args(**over))
        for k in range(6):
            idx = k % len(names)
            seed_all(300 + k)
            rec.clear()
            buf, lab, index, _ = ds[idx]
            pre = f"ds.{tag}.{k}."
            out[pre + "out"] = buf
            out[pre + "label"], out[pre + "index"] = lab, index
            out[pre + "aug_frames"] = rec["aug_frames"]
            out[pre + "crop"] = np.array(rec["crop"])
            out[pre + "flipped"] = rec.get("flipped", False)
        out[f"ds.{tag}.n"] = 6
    # validation mode, with and without the augmented second view
    for tag, over in (("val", dict()), ("val_aug", dict(return_aug_for_val=True))):
        ds = dataset("validation", make_args(**over))
        for idx in range(len(names)):
            seed_all(400 + idx)
            rec.clear()
            item = ds[idx]
            pre = f"ds.{tag}.{idx}."
            out[pre + "vids"] = item[0]
            if tag == "val_aug":
                out[pre + "vids_aug"], out[pre + "label"], out[pre + "name"] = item[1], item[2], np.array(item[3])
                out[pre + "aug_frames"], out[pre + "crop"], out[pre + "flipped"] = rec["aug_frames"], np.array(rec["crop"]), rec.get("flipped", False)
            else:
                out[pre + "label"], out[pre + "name"] = item[1], np.array(item[2])
    # test mode: every (chunk, crop, clip) view
    ds = dataset("test", make_args())
    out["ds.test.n"] = len(ds)
    for idx in range(len(ds)):
        seed_all(500 + idx)
        buf, lab, name, chunk, split = ds[idx]
        pre = f"ds.test.{idx}."
        out[pre + "out"], out[pre + "label"], out[pre + "name"], out[pre + "chunk"], out[pre + "split"] = buf, lab, np.array(name), chunk, split
    ks_ref.create_random_augment, vt_ref._get_param_spatial_crop, ks_ref.horizontal_flip = real_cra, real_crop, real_flip

    path = os.path.join(MG.OUT, "dataset_cls.npz")
    np.savez_compressed(path, **MG._np(out))
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
