"""Stage-2 / stage-3 input path: the labelled video dataset of src/datasets/kinetics_sparse.py (``VideoClsDataset_sparse``, train / validation /
test modes) and ``build_dataset`` (src/datasets/build.py:105-257), split between loader workers and the GPU like the stage-1 path
(unite_amd/datasets.py).

What runs where:
  workers : annotation list, frame numbers (``data.get_seq_frames``: the reference's ``_get_seq_frames``), decoding, and EVERY random draw in the
            reference's order -- RandAugment (unite_amd/rand_augment.py, Pillow on the host: see that file for why), the box of
            ``random_resized_crop``, the flip, the erasing rectangle and its noise (``torch.empty(...).normal_()`` from the CPU generator, as
            the reference draws it) -- so a seeded run sees the reference's clips;
  device  : all the per-pixel arithmetic behind RandAugment: ToTensor + normalise + crop + bilinear resize + flip in one kernel
            (``unite_train_clip_u8``), the erasing rectangle, and for validation / test views the short-side resize
            (``unite_resize_u8_linear``: OpenCV's 8-bit linear resize), centre / three-position crop and normalisation
            (``unite_clip_u8_to_f32``).
``__getitem__`` therefore returns RAW samples (dicts of uint8 frames + draws); ``datasets.DeviceLoader`` hands lists of them to
``dataset.transform.batch``, which returns the reference's batch tuples:
  train       (videos (B,3,T,S,S) f32, labels (B,) i64, indices (B,) i64, {})                                   kinetics_sparse.py:160
  validation  (videos, labels, names)  |  with return_aug_for_val: (videos, augmented videos, labels, names)     :178-182
  test        (videos, labels, names, chunk_nb (B,), split_nb (B,))                                              :214-215
Pinned on the reference's own class (tests/golden/dataset_cls.npz, oracle/make_golden_dataset_cls.py): train outputs to 5e-6 (f32 bilinear),
validation / test outputs bit for bit on clips that need no resize; the resize itself is OpenCV's published algorithm, parity unpinned (cv2 is
not in the image: oracle/cv2_resize.py)."""
from __future__ import annotations

import copy
import math
import os
import random
from typing import Callable, List, Optional

import numpy as np
import torch
from PIL import Image

from . import data as D
from .datasets import open_video, read_annotations
from .rand_augment import create_random_augment

MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)          # kinetics_sparse.py:113-114,246-248


# ----------------------------------------------------------------------------- draws (worker side)
def spatial_crop_params(scale, ratio, height: int, width: int, num_repeat: int = 10):
    """video_transforms.py:518-557 ``_get_param_spatial_crop`` (log-uniform aspect ratio, no height / width switch): up to ten attempts at an
    area in scale x the frame and an aspect ratio in `ratio`, then the central fallback.  Draws per attempt: random.uniform (area),
    random.uniform (log ratio), numpy.random.uniform (the unused switch coin is still tossed), and on success two random.randint."""
    area = height * width
    log_lo, log_hi = math.log(ratio[0]), math.log(ratio[1])
    for _ in range(num_repeat):
        target = random.uniform(*scale) * area
        aspect = math.exp(random.uniform(log_lo, log_hi))
        w, h = int(round(math.sqrt(target * aspect))), int(round(math.sqrt(target / aspect)))
        np.random.uniform()                                    # `np.random.uniform() < 0.5 and switch_hw` with switch_hw False
        if 0 < w <= width and 0 < h <= height:
            return random.randint(0, height - h), random.randint(0, width - w), h, w
    frame = float(width) / float(height)
    if frame < min(ratio):
        w, h = width, int(round(width / min(ratio)))
    elif frame > max(ratio):
        h, w = height, int(round(height * max(ratio)))
    else:
        w, h = width, height
    return (height - h) // 2, (width - w) // 2, h, w


def erase_draw(prob: float, mode: str, max_count: int, num_splits: int, frames: int, chan: int, img_h: int, img_w: int, min_area=0.02,
               max_area=1 / 3, min_aspect=0.3):
    """random_erasing.py:49-192 as ``_aug_frame`` configures it (RandomErasing(reprob, mode=remode, max_count=recount, num_splits=recount,
    device='cpu'), ``cube`` form: ONE rectangle for the clip, fresh noise per frame): the rectangles and their fill, or [] -- drawn from
    ``random`` and torch's CPU generator in the reference's order."""
    if random.random() > prob:
        return []
    log_lo, log_hi = math.log(min_aspect), math.log(1 / min_aspect)
    max_count = max_count or 1
    count = 1 if max_count == 1 else random.randint(1, max_count)
    start = frames // num_splits if num_splits > 1 else 0
    area, rects = img_h * img_w, []
    for _ in range(count):
        for _ in range(100):
            target = random.uniform(min_area, max_area) * area / count
            aspect = math.exp(random.uniform(log_lo, log_hi))
            h, w = int(round(math.sqrt(target * aspect))), int(round(math.sqrt(target / aspect)))
            if w < img_w and h < img_h:
                top, left = random.randint(0, img_h - h), random.randint(0, img_w - w)
                if mode == "pixel":
                    fill = torch.stack([torch.empty((chan, h, w), dtype=torch.float32).normal_() for _ in range(start, frames)])
                elif mode == "rand":
                    fill = torch.stack([torch.empty((chan, 1, 1), dtype=torch.float32).normal_() for _ in range(start, frames)])
                else:
                    fill = torch.zeros((frames - start, chan, 1, 1), dtype=torch.float32)
                rects.append((start, top, left, h, w, fill))
                break
    return rects


# ----------------------------------------------------------------------------- device side
class DeviceClsTransform:
    """assembles the reference's batch tuples on the GPU from the workers' raw samples (one instance per dataset / mode)"""

    def __init__(self, mode: str, crop_size: int, short_side_size: int, return_aug_for_val: bool = False):
        self.mode, self.crop_size, self.short_side_size, self.return_aug_for_val = mode, int(crop_size), int(short_side_size), return_aug_for_val

    # -- pieces
    def _train_clip(self, s, device):
        from . import ops
        frames = torch.from_numpy(s["aug_frames"]).to(device, non_blocking=True).contiguous()
        S = self.crop_size
        out = torch.empty(3, frames.shape[0], S, S, dtype=torch.float32, device=device)
        ops.train_clip_u8(frames, out, s["crop"], s["flip"], MEAN, STD)
        for start, top, left, h, w, fill in s["erase"]:
            out[:, start:, top:top + h, left:left + w] = fill.to(device, non_blocking=True).permute(1, 0, 2, 3)
        return out

    def _short_side(self, frames):
        """Resize(short_side_size, 'bilinear') of a uint8 clip on the device (functional_umt.py:44-66, sizes :93-100)"""
        from . import ops
        T, H, W, _ = frames.shape
        size = self.short_side_size
        if (W <= H and W == size) or (H <= W and H == size):
            return frames
        oh, ow = (int(size * H / W), size) if W < H else (size, int(size * W / H))
        return ops.resize_u8_linear(frames, torch.empty(T, oh, ow, 3, dtype=torch.uint8, device=frames.device))

    @staticmethod
    def _to_float(frames_u8):
        """ClipToTensor + Normalize (volume_transforms.py:40-86, functional_umt.py:103-116) -> (3, T, H, W)"""
        from . import ops
        T, H, W, _ = frames_u8.shape
        if W % 4:
            raise NotImplementedError(f"clip width {W}: the device conversion moves four pixels per lane (every crop of the shipped configs is 224 wide)")
        out = torch.empty(1, 3, T, H, W, dtype=torch.float32, device=frames_u8.device)
        return ops.clip_u8_to_f32(frames_u8.contiguous().unsqueeze(0), out, MEAN, STD)[0]

    def _val_clip(self, frames_np, device):
        fr = self._short_side(torch.from_numpy(frames_np).to(device, non_blocking=True).contiguous())
        _, H, W, _ = fr.shape
        c = self.crop_size
        if c > W or c > H:
            raise ValueError(f"Initial image size should be larger then cropped size but got cropped sizes : ({c}, {c}) while initial image is ({W}, {H})")
        x1, y1 = int(round((W - c) / 2.)), int(round((H - c) / 2.))                    # CenterCrop, video_transforms.py:1183-1185
        return self._to_float(fr[:, y1:y1 + c, x1:x1 + c, :])

    def _test_clip(self, frames_np, split_nb, test_num_crop, device):
        fr = self._short_side(torch.from_numpy(frames_np).to(device, non_blocking=True).contiguous())
        _, H, W, _ = fr.shape
        size = self.short_side_size
        if test_num_crop == 1:                                                         # kinetics_sparse.py:199-210
            start = int(1.0 * (max(H, W) - size) / 2)
        else:
            start = int(split_nb * (1.0 * (max(H, W) - size) / (test_num_crop - 1)))
        view = fr[:, start:start + size, :, :] if H >= W else fr[:, :, start:start + size, :]
        return self._to_float(view)

    # -- batches
    def batch(self, samples: List[dict], device):
        device = torch.device(device)
        labels = torch.tensor([int(s["label"]) for s in samples], dtype=torch.int64)
        if self.mode == "train":
            videos = torch.stack([self._train_clip(s, device) for s in samples])
            return videos, labels, torch.tensor([int(s["index"]) for s in samples], dtype=torch.int64), {}
        names = [s["name"] for s in samples]
        if self.mode == "validation":
            videos = torch.stack([self._val_clip(s["frames"], device) for s in samples])
            if self.return_aug_for_val:
                return videos, torch.stack([self._train_clip(s, device) for s in samples]), labels, names
            return videos, labels, names
        videos = torch.stack([self._test_clip(s["frames"], s["split_nb"], s["test_num_crop"], device) for s in samples])
        return (videos, labels, names, torch.tensor([int(s["chunk_nb"]) for s in samples], dtype=torch.int64),
                torch.tensor([int(s["split_nb"]) for s in samples], dtype=torch.int64))


# ----------------------------------------------------------------------------- the dataset
class VideoClsDataset_sparse(torch.utils.data.Dataset):
    """src/datasets/kinetics_sparse.py:48-357, constructor arguments as build.py:131-148 passes them.  ``reader(path)`` opens a video
    (default: .npy arrays, else decord when installed)."""

    def __init__(self, anno_path, prefix='', split=' ', mode='train', clip_len=8, frame_sample_rate=2, crop_size=224, short_side_size=256,
                 new_height=256, new_width=340, keep_aspect_ratio=True, num_segment=1, num_crop=1, test_num_segment=10, test_num_crop=3,
                 args=None, reader: Optional[Callable] = None):
        super().__init__()
        assert num_segment == 1
        if not keep_aspect_ratio:
            raise NotImplementedError("keep_aspect_ratio=False (decode-time resize to new_width x new_height) is not built: build_dataset passes True")
        if mode not in ("train", "validation", "test"):
            raise NameError('mode {} unkown'.format(mode))
        self.anno_path, self.prefix, self.split, self.mode = anno_path, prefix, split, mode
        self.clip_len, self.frame_sample_rate, self.crop_size, self.short_side_size = clip_len, frame_sample_rate, crop_size, short_side_size
        self.test_num_segment, self.test_num_crop, self.args = test_num_segment, test_num_crop, args
        self.aug = mode == 'train'
        self.rand_erase = mode == 'train' and args.reprob > 0
        self.fraction = args.train_fraction
        self.return_aug_for_val = bool(getattr(args, 'return_aug_for_val', False))
        self.reader = reader or open_video
        if getattr(args, "num_sample", 1) > 1:
            raise NotImplementedError("num_sample > 1 (repeated augmentation) is not built (num_sample: 1 in every UNITE config)")
        clips = read_annotations(anno_path, split)
        self.dataset_samples, self.label_array = [c[0] for c in clips], [c[1] for c in clips]
        if self.fraction < 1.0 and mode == 'train':
            keep = int(self.fraction * len(self.dataset_samples))
            print(f"Downsampling the dataset to {keep} samples (fraction={self.fraction}))")
            chosen = random.sample(range(len(self.dataset_samples)), keep)
            self.dataset_samples = [self.dataset_samples[i] for i in chosen]
            self.label_array = [self.label_array[i] for i in chosen]
        if mode == 'test':                       # every (temporal chunk, spatial crop) view of every video, chunk-major (:118-130)
            self.test_seg = [(ck, cp) for ck in range(test_num_segment) for cp in range(test_num_crop) for _ in self.label_array]
            self.test_dataset = self.dataset_samples * (test_num_segment * test_num_crop)
            self.test_label_array = self.label_array * (test_num_segment * test_num_crop)
        self.transform = DeviceClsTransform(mode, crop_size, short_side_size, self.return_aug_for_val)

    def __len__(self):
        return len(self.test_dataset) if self.mode == 'test' else len(self.dataset_samples)

    # -- decoding (loadvideo_decord, :314-349)
    def loadvideo(self, sample: str, chunk_nb: int = 0) -> np.ndarray:
        path = os.path.join(self.prefix, sample)
        try:
            video = self.reader(path)
            ids = D.get_seq_frames(len(video), self.clip_len, clip_idx=chunk_nb, skip_frames=self.frame_sample_rate, mode=self.mode,
                                   test_num_segment=self.test_num_segment)
            return np.ascontiguousarray(video.get_batch(ids))
        except Exception as e:                                  # noqa: BLE001 (the reference catches everything and raises this, :345-348)
            print("video cannot be loaded by decord: ", path)
            raise FileNotFoundError(path) from e

    # -- the training augmentation's host half (_aug_frame, :218-281): RandAugment in Pillow + every draw behind it
    def aug_draw(self, buffer: np.ndarray, args, erase: bool) -> dict:
        policy = create_random_augment(input_size=(self.crop_size, self.crop_size), auto_augment=args.aa, interpolation=args.train_interpolation)
        frames = policy([Image.fromarray(f) for f in buffer])
        aug = np.stack([np.asarray(f) for f in frames])
        T, H, W, _ = aug.shape
        crop = spatial_crop_params((0.08, 1.0), (0.75, 1.3333), H, W)
        flip = bool(np.random.uniform() < 0.5) if args.data_set != 'SSV2' else False
        rects = erase_draw(args.reprob, args.remode, args.recount, args.recount, T, 3, self.crop_size, self.crop_size) if erase else []
        return dict(aug_frames=aug, crop=crop, flip=flip, erase=rects)

    @staticmethod
    def name_of(sample: str) -> str:
        return sample.split("/")[-1].split(".")[0]

    def __getitem__(self, index):
        if self.mode == 'train':
            sample = self.dataset_samples[index]
            raw = self.aug_draw(self.loadvideo(sample, chunk_nb=-1), self.args, self.rand_erase)
            raw.update(label=self.label_array[index], index=index)
            return raw
        if self.mode == 'validation':
            sample = self.dataset_samples[index]
            buffer = self.loadvideo(sample, chunk_nb=0)
            raw = dict(frames=buffer, label=self.label_array[index], name=self.name_of(sample))
            if self.return_aug_for_val:          # the weaker policy of the augmented second view, no erasing (:171-178)
                val_args = copy.deepcopy(self.args)
                val_args.aa, val_args.reprob = 'rand-m3-n2-mstd0.5-inc1', 0.00
                raw.update(self.aug_draw(buffer, val_args, False))
            return raw
        sample = self.test_dataset[index]
        chunk_nb, split_nb = self.test_seg[index]
        return dict(frames=self.loadvideo(sample, chunk_nb=chunk_nb), label=self.test_label_array[index], name=self.name_of(sample),
                    chunk_nb=chunk_nb, split_nb=split_nb, test_num_crop=self.test_num_crop)


def build_dataset(is_train, test_mode, args, annotation_file=None, reader=None):
    """src/datasets/build.py:105-257 for the data sets UNITE's configs name ('Kinetics_sparse', 'mitv1_sparse'): -> (dataset, nb_classes)"""
    print(f'Use Dataset: {args.data_set}')
    if args.data_set not in ('Kinetics_sparse', 'mitv1_sparse'):
        print(f'Wrong: {args.data_set}')
        raise NotImplementedError(f"data_set {args.data_set!r}: only the sparse-sampling video list of the UNITE configs is built "
                                  "(the dense-sampling / raw-frame / SSV2 / UCF101 / HMDB51 branches of build.py are not)")
    if is_train is True:
        mode, anno_path = 'train', args.ann_file_train
    elif test_mode is True:
        mode, anno_path = 'test', args.ann_file_test
    else:
        mode, anno_path = 'validation', args.ann_file_val
    if annotation_file is not None:
        anno_path = annotation_file
    dataset = VideoClsDataset_sparse(anno_path=anno_path, prefix=args.prefix, split=args.split, mode=mode, clip_len=args.num_frames,
                                     frame_sample_rate=args.sampling_rate, num_segment=1, test_num_segment=args.test_num_segment,
                                     test_num_crop=args.test_num_crop, num_crop=1 if not test_mode else 3, keep_aspect_ratio=True,
                                     crop_size=args.input_size, short_side_size=args.short_side_size, new_height=256, new_width=320, args=args,
                                     reader=reader)
    print("Number of the class = %d" % args.nb_classes)
    return dataset, args.nb_classes
