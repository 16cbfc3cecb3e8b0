"""Stage-1 input path around the device transform: annotation list -> sampled frames -> uint8 clips -> device batch.

What the reference does per clip inside the DataLoader workers (src/datasets/mae.py:161-222 ``VideoMAE.__getitem__`` with
build.py:32-84 ``DataAugmentationForVideoMAE``): open the video, draw the frame numbers, decode, GroupMultiScaleCrop (crop + Pillow
bilinear resize of every frame), random flip, stack, /255, normalise, (C,T,H,W), plus the tube / random mask.  Here the workers only
DRAW (frame numbers, crop box, flip, mask: same generators, same order of draws) and DECODE; they hand over the raw uint8 frames.  The
main process moves them to the GPU and two kernels do the per-pixel work (unite_crop_resize_u8: Pillow's arithmetic bit for bit;
unite_clip_u8_to_f32: flip / scale / normalise / transpose), producing the batch tuple the engine expects:
``(videos f32 (B,3,T,S,S), bool_masked_pos (B, T*N) or -1 per clip, target (B,))``.

Decoding: ``open_video`` reads ``.npy`` files (uint8 [frames,H,W,3], memory-mapped) itself and hands every other extension to decord
when that package is importable (it is not in this image; the reference requires it, kinetics_sparse.py:80-81).  Any object with
``__len__`` and ``get_batch(list_of_frame_numbers) -> uint8 ndarray [n,H,W,3]`` can be supplied instead (``reader=``).

Not built: the stage-2/3 training augmentation (``VideoClsDataset_sparse._aug_frame``: RandAugment + random erasing on PIL images,
kinetics_sparse.py:218-281 -- CPU work in the workers with no arithmetic on the device path) and the cv2-resampled validation / test
transforms (cv2 is not in the image: its fixed-point bilinear could not be pinned); the evaluation engines take tensors from any loader.
"""
from __future__ import annotations

import os
import random
from typing import Callable, Optional

import numpy as np
import torch

from . import data as D


# ----------------------------------------------------------------------------- decoding
class NpyVideo:
    """a video stored as one uint8 array [frames, H, W, 3] in a .npy file (memory-mapped: only the sampled frames are read)"""

    def __init__(self, path: str):
        self.frames = np.load(path, mmap_mode="r")
        if self.frames.ndim != 4 or self.frames.shape[-1] != 3 or self.frames.dtype != np.uint8:
            raise ValueError(f"{path}: expected uint8 [frames, H, W, 3], found {self.frames.dtype} {self.frames.shape}")

    def __len__(self):
        return self.frames.shape[0]

    def get_batch(self, ids):
        return np.ascontiguousarray(self.frames[np.asarray(ids, dtype=np.int64)])


class _Decord:
    def __init__(self, path):
        import decord
        self.vr = decord.VideoReader(path, num_threads=1, ctx=decord.cpu(0))     # mae.py:177

    def __len__(self):
        return len(self.vr)

    def get_batch(self, ids):
        return self.vr.get_batch(list(ids)).asnumpy()


def open_video(path: str):
    if path.endswith(".npy"):
        return NpyVideo(path)
    try:
        return _Decord(path)
    except ImportError as e:
        raise RuntimeError(f"{path}: decoding needs the decord package (not installed); store clips as .npy uint8 [frames,H,W,3] or pass "
                           f"reader=<callable returning an object with __len__ and get_batch>") from e


# ----------------------------------------------------------------------------- masks drawn in the loader (build.py:55-64)
class TubeMaskingGenerator:
    """masking_generator.py:23-45: ONE random set of int(mask_ratio * H*W) positions, repeated for every frame (1 = masked).  Draws with
    numpy's global generator exactly as the reference does (one np.random.shuffle of [zeros, ones]), so a seeded worker produces the same
    masks (tests/golden/sampling.json)."""

    def __init__(self, input_size, mask_ratio):
        self.frames, self.height, self.width = input_size
        self.per_frame = self.height * self.width
        self.masked_per_frame = int(mask_ratio * self.per_frame)

    def __repr__(self):
        return "Mask: total patches {}, mask patches {}".format(self.frames * self.per_frame, self.frames * self.masked_per_frame)

    def __call__(self):
        one = np.hstack([np.zeros(self.per_frame - self.masked_per_frame), np.ones(self.masked_per_frame)])
        np.random.shuffle(one)
        return np.tile(one, (self.frames, 1)).flatten()


class RandomMaskingGenerator:
    """masking_generator.py:48-69: int(mask_ratio * T*H*W) positions anywhere in the clip"""

    def __init__(self, input_size, mask_ratio):
        if not isinstance(input_size, tuple):
            input_size = (input_size,) * 3
        self.frames, self.height, self.width = input_size
        self.total = self.frames * self.height * self.width
        self.masked = int(mask_ratio * self.total)

    def __repr__(self):
        return "Mask: total patches {}, mask patches {}".format(self.total, self.masked)

    def __call__(self):
        m = np.hstack([np.zeros(self.total - self.masked), np.ones(self.masked)])
        np.random.shuffle(m)
        return m


# ----------------------------------------------------------------------------- the transform, split between worker and device
class DeviceAugmentationForVideoMAE:
    """build.py:32-84 ``DataAugmentationForVideoMAE`` with its arithmetic on the GPU.  ``draw`` runs in the worker (crop box from the
    ``random`` module, the flip's ``random.random()`` -- drawn whether or not flipping is on, as transforms.py:73 does -- then the mask
    from numpy); ``batch`` runs in the main process on the device."""

    def __init__(self, args):
        if getattr(args, "color_jitter", 0) and args.color_jitter > 0:
            raise NotImplementedError("GroupColorJitter is not built (color_jitter: 0 in the stage-1 config)")
        self.input_size = int(args.input_size)
        self.crop = D.MultiScaleCrop(self.input_size, (1, .875, .75, .66))
        self.flip = bool(getattr(args, "flip", False))
        if args.mask_type == 'tube':
            self.masked_position_generator = TubeMaskingGenerator(args.window_size, args.mask_ratio)
        elif args.mask_type == 'random':
            self.masked_position_generator = RandomMaskingGenerator(args.window_size, args.mask_ratio)
        elif args.mask_type in 'attention':               # (sic: substring test, build.py:63) the engine samples from the teacher's attention
            self.masked_position_generator = None
        else:
            raise ValueError(f"mask_type {args.mask_type!r}")
        self.to_tensor = D.ClipToTensor()
        self._u8 = self._ws = None

    # ---- worker side
    def draw(self, im_w: int, im_h: int):
        box = self.crop(im_w, im_h)
        v = random.random()
        mask = -1 if self.masked_position_generator is None else self.masked_position_generator()
        return box, bool(self.flip and v < 0.5), mask

    # ---- device side
    def batch(self, samples, device):
        """samples: list of (frames uint8 (T,H,W,3), box, flip, mask, target) -> (videos, bool_masked_pos, targets) as mae.py:217-222 collates"""
        from . import ops
        B, T, S = len(samples), samples[0][0].shape[0], self.input_size
        if self._u8 is None or tuple(self._u8.shape) != (B, T, S, S, 3) or self._u8.device != device:
            self._u8 = torch.empty(B, T, S, S, 3, dtype=torch.uint8, device=device)
        for i, (frames, box, _, _, _) in enumerate(samples):
            fr = frames.to(device, non_blocking=True).unsqueeze(0)
            need = ops.crop_resize_workspace(1, T, fr.shape[2], S, S)
            if self._ws is None or self._ws.numel() < need:
                self._ws = torch.empty(need, dtype=torch.uint8, device=device)
            ops.crop_resize_u8(fr, [box], self._u8[i:i + 1], self._ws)
        flips = torch.tensor([1 if s[2] else 0 for s in samples], dtype=torch.uint8, device=device)
        videos = self.to_tensor(self._u8, flips)
        if self.masked_position_generator is None:
            masks = torch.full((B,), -1, dtype=torch.int64)
        else:
            masks = torch.from_numpy(np.stack([s[3] for s in samples]))
        targets = torch.tensor([int(s[4]) for s in samples], dtype=torch.int64)
        return videos, masks, targets

    def __repr__(self):
        return ("(DeviceAugmentationForVideoMAE,\n  crop = GroupMultiScaleCrop(%d, [1, .875, .75, .66]) + flip=%s on the device,\n"
                "  Masked position generator = %s,\n)" % (self.input_size, self.flip, self.masked_position_generator))


def read_annotations(setting: str, split: str = ' ', with_duration: bool = False):
    """mae.py:229-251: one clip per line, '<path><split><label>' (video files) or '<path><split><frames><split><label>' (frame folders)"""
    if not os.path.exists(setting):
        raise RuntimeError("Setting file %s doesn't exist. Check opt.train-list and opt.val-list. " % setting)
    clips = []
    with open(setting) as f:
        for line in f:
            fields = line.split(split)
            if len(fields) < (3 if with_duration else 2):
                raise RuntimeError('Video input format is not correct, missing one or more element. %s' % line)
            clips.append((fields[0], int(fields[1]), int(fields[2])) if with_duration else (fields[0], int(fields[1])))
    return clips


class VideoMAE(torch.utils.data.Dataset):
    """Stage-1 training clips (src/datasets/mae.py:37-306; constructor arguments as build.py:87-106 passes them).  ``__getitem__`` returns
    ``(frames uint8 (T,H,W,3), crop box (x0,y0,w,h), flip, mask, target)`` -- raw material for ``transform.batch`` -- instead of the
    normalised float clip; ``DeviceLoader`` turns lists of them into the reference's batch tuple.  A clip that cannot be read is replaced
    by a uniformly drawn other one (mae.py:206-209)."""

    def __init__(self, root, setting, prefix='', split=' ', train=True, test_mode=False, name_pattern='img_%05d.jpg', video_ext='mp4',
                 is_color=True, modality='rgb', num_segments=1, num_crop=1, new_length=1, new_step=1, transform=None, temporal_jitter=False,
                 video_loader=False, use_decord=True, lazy_init=False, num_sample=1, fraction=1.0, reader: Optional[Callable] = None):
        super().__init__()
        if not use_decord or not video_loader:
            raise NotImplementedError("raw-frame folders (use_decord / video_loader False: cv2.imdecode of jpg files) are not built")
        if num_sample != 1:
            raise NotImplementedError("num_sample > 1 (repeated augmentation) is not built (num_sample: 1 in every UNITE config)")
        self.prefix, self.split, self.video_ext = prefix, split, video_ext
        self.num_segments, self.new_length, self.new_step = num_segments, new_length, new_step
        self.skip_length = self.new_length * self.new_step
        self.temporal_jitter, self.transform = temporal_jitter, transform
        self.reader = reader or open_video
        if self.num_segments != 1:                      # sparse sampling: one frame per segment (mae.py:142-146)
            print('Use sparse sampling, change frame and stride')
            self.new_length, self.skip_length = self.num_segments, 1
        self.clips = [] if lazy_init else read_annotations(setting, split)
        if not lazy_init and len(self.clips) == 0:
            raise RuntimeError("Found 0 video clips in " + str(setting))
        if not lazy_init and fraction < 1.0:
            self.clips = random.sample(self.clips, int(len(self.clips) * fraction))
            print(f'Using {fraction}x of the dataset for masked training, {len(self.clips)} clips left.')

    def __len__(self):
        return len(self.clips)

    def path_of(self, name: str) -> str:
        if '.' not in name.split('/')[-1]:
            name = '{}.{}'.format(name, self.video_ext)
        return os.path.join(self.prefix, name)

    def frame_numbers(self, duration: int):
        idx, skip = D.sample_train_indices(duration, self.num_segments, self.skip_length, self.new_step, self.temporal_jitter)
        return D.frame_id_list(duration, idx, skip, self.skip_length, self.new_step)

    def __getitem__(self, index):
        failures = 0
        while True:
            name, target = self.clips[index]
            try:
                video = self.reader(self.path_of(name))
                frames = video.get_batch(self.frame_numbers(len(video)))
                break
            except Exception as e:                                     # noqa: BLE001 -- any unreadable clip is replaced, as the reference does
                print("Failed to load video from {} with error {}".format(name, e))
                failures += 1
                if failures >= 64:                                     # (the reference would spin for ever on a list with nothing readable)
                    raise RuntimeError(f"64 clips in a row could not be read (last: {name}): check prefix / video_ext / the reader") from e
                index = random.randint(0, len(self.clips) - 1)
        frames = torch.from_numpy(np.ascontiguousarray(frames))
        box, flip, mask = self.transform.draw(frames.shape[2], frames.shape[1])
        return frames, box, flip, mask, target


def build_pretraining_dataset(args, annotation_file, fraction=1.0, reader=None):
    """build.py:87-108"""
    transform = DeviceAugmentationForVideoMAE(args)
    dataset = VideoMAE(root=None, setting=annotation_file, prefix=args.prefix, split=args.split, video_ext='mp4', is_color=True, modality='rgb',
                       num_segments=args.num_segments, new_length=args.num_frames, new_step=args.umt_step, transform=transform,
                       temporal_jitter=False, video_loader=True, use_decord=args.use_decord, lazy_init=False, num_sample=args.num_sample,
                       fraction=fraction, reader=reader)
    print("Data Aug = %s" % str(transform))
    return dataset


class DeviceLoader:
    """``torch.utils.data.DataLoader`` whose workers return the raw samples and whose batches are assembled on the GPU by the dataset's
    transform (in the process that owns the device).  Iterating yields the reference's batch tuple; ``sampler`` / ``dataset`` / ``len``
    as on a DataLoader (the drivers call ``loader.sampler.set_epoch``)."""

    def __init__(self, dataset, batch_size, device, sampler=None, num_workers=0, drop_last=True, worker_init_fn=None, **kw):
        self.dataset, self.device, self.sampler = dataset, torch.device(device), sampler
        self.loader = torch.utils.data.DataLoader(dataset, batch_size=batch_size, sampler=sampler, num_workers=num_workers, drop_last=drop_last,
                                                  collate_fn=list, worker_init_fn=worker_init_fn, **kw)

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        for samples in self.loader:
            yield self.dataset.transform.batch(samples, self.device)
