"""OpenCV's 8-bit INTER_LINEAR resize, restated in numpy.  TEST INFRASTRUCTURE -- parity unpinned.

The reference's validation / test transforms resize decoded uint8 frames with ``cv2.resize(img, size, interpolation=cv2.INTER_LINEAR)``
(src/datasets/functional_umt.py:44-66; opencv-python is a dependency of the reference, environment.yaml, and is NOT vendored in /root/reference
nor installed in this image).  This file restates the published algorithm of OpenCV 4.x ``modules/imgproc/src/resize.cpp`` for 8UC3 images --
``resizeGeneric_`` with ``HResizeLinear<uchar,int,short,INTER_RESIZE_COEF_SCALE>`` and ``VResizeLinear<uchar,int,short,FixedPtCast<...>>``:

  inv_scale = dsize / ssize (double), scale = 1 / inv_scale
  per output column dx:  fx = (float)((dx + 0.5) * scale_x - 0.5);  sx = floor(fx);  fx -= sx
      sx < 0          -> fx = 0, sx = 0                    sx >= width - 1 -> fx = 0, sx = width - 1
      weights ialpha = saturate_cast<short>((1 - fx) * 2048), saturate_cast<short>(fx * 2048)      (cvRound: half to even)
  per output row dy the same fraction; the two source rows sy, sy + 1 are clipped into the image, the weights are not touched
  horizontal pass (int):  D[dx] = S[sx] * a0 + S[sx + 1] * a1            (S[sx] * 2048 where sx + 1 would leave the row)
  vertical pass:          dst = uchar((((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2)

There is no cv2 here to check it against, and no cv2 output among the reference's files: the device kernel ``unite_resize_u8_linear`` is held to
THIS file bit for bit, and the header of both says "parity unpinned".  What IS pinned of the validation / test path (sizes, crop offsets,
order of operations, normalisation) is pinned on clips that need no resize (tests/golden/dataset_cls.npz)."""
from __future__ import annotations

import numpy as np


def _coefficients(n_dst: int, n_src: int, horizontal: bool):
    inv = np.float64(n_dst) / np.float64(n_src)
    scale = 1.0 / inv
    d = np.arange(n_dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    if horizontal:
        low, high = s < 0, s >= n_src - 1
        f = np.where(low | high, np.float32(0), f)
        s = np.where(low, 0, np.where(high, n_src - 1, s))
        s0, s1 = s, np.minimum(s + 1, n_src - 1)
    else:
        s0, s1 = np.clip(s, 0, n_src - 1), np.clip(s + 1, 0, n_src - 1)
    a0 = np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int16).astype(np.int64)      # rint: half to even, as cvRound
    a1 = np.rint(f * np.float32(2048)).astype(np.int16).astype(np.int64)
    return s0, s1, a0, a1


def resize_linear_u8(frames: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """frames uint8 (..., H, W, C) -> uint8 (..., out_h, out_w, C)"""
    H, W = frames.shape[-3], frames.shape[-2]
    if (H, W) == (out_h, out_w):
        return frames.copy()
    x0, x1, a0, a1 = _coefficients(out_w, W, True)
    y0, y1, b0, b1 = _coefficients(out_h, H, False)
    src = frames.astype(np.int64)
    horiz = src[..., :, x0, :] * a0[:, None] + src[..., :, x1, :] * a1[:, None]          # (..., H, out_w, C), scaled by 2048
    r0, r1 = horiz[..., y0, :, :], horiz[..., y1, :, :]
    out = (((b0[:, None, None] * (r0 >> 4)) >> 16) + ((b1[:, None, None] * (r1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def resize_sizes(im_h: int, im_w: int, size: int):
    """functional_umt.py:93-100 ``get_resize_sizes``: the short side becomes `size`, the long side is truncated"""
    if im_w < im_h:
        return int(size * im_h / im_w), size
    return size, int(size * im_w / im_h)
