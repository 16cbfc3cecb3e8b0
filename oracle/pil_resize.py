"""CPU restatement of Pillow's bilinear ``Image.resize`` for 8-bit images (TEST INFRASTRUCTURE: the checker of unite_crop_resize_u8).

The reference resizes every cropped frame with ``img.resize((w, h), Image.BILINEAR)`` (src/datasets/transforms.py:136-152
GroupMultiScaleCrop; build.py:37) through Pillow, pinned as pillow==10.0.1 in environment.yaml:259 and not vendored in /root/reference.  This
follows the published algorithm of that version (src/libImaging/Resample.c: ``precompute_coeffs`` with the triangle filter whose support grows
with the down-scaling factor, ``normalize_coeffs_8bpc`` to 22-bit fixed point, a horizontal pass then a vertical pass, each rounded to uint8
through ``clip8``).  Pinned: tests/test_host_logic.py compares it bit for bit with the Pillow installed in the image (12.2, same routine).
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def _tri(x):
    x = -x if x < 0.0 else x
    return 1.0 - x if x < 1.0 else 0.0


def coeffs(in_size: int, out_size: int):
    """-> (bounds int32 [out, 2] = (first tap, number of taps), kk int32 [out, ksize] fixed-point weights)"""
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [_tri((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _pass(img, bounds, kk, axis):
    """img uint8 [H, W, C]; resample along `axis` (1: horizontal, 0: vertical)"""
    src = img.astype(np.int64)
    n_out = bounds.shape[0]
    shape = list(img.shape)
    shape[axis] = n_out
    out = np.empty(shape, dtype=np.uint8)
    for o in range(n_out):
        lo, n = int(bounds[o, 0]), int(bounds[o, 1])
        taps = np.take(src, range(lo, lo + n), axis=axis)
        k = kk[o, :n].astype(np.int64)
        acc = (taps * (k[None, :, None] if axis == 1 else k[:, None, None])).sum(axis=axis) + (1 << (PRECISION_BITS - 1))
        v = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
        if axis == 1:
            out[:, o] = v
        else:
            out[o] = v
    return out


def crop_resize_bilinear(img: np.ndarray, box, out_hw):
    """img uint8 [H, W, C]; box (x0, y0, w, h); -> uint8 [OH, OW, C]: ``img.crop(box).resize((OW, OH), Image.BILINEAR)``"""
    x0, y0, w, h = box
    crop = img[y0:y0 + h, x0:x0 + w]
    OH, OW = out_hw
    if (h, w) == (OH, OW):
        return crop.copy()
    bx, kx = coeffs(w, OW)
    by, ky = coeffs(h, OH)
    tmp = _pass(crop, bx, kx, 1) if w != OW else crop
    return _pass(tmp, by, ky, 0) if h != OH else tmp
