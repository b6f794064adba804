"""ORACLE (test infrastructure, never the product path): CPU restatement of the datasets' image preprocessing
(SURVEY.md §8f row 3).

  resize_longest_side()  /root/reference/model/segment_anything/utils/transforms.py:27-36,102-113 -- ResizeLongestSide.apply_image,
                         i.e. torchvision.transforms.functional.resize(to_pil_image(img), size) = PIL Image.resize(BILINEAR).
                         The arithmetic lives in a third-party dependency absent from /root/reference: Pillow (requirements.txt
                         pins pillow==9.4.0; 12.2.0 is installed here), src/libImaging/Resample.c: precompute_coeffs,
                         normalize_coeffs_8bpc, ImagingResampleHorizontal_8bpc / Vertical_8bpc, restated below in numpy integers.
  preprocess()           /root/reference/utils/PAVE_dataset.py:115-121 -- (x - pixel_mean) / pixel_std, zero pad to the square.
Pinned by tests/golden/prep_*.npz: outputs of the reference's ResizeLongestSide.apply_image driven through the installed Pillow.
"""
import math

import numpy as np

PRECISION_BITS = 22


def _coeffs(in_size, out_size):
    scale = filterscale = in_size / out_size
    filterscale = max(filterscale, 1.0)
    support = filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    out = []
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = []
        ss = 1.0 / filterscale
        ww = 0.0
        for x in range(xmax):
            a = abs((x + xmin - center + 0.5) * ss)
            v = 1.0 - a if a < 1.0 else 0.0
            w.append(v)
            ww += v
        k = [int(0.5 + (v / ww if ww != 0.0 else v) * (1 << PRECISION_BITS)) for v in w]
        out.append((xmin, k))
    return out, ksize


def _pass(img, out_size, axis):
    """one 8-bit pass along `axis` of an [H,W,C] uint8 array"""
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    co, _ = _coeffs(src.shape[0], out_size)
    dst = np.empty((out_size,) + src.shape[1:], dtype=np.uint8)
    for i, (lo, k) in enumerate(co):
        acc = np.full(src.shape[1:], 1 << (PRECISION_BITS - 1), dtype=np.int64)
        for j, c in enumerate(k):
            acc += src[lo + j] * c
        dst[i] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(dst, 0, axis)


def preprocess_shape(oldh, oldw, long_side):
    scale = long_side * 1.0 / max(oldh, oldw)
    return int(oldh * scale + 0.5), int(oldw * scale + 0.5)


def resize_longest_side(img, target):
    """img [H,W,3] uint8 -> [Ho,Wo,3] uint8 (horizontal pass first, then vertical, each only if that size changes)."""
    ho, wo = preprocess_shape(img.shape[0], img.shape[1], target)
    out = img
    if wo != img.shape[1]:
        out = _pass(out, wo, 1)
    if ho != img.shape[0]:
        out = _pass(out, ho, 0)
    return out


def preprocess(resized, target, mean, std):
    """[Ho,Wo,3] uint8 -> [3,target,target] float32"""
    x = (resized.astype(np.float32).transpose(2, 0, 1) - np.asarray(mean, np.float32)[:, None, None]) / np.asarray(std, np.float32)[:, None, None]
    out = np.zeros((3, target, target), dtype=np.float32)
    out[:, :x.shape[1], :x.shape[2]] = x
    return out
