"""GPU input pipeline behind the reference's ResizeLongestSide name.

    /root/reference/model/segment_anything/utils/transforms.py:16-36,102-113   ResizeLongestSide.apply_image / get_preprocess_shape
    /root/reference/utils/PAVE_dataset.py:49-51,115-121,217-236                (x - pixel_mean) / pixel_std, zero pad to the square

The reference resizes every frame on the CPU through PIL (torchvision's resize of a PIL image = Image.resize(BILINEAR):
an antialiasing two-pass convolution in 8-bit fixed point).  Here the frames stay in HBM as uint8 and walkgpt_hip's
wg_preprocess_frames_u8 reproduces that arithmetic bit for bit; this module only builds Pillow's coefficient tables
(a function of the two sizes, float64 on the host exactly as Resample.c computes them) and caches them per device.
"""
import math

import numpy as np
import torch

from . import _lib

PRECISION_BITS = 32 - 8 - 2   # Resample.c
PAVE_PIXEL_MEAN = (97.17, 105.73, 108.16)   # utils/PAVE_dataset.py:49-50
PAVE_PIXEL_STD = (53.05, 56.40, 61.93)


def pil_bilinear_coeffs(in_size, out_size):
    """Resample.c precompute_coeffs (bilinear, support 1.0, box = the whole axis) + normalize_coeffs_8bpc.
    Returns (bounds int32 [out,2] = (first input index, tap count), kk int32 [out,ksize], ksize)."""
    scale = filterscale = float(in_size) / float(out_size)
    if filterscale < 1.0:
        filterscale = 1.0
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        ws = []
        ww = 0.0
        for x in range(xmax):
            a = (x + xmin - center + 0.5) * ss
            if a < 0.0:
                a = -a
            w = 1.0 - a if a < 1.0 else 0.0
            ws.append(w)
            ww += w
        for x in range(xmax):
            w = ws[x] / ww if ww != 0.0 else ws[x]
            kk[xx, x] = int(-0.5 + w * (1 << PRECISION_BITS)) if w < 0 else int(0.5 + w * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk, ksize


_TABLES = {}
_LUTS = {}


def _norm_lut(mean, std, device):
    """[3, 256] fp32: (v - mean_c) / std_c evaluated in IEEE float32 exactly as torch evaluates PAVE_dataset.py:116 on the CPU."""
    key = (tuple(float(v) for v in mean), tuple(float(v) for v in std), str(device))
    if key not in _LUTS:
        v = np.arange(256, dtype=np.float32)[None, :]
        lut = (v - np.asarray(key[0], np.float32)[:, None]) / np.asarray(key[1], np.float32)[:, None]
        _LUTS[key] = torch.from_numpy(lut.astype(np.float32)).contiguous().to(device)
    return _LUTS[key]


def _tables(in_size, out_size, device):
    key = (in_size, out_size, str(device))
    if key not in _TABLES:
        b, k, ks = pil_bilinear_coeffs(in_size, out_size)
        _TABLES[key] = (torch.from_numpy(b).to(device), torch.from_numpy(k).to(device), ks)
    return _TABLES[key]


class ResizeLongestSide:
    """transforms.py:16-113 for GPU-resident uint8 frames (coords / boxes helpers are host arithmetic, unchanged in meaning)."""

    def __init__(self, target_length: int) -> None:
        self.target_length = target_length

    @staticmethod
    def get_preprocess_shape(oldh: int, oldw: int, long_side_length: int):
        scale = long_side_length * 1.0 / max(oldh, oldw)
        newh, neww = oldh * scale, oldw * scale
        return int(newh + 0.5), int(neww + 0.5)

    def apply_image(self, frames):
        """frames [H,W,3] or [B,H,W,3] uint8 on the GPU -> the resized uint8 frames (same rank)."""
        single = frames.dim() == 3
        _, resized, _ = preprocess_frames(frames[None] if single else frames, self.target_length, (0.0, 0.0, 0.0), (1.0, 1.0, 1.0),
                                          want_resized=True, out_dtype=torch.float32)
        return resized[0] if single else resized

    def apply_coords(self, coords, original_size):
        old_h, old_w = original_size
        new_h, new_w = self.get_preprocess_shape(original_size[0], original_size[1], self.target_length)
        coords = np.array(coords, dtype=float, copy=True)
        coords[..., 0] = coords[..., 0] * (new_w / old_w)
        coords[..., 1] = coords[..., 1] * (new_h / old_h)
        return coords

    def apply_boxes(self, boxes, original_size):
        return self.apply_coords(np.asarray(boxes).reshape(-1, 2, 2), original_size).reshape(-1, 4)


def preprocess_frames(frames, target_length, pixel_mean=PAVE_PIXEL_MEAN, pixel_std=PAVE_PIXEL_STD, out_dtype=torch.bfloat16,
                      want_resized=False):
    """frames [B,H,W,3] uint8 (GPU) -> (images [B,3,S,S] out_dtype, resized uint8 [B,Ho,Wo,3] or None, (Ho, Wo)).

    images = pad((apply_image(frame) - mean) / std) exactly as the datasets build `images` / padded `images_clip`
    (PAVE_dataset.py:217-236); (Ho, Wo) is the `resize` entry of resize_list / clip_resize_list."""
    if not frames.is_cuda or frames.dtype != torch.uint8 or frames.dim() != 4 or frames.shape[-1] != 3:
        raise RuntimeError("preprocess_frames needs a [B,H,W,3] uint8 GPU tensor (got %s %s on %s); there is no CPU fallback"
                           % (tuple(frames.shape), frames.dtype, frames.device))
    assert out_dtype in (torch.bfloat16, torch.float32)
    frames = frames.contiguous()
    B, H, W, _ = frames.shape
    S = int(target_length)
    Ho, Wo = ResizeLongestSide.get_preprocess_shape(H, W, S)
    dev = frames.device
    hb = hk = vb = vk = None
    hks = vks = 0
    tmp = None
    if Wo != W:
        hb, hk, hks = _tables(W, Wo, dev)
        tmp = torch.empty(B, H, Wo, 3, device=dev, dtype=torch.uint8)
    if Ho != H:
        vb, vk, vks = _tables(H, Ho, dev)
    out = torch.empty(B, 3, S, S, device=dev, dtype=out_dtype)
    resized = torch.empty(B, Ho, Wo, 3, device=dev, dtype=torch.uint8) if want_resized else None
    lut = _norm_lut(pixel_mean, pixel_std, dev)
    p = lambda t: t.data_ptr() if t is not None else None
    rc = _lib.lib().wg_preprocess_frames_u8(frames.data_ptr(), p(tmp), p(resized), out.data_ptr(), 1 if out_dtype == torch.bfloat16 else 0,
                                            p(hb), p(hk), hks, p(vb), p(vk), vks, B, H, W, Ho, Wo, S, lut.data_ptr(),
                                            torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(rc, "wg_preprocess_frames_u8")
    return out, resized, (Ho, Wo)
