"""``ResizeLongestSide`` of the SamPredictor boundary (``predictor.transform``; reference call sites
micro_sam/util.py:663, micro_sam/instance_segmentation.py:358, micro_sam/training/trainable_sam.py:22,36)."""
from __future__ import annotations

from copy import deepcopy
from typing import Tuple

import numpy as np
import torch
import torch.nn.functional as F


# ---- Pillow's bilinear resampling, restated ("INT (PIL fixed-point)").  ``ResizeLongestSide.apply_image`` is
# ``np.array(resize(to_pil_image(image), target_size))`` in segment_anything, i.e. ``Image.resize(size, BILINEAR)`` =
# libImaging/Resample.c: per axis a table of (first tap, number of taps, coefficients) from the triangle filter - support 1 when
# enlarging, ``scale`` when shrinking (antialiasing) -, coefficients normalised in double, converted to fixed point with 22 fractional bits,
# a horizontal pass to an 8-bit intermediate image, then a vertical pass; each pass: ``clip8((2^21 + sum(pixel * coef)) >> 22)``.
# The tables below are computed exactly as the C code does (double arithmetic, truncating int casts); ``resize_bilinear_u8_numpy`` is
# the two integer passes in numpy (the check of the tables against Pillow itself: tests/test_resize_host.py) and csrc/image.hip
# resample_u8_kernel the same passes on the device (tiles that need a resize then never touch the host after the upload).
PIL_PRECISION_BITS = 32 - 8 - 2


def pil_bilinear_tables(in_size: int, out_size: int):
    """(bounds int32 [out, 2] = (first tap, number of taps), coefs int32 [out, ksize]) of Pillow's precompute_coeffs + normalize_coeffs_8bpc
    for the full-image box (in0 = 0, in1 = in_size) and the BILINEAR filter (support 1)."""
    import math
    scale = float(in_size) / out_size
    filterscale = scale if scale >= 1.0 else 1.0
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    coefs = np.zeros((out_size, ksize), dtype=np.int32)
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
        k = np.zeros(ksize, dtype=np.float64)
        ww = 0.0
        for x in range(xmax):
            w = abs((x + xmin - center + 0.5) * ss)
            w = 1.0 - w if w < 1.0 else 0.0
            k[x] = w
            ww += w
        if ww != 0.0:
            k[:xmax] /= ww
        fixed = k * float(1 << PIL_PRECISION_BITS)
        coefs[xx] = np.where(k < 0, (-0.5 + fixed), (0.5 + fixed)).astype(np.int64).astype(np.int32)       # (int) truncates toward zero
        bounds[xx] = (xmin, xmax)
    return bounds, coefs


def _resample_axis_numpy(img: np.ndarray, out_size: int, axis: int) -> np.ndarray:
    bounds, coefs = pil_bilinear_tables(img.shape[axis], out_size)
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((out_size,) + src.shape[1:], dtype=np.uint8)
    for xx in range(out_size):
        x0, n = int(bounds[xx, 0]), int(bounds[xx, 1])
        acc = np.full(src.shape[1:], 1 << (PIL_PRECISION_BITS - 1), dtype=np.int64)
        for t in range(n):
            acc += src[x0 + t] * int(coefs[xx, t])
        out[xx] = np.clip(acc >> PIL_PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def resize_bilinear_u8_numpy(image: np.ndarray, newh: int, neww: int) -> np.ndarray:
    """``np.array(Image.fromarray(image).resize((neww, newh), Image.BILINEAR))`` as the two integer passes of Pillow (horizontal, then
    vertical; a pass is skipped when that size does not change).  uint8 [H, W] or [H, W, C]."""
    out = image
    if neww != image.shape[1]:
        out = _resample_axis_numpy(out, neww, 1)
    if newh != image.shape[0]:
        out = _resample_axis_numpy(out, newh, 0)
    return np.ascontiguousarray(out)


class ResizeLongestSide:
    def __init__(self, target_length: int) -> None:
        self.target_length = target_length

    @staticmethod
    def get_preprocess_shape(oldh: int, oldw: int, long_side_length: int) -> Tuple[int, int]:
        scale = long_side_length * 1.0 / max(oldh, oldw)
        return int(oldh * scale + 0.5), int(oldw * scale + 0.5)

    def apply_image(self, image: np.ndarray) -> np.ndarray:
        """uint8 HWC -> long side 1024 (PIL bilinear, as torchvision's resize(to_pil_image(.)) in the reference)."""
        h, w = image.shape[:2]
        newh, neww = self.get_preprocess_shape(h, w, self.target_length)
        if (newh, neww) == (h, w):
            return np.asarray(image)
        from PIL import Image
        return np.array(Image.fromarray(np.asarray(image)).resize((neww, newh), Image.BILINEAR))

    def apply_coords(self, coords: np.ndarray, original_size: Tuple[int, ...]) -> np.ndarray:
        old_h, old_w = original_size
        new_h, new_w = self.get_preprocess_shape(old_h, old_w, self.target_length)
        coords = deepcopy(coords).astype(float)
        coords[..., 0] = coords[..., 0] * (new_w / old_w)
        coords[..., 1] = coords[..., 1] * (new_h / old_h)
        return coords

    def apply_boxes(self, boxes: np.ndarray, original_size: Tuple[int, ...]) -> np.ndarray:
        return self.apply_coords(boxes.reshape(-1, 2, 2), original_size).reshape(-1, 4)

    def apply_image_torch(self, image: torch.Tensor) -> torch.Tensor:
        target = self.get_preprocess_shape(image.shape[2], image.shape[3], self.target_length)
        return F.interpolate(image, target, mode="bilinear", align_corners=False, antialias=True)

    def apply_coords_torch(self, coords: torch.Tensor, original_size: Tuple[int, ...]) -> torch.Tensor:
        old_h, old_w = original_size
        new_h, new_w = self.get_preprocess_shape(old_h, old_w, self.target_length)
        coords = deepcopy(coords).to(torch.float)
        coords[..., 0] = coords[..., 0] * (new_w / old_w)
        coords[..., 1] = coords[..., 1] * (new_h / old_h)
        return coords

    def apply_boxes_torch(self, boxes: torch.Tensor, original_size: Tuple[int, ...]) -> torch.Tensor:
        return self.apply_coords_torch(boxes.reshape(-1, 2, 2), original_size).reshape(-1, 4)
