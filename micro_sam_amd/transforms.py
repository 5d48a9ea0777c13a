"""``ResizeLongestSide`` of the SamPredictor boundary (``predictor.transform``; reference call sites
micro_sam/util.py:663, micro_sam/instance_segmentation.py:358, micro_sam/training/trainable_sam.py:22,36)."""
from __future__ import annotations

from copy import deepcopy
from typing import Tuple

import numpy as np
import torch
import torch.nn.functional as F


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
