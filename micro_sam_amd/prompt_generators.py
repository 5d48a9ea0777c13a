"""Prompt generators for training (reference ``micro_sam/prompt_generators.py``): host-side sampling logic, kept as in the
reference where it is plain torch / numpy; ``PointAndBoxPromptGenerator`` is restated without kornia (absent here): the
safety border the reference obtains by a binary dilation of the object (``dilation_strength``) is computed with
``scipy.ndimage.binary_dilation``."""
from __future__ import annotations

from typing import List, Optional, Tuple

import numpy as np
import torch


class PromptGeneratorBase:
    def __call__(self, segmentation: torch.Tensor, prediction: Optional[torch.Tensor] = None, bbox_coordinates=None,
                 center_coordinates=None):
        raise NotImplementedError("PromptGeneratorBase is just a class template.")


class PointAndBoxPromptGenerator(PromptGeneratorBase):
    """Reference :58-250.  One-hot object masks [N, 1, H, W] (+ boxes [y0, x0, y1, x1]) -> point / box prompts.
    Positive points: the object's centre when ``center_coordinates`` are given (the reference's training path passes none:
    every positive point is a random object pixel), further ones random inside the object; negative points: random in the bounding
    box grown by ``dilation_strength`` outside the object dilated ``dilation_strength`` times with a 3 x 3 square; missing points are
    filled with random background pixels (label 0)."""

    def __init__(self, n_positive_points: int, n_negative_points: int, dilation_strength: int, get_point_prompts: bool = True,
                 get_box_prompts: bool = False) -> None:
        self.n_positive_points, self.n_negative_points = n_positive_points, n_negative_points
        self.dilation_strength = dilation_strength
        self.get_box_prompts, self.get_point_prompts = get_box_prompts, get_point_prompts
        if not self.get_point_prompts and not self.get_box_prompts:
            raise ValueError("You need to request box prompts, point prompts or both.")

    @staticmethod
    def _choice(mask: np.ndarray, n: int, replace_if_short: bool) -> List[Tuple[int, int]]:
        """n pixels of ``mask``: without replacement; when the mask has fewer than n pixels either with replacement (positive
        points, reference :121-125) or all of them (negative points, :160-163)."""
        ys, xs = np.where(mask)
        if n <= 0 or len(ys) == 0:
            return []
        if n > len(ys) and replace_if_short:
            idx = np.random.choice(len(ys), size=n, replace=True)
        else:
            idx = np.random.choice(len(ys), size=min(n, len(ys)), replace=False)
        return [(int(ys[i]), int(xs[i])) for i in idx]

    def __call__(self, segmentation: torch.Tensor, bbox_coordinates: List[tuple], center_coordinates: Optional[List] = None,
                 **kwargs):
        from scipy import ndimage
        seg = segmentation.numpy().astype(bool)
        coords, labels, boxes = [], [], []
        square = np.ones((3, 3), dtype=bool)      # kornia dilation with torch.ones(3, 3): Chebyshev radius d (reference :143-146)
        for i in range(seg.shape[0]):
            obj = seg[i, 0]
            y0, x0, y1, x1 = [int(v) for v in bbox_coordinates[i]]
            if self.get_box_prompts:
                boxes.append([x0, y0, x1, y1])                                   # SAM expects xyxy
            if not self.get_point_prompts:
                continue
            pts, lab = [], []
            if self.n_positive_points > 0:                                       # reference _sample_positive_points :105-134
                if center_coordinates is not None and center_coordinates[i] is not None:
                    cy, cx = center_coordinates[i]
                    pts.append((int(cy), int(cx)))
                pts += self._choice(obj, self.n_positive_points - len(pts), replace_if_short=True)
                lab += [1] * len(pts)
            if self.n_negative_points > 0:                                       # reference _sample_negative_points :136-171
                d = self.dilation_strength
                dil = ndimage.binary_dilation(obj, structure=square, iterations=d) if d > 0 else obj
                box = np.zeros_like(obj)
                box[max(y0 - d, 0): min(y1 + d, obj.shape[0]), max(x0 - d, 0): min(x1 + d, obj.shape[1])] = True
                neg = self._choice(box ^ dil, self.n_negative_points, replace_if_short=False)      # |box - dilated object|
                pts += neg
                lab += [0] * len(neg)
            short = self.n_positive_points + self.n_negative_points - len(pts)
            if short > 0:                                                        # reference _ensure_num_points :173-190:
                pts += self._choice(~obj, short, replace_if_short=False)         # random background pixels, label 0
                lab += [0] * short
            coords.append([[x, y] for y, x in pts])                              # SAM expects (x, y)
            labels.append(lab)
        point_prompts = torch.tensor(coords, dtype=torch.float32) if self.get_point_prompts else None
        label_prompts = torch.tensor(labels, dtype=torch.float32) if self.get_point_prompts else None
        box_prompts = torch.tensor(boxes, dtype=torch.float32) if self.get_box_prompts else None
        return point_prompts, label_prompts, box_prompts, None


class IterativePromptGenerator(PromptGeneratorBase):
    """Reference :252-377 (2-d): per object one positive point where the prediction misses the object (or, if nothing is
    missed, where it is already right) and one negative point where it over-segments (or in the object's bounding box grown by 3 px,
    or anywhere in the background).

    Same regions, same fallbacks and the same uniform choice inside a region as the reference, but for ALL objects at once on the
    tensors' device: the reference walks the objects in Python with a ``torch.where`` (a device synchronisation and a host round trip)
    per object and region - 25 objects x 7 sub-iterations x 2 images x 3-4 regions per training step.  A uniformly random pixel of a
    region is the arg-max of i.i.d. uniform scores restricted to the region (torch's generator instead of ``np.random.choice``: the
    same distribution, a different stream)."""

    @staticmethod
    def _pick(regions: List[torch.Tensor]) -> torch.Tensor:
        """regions: bool [N, H, W] in order of preference -> int64 [N, 2] (x, y) of a uniformly random pixel of the first non-empty
        region of every object ((0, 0) if all are empty)."""
        n, h, w = regions[0].shape
        chosen = regions[-1]
        for r in reversed(regions[:-1]):
            nonempty = r.flatten(1).any(dim=1).view(n, 1, 1)
            chosen = torch.where(nonempty, r, chosen)
        scores = torch.rand((n, h * w), device=chosen.device)
        scores = torch.where(chosen.flatten(1), scores, torch.full_like(scores, -1.0))
        idx = scores.argmax(dim=1)
        return torch.stack([idx % w, idx // w], dim=1)

    def __call__(self, segmentation: torch.Tensor, prediction: torch.Tensor, **kwargs):
        assert segmentation.shape == prediction.shape, "The segmentation and prediction tensors should have the same shape."
        if segmentation.ndim != 4:
            raise ValueError("The segmentation and prediction tensors should have '4' dimensions (NUM_OBJECTS x 1 x H x W).")
        true = segmentation.to(prediction.device)[:, 0] == 1                         # [N, H, W]
        pred = prediction[:, 0] == 1
        n, h, w = true.shape
        pos_region = true & ~pred                                                  # diff == -1: missed
        neg_region = pred & ~true                                                  # diff == 1: over-segmented
        overlap = pred & true
        # negative fallback 1: the object's bounding box grown by 3 px, minus the object (reference _get_negative_locations_in_obj_bbox)
        rows, cols = true.any(dim=2), true.any(dim=1)                              # [N, H], [N, W]
        yy = torch.arange(h, device=true.device).view(1, h)
        xx = torch.arange(w, device=true.device).view(1, w)
        big = max(h, w) + 8
        y0 = torch.where(rows, yy, big).amin(dim=1, keepdim=True); y1 = torch.where(rows, yy, -big).amax(dim=1, keepdim=True) + 1
        x0 = torch.where(cols, xx, big).amin(dim=1, keepdim=True); x1 = torch.where(cols, xx, -big).amax(dim=1, keepdim=True) + 1
        in_y = (yy >= (y0 - 3).clamp(min=0)) & (yy < (y1 + 3).clamp(max=h))
        in_x = (xx >= (x0 - 3).clamp(min=0)) & (xx < (x1 + 3).clamp(max=w))
        bbox_ring = (in_y.view(n, h, 1) & in_x.view(n, 1, w)) ^ true               # |bbox mask - object|
        pc = self._pick([pos_region, overlap, true])
        nc = self._pick([neg_region, bbox_ring, ~true])
        coords = torch.stack([pc, nc], dim=1).cpu()                                # [N, 2, 2]: (positive, negative) x (x, y)
        labels = torch.tensor([[1, 0]]).expand(n, 2).clone()
        return coords, labels, None, None
