"""``get_trainable_sam_model`` (reference ``micro_sam/training/util.py:77-151``) and ``ConvertToSamInputs`` (``:153-288``: data-loader
batch -> SAM's batched inputs)."""
from __future__ import annotations

from typing import Dict, List, Optional, Union

import numpy as np
import torch

from ..prompt_generators import PointAndBoxPromptGenerator


def identity(x):
    return x


def get_trainable_sam_model(model_type: str = "vit_b", device=None, checkpoint_path=None,
                            freeze: Optional[Union[str, List[str]]] = None, return_state: bool = False,
                            peft_kwargs: Optional[Dict] = None, flexible_load_checkpoint: bool = False, **model_kwargs):
    """Reference ``get_trainable_sam_model`` (training/util.py:77-151): the SAM of ``util.get_sam_model`` (checkpoint or the
    ``state_dict=`` extension), optional LoRA surgery of the image encoder, the parts named in ``freeze`` (``image_encoder``,
    ``prompt_encoder``, ``mask_decoder``) set to ``requires_grad = False``, wrapped in ``TrainableSAM``.  By default nothing is
    frozen and the whole model trains."""
    from .. import util as msam_util
    from ..models import peft_sam
    from .trainable_sam import TrainableSAM
    device = msam_util.get_device(device)
    _, sam, state = msam_util.get_sam_model(model_type=model_type, device=device, checkpoint_path=checkpoint_path, return_sam=True,
                                            return_state=True, flexible_load_checkpoint=flexible_load_checkpoint, **model_kwargs)
    use_peft = bool(peft_kwargs) and isinstance(peft_kwargs, dict)
    if use_peft:
        if model_type[:5] == "vit_t":
            raise ValueError("'micro-sam' does not support parameter efficient finetuning for 'mobile-sam'.")
        sam = peft_sam.PEFT_Sam(sam, **peft_kwargs).sam
        sam.to(device)
    if freeze is not None:
        freeze = freeze if isinstance(freeze, list) else [freeze]
        if use_peft and peft_kwargs.get("rank") is not None and "image_encoder" in freeze:
            raise ValueError("You cannot use PEFT & freeze the image encoder at the same time.")
        for name, param in sam.named_parameters():
            if any(name.startswith(part) for part in freeze):
                param.requires_grad = False
    model = TrainableSAM(sam)
    return (model, state) if return_state else model


def _bounding_boxes(gt: np.ndarray):
    """Per id: the bounding box [y0, x0, y1, x1) (the boxes of the reference's ``util.get_centers_and_bounding_boxes(gt, mode="p")``;
    its centres are not used on the training path: ``_get_prompt_lists`` (training/util.py:192-216) calls the prompt generator
    without centre coordinates, so every positive training point is a random pixel of the object)."""
    from scipy import ndimage
    boxes = {}
    for i, sl in enumerate(ndimage.find_objects(gt), start=1):
        if sl is not None:
            boxes[i] = (int(sl[0].start), int(sl[1].start), int(sl[0].stop), int(sl[1].stop))
    return boxes


class ConvertToSamInputs:
    def __init__(self, transform=None, dilation_strength: int = 10, box_distortion_factor: Optional[float] = None) -> None:
        self.dilation_strength = dilation_strength
        self.transform = transform
        self.box_distortion_factor = box_distortion_factor

    def _distort_boxes(self, bbox_coordinates, shape):
        out = []
        for y0, x0, y1, x1 in bbox_coordinates:
            ly, lx = y1 - y0, x1 - x0
            f = self.box_distortion_factor
            out.append([int(round(max(0, y0 - np.random.uniform(0, f) * ly))), int(round(max(0, x0 - np.random.uniform(0, f) * lx))),
                        int(round(min(shape[0], y1 + np.random.uniform(0, f) * ly))),
                        int(round(min(shape[1], x1 + np.random.uniform(0, f) * lx)))])
        return out

    def __call__(self, x, y, n_pos, n_neg, get_boxes=False, n_samples=None):
        get_points = not (n_pos == 0 and n_neg == 0)
        gen = PointAndBoxPromptGenerator(n_positive_points=n_pos, n_negative_points=n_neg, dilation_strength=self.dilation_strength,
                                         get_box_prompts=get_boxes, get_point_prompts=get_points)
        batched_inputs, batched_ids = [], []
        for image, gt in zip(x, y):
            gt = gt.squeeze().numpy().astype(np.int64)
            boxes = _bounding_boxes(gt)
            cell_ids = np.unique(gt)[1:]
            if n_samples is not None:
                cell_ids = np.sort(np.random.choice(cell_ids, size=min(n_samples, len(cell_ids)), replace=False))
            bbox = [boxes[int(i)] for i in cell_ids]
            if self.box_distortion_factor is not None:
                bbox = self._distort_boxes(bbox, gt.shape[-2:])
            one_hot = torch.from_numpy(np.stack([(gt == i) for i in cell_ids])[:, None].astype(np.float32))
            pts, lbl, bx, _ = gen(one_hot, bbox)
            rec = {"image": image, "original_size": image.shape[1:]}
            if get_boxes:
                rec["boxes"] = bx if self.transform is None else self.transform.apply_boxes_torch(bx, gt.shape[-2:])
            if get_points:
                rec["point_coords"] = pts if self.transform is None else self.transform.apply_coords_torch(pts, gt.shape[-2:])
                rec["point_labels"] = lbl
            batched_inputs.append(rec)
            batched_ids.append(cell_ids)
        return batched_inputs, batched_ids
