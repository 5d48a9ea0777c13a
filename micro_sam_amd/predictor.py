"""``SamPredictor``: the drop-in boundary object returned by ``util.get_sam_model``.

Same attributes / methods / error behaviour as upstream ``segment_anything.SamPredictor`` as used by micro_sam
(``micro_sam/util.py:655-681,915-919,1239-1256``, ``instance_segmentation.py:358-366``, ``inference.py:212-255``,
``prompt_based_segmentation.py:279-305``); everything below it runs on libmsam_hip.so.
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np
import torch

from .modeling import Sam
from .transforms import ResizeLongestSide


class SamPredictor:
    def __init__(self, sam_model: Sam) -> None:
        self.model = sam_model
        self.transform = ResizeLongestSide(sam_model.image_encoder.img_size)
        self.reset_image()

    @property
    def device(self) -> torch.device:
        return self.model.device

    def set_precision(self, mode: str) -> None:
        """"default" (16-bit throughput path), "split16" (the reference's formulation, every product on fp16 operand pairs: fp32-level
        accuracy on the 16-bit MFMA) or "strict" (the reference's formulation on fp32 kernels): ``Sam.set_precision``.  Embeddings
        computed before the switch stay what they are: call ``set_image`` / ``precompute_image_embeddings`` again after it."""
        self.model.set_precision(mode)

    def reset_image(self) -> None:
        self.model._img_state = None            # prepared decoder state of the previous embedding
        self.is_image_set = False
        self.features = None
        self.orig_h = self.orig_w = self.input_h = self.input_w = None
        self.original_size = None
        self.input_size = None

    def set_image(self, image: np.ndarray, image_format: str = "RGB") -> None:
        assert image_format in ["RGB", "BGR"], f"image_format must be in ['RGB', 'BGR'], is {image_format}."
        if image_format != self.model.image_format:
            image = image[..., ::-1]
        input_image = self.transform.apply_image(image)
        self.reset_image()
        self.original_size = tuple(image.shape[:2])
        self.input_size = tuple(input_image.shape[:2])
        u8 = torch.as_tensor(np.ascontiguousarray(input_image), device=self.device)[None]
        # Sam.preprocess (normalise + pad) is fused into the encoder's patch gather for uint8 input
        self.features = self.model.image_encoder.forward_u8(u8)
        self.is_image_set = True

    @torch.no_grad()
    def set_torch_image(self, transformed_image: torch.Tensor, original_image_size: Tuple[int, ...]) -> None:
        assert (len(transformed_image.shape) == 4 and transformed_image.shape[1] == 3
                and max(*transformed_image.shape[2:]) == self.model.image_encoder.img_size), \
            f"set_torch_image input must be BCHW with long side {self.model.image_encoder.img_size}."
        self.reset_image()
        self.original_size = tuple(original_image_size)
        self.input_size = tuple(transformed_image.shape[-2:])
        input_image = self.model.preprocess(transformed_image.to(self.device).float())
        self.features = self.model.image_encoder(input_image)
        self.is_image_set = True

    def get_image_embedding(self) -> torch.Tensor:
        if not self.is_image_set:
            raise RuntimeError("An image must be set with .set_image(...) to generate an embedding.")
        assert self.features is not None, "Features must exist if an image has been set."
        return self.features

    def predict(self, point_coords: Optional[np.ndarray] = None, point_labels: Optional[np.ndarray] = None,
                box: Optional[np.ndarray] = None, mask_input: Optional[np.ndarray] = None,
                multimask_output: bool = True, return_logits: bool = False):
        if not self.is_image_set:
            raise RuntimeError("An image must be set with .set_image(...) before mask prediction.")
        coords_torch = labels_torch = box_torch = mask_input_torch = None
        if point_coords is not None:
            assert point_labels is not None, "point_labels must be supplied if point_coords is supplied."
            point_coords = self.transform.apply_coords(point_coords, self.original_size)
            coords_torch = torch.as_tensor(point_coords, dtype=torch.float, device=self.device)[None, :, :]
            labels_torch = torch.as_tensor(point_labels, dtype=torch.int, device=self.device)[None, :]
        if box is not None:
            box = self.transform.apply_boxes(box, self.original_size)
            box_torch = torch.as_tensor(box, dtype=torch.float, device=self.device)[None, :]
        if mask_input is not None:
            mask_input_torch = torch.as_tensor(mask_input, dtype=torch.float, device=self.device)[None, :, :, :]
        masks, iou, low = self.predict_torch(coords_torch, labels_torch, box_torch, mask_input_torch, multimask_output,
                                             return_logits=return_logits)
        return masks[0].detach().cpu().numpy(), iou[0].detach().cpu().numpy(), low[0].detach().cpu().numpy()

    @torch.no_grad()
    def predict_torch(self, point_coords: Optional[torch.Tensor], point_labels: Optional[torch.Tensor],
                      boxes: Optional[torch.Tensor] = None, mask_input: Optional[torch.Tensor] = None,
                      multimask_output: bool = True, return_logits: bool = False):
        """(masks [B,C,H,W], iou [B,C], low_res [B,C,256,256]); prompts are in the 1024 input frame."""
        if not self.is_image_set:
            raise RuntimeError("An image must be set with .set_image(...) before mask prediction.")
        low, iou = self.model.decode(self.features, point_coords, point_labels, boxes, mask_input, multimask_output)
        from .ops import postprocess_masks, unpack_bits
        b, c = low.shape[:2]
        res = postprocess_masks(low.reshape(b * c, 256, 256), self.input_size, self.original_size,
                                self.model.mask_threshold, 1.0, want_logits=return_logits)
        if return_logits:
            masks = res["logits"].reshape(b, c, *self.original_size)
        else:
            masks = unpack_bits(res["bits"], self.original_size[0]).reshape(b, c, *self.original_size)
        return masks, iou, low

    @torch.no_grad()
    def predict_masks_device(self, point_coords, point_labels, boxes=None, multimask_output: bool = True,
                             stability_score_offset: float = 1.0):
        """AMG fast path: decode + fused post-processing without materialising full-resolution logits.

        Returns (iou [B,C], dict(counts, boxes, bits)) - see ops.postprocess_masks."""
        if not self.is_image_set:
            raise RuntimeError("An image must be set with .set_image(...) before mask prediction.")
        # the low-res logits go from the up-scaling kernel to the post-processing kernel as fp16 (never returned to the caller)
        low, iou = self.model.decode(self.features, point_coords, point_labels, boxes, None, multimask_output,
                                     low_res_dtype=self.model.amg_low_res_dtype)
        from .ops import postprocess_masks
        b, c = low.shape[:2]
        res = postprocess_masks(low.reshape(b * c, 256, 256), self.input_size, self.original_size,
                                self.model.mask_threshold, stability_score_offset, want_logits=False)
        return iou, res
