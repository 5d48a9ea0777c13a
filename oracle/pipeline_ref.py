"""CPU oracle for micro_sam's embed + AMG hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this.

Restates the reference's own orchestration on top of the model / AMG oracles:

* ``compute_embeddings``  = ``util._compute_embeddings_batched`` (micro_sam/util.py:654-681) and
                            ``util._compute_2d`` / ``_compute_3d`` value flow (util.py:902-1018).
* ``amg_initialize``      = ``AutomaticMaskGenerator.initialize`` -> ``_process_crop`` -> ``_process_batch``
                            -> ``AMGBase._to_mask_data`` (instance_segmentation.py:229-255,356-461),
                            single-crop (crop_n_layers=0) configuration.
* ``batched_inference``   = ``inference.batched_inference`` (micro_sam/inference.py:154-286), float thresholds.
* ``amg_generate``        = ``AutomaticMaskGenerator.generate`` -> ``_postprocess_batch`` ->
                            ``_postprocess_masks`` -> ``util.mask_data_to_segmentation``
                            (instance_segmentation.py:99-144,188-227,463-530; util.py:1773-1848).
"""
from __future__ import annotations

import os
import time
from copy import deepcopy
from typing import Any, Dict, List, Optional, Tuple

import numpy as np
import torch

from . import amg_ref as A
from . import sam_ref as S


@torch.no_grad()
def compute_embeddings(sd, images: List[np.ndarray], model_type: str = "vit_b", precision: str = "fp32"):
    """images: list of uint8 HWC (already through to_image).  Returns (features [B,256,64,64], original_sizes,
    input_sizes)."""
    tensors, original_sizes, input_sizes = [], [], []
    for image in images:
        t = torch.as_tensor(S.apply_image(image)).permute(2, 0, 1).contiguous()[None]
        original_sizes.append(tuple(image.shape[:2]))
        input_sizes.append(tuple(t.shape[-2:]))
        tensors.append(S.preprocess(t))
    x = torch.cat(tensors)
    feats = S.image_encoder(sd, x, model_type=model_type, precision=precision)
    return feats, original_sizes, input_sizes


def to_mask_data(masks: torch.Tensor, iou_preds: torch.Tensor, crop_box, original_size, points=None,
                 mask_threshold: float = 0.0, stability_score_offset: float = 1.0) -> A.MaskData:
    """AMGBase._to_mask_data, instance_segmentation.py:229-255."""
    orig_h, orig_w = original_size
    data = A.MaskData(masks=masks.flatten(0, 1), iou_preds=iou_preds.flatten(0, 1))
    if points is not None:
        data["points"] = torch.as_tensor(points.repeat(masks.shape[1], axis=0), dtype=torch.float)
    data["stability_score"] = A.calculate_stability_score(data["masks"], mask_threshold, stability_score_offset)
    data["masks"] = (data["masks"] > mask_threshold).type(torch.bool)
    data["boxes"] = A.batched_mask_to_box(data["masks"])
    data["masks"] = A.uncrop_masks(data["masks"], crop_box, orig_h, orig_w)
    data["rles"] = A.mask_to_rle(data["masks"])
    del data["masks"]
    return data


@torch.no_grad()
def amg_initialize(sd, image: np.ndarray, features: torch.Tensor, input_size, original_size,
                   points_per_side: int = 32, points_per_batch: int = 64, precision: str = "fp32",
                   stability_score_offset: float = 1.0, timings: Optional[Dict[str, float]] = None,
                   max_batches: Optional[int] = None) -> Dict[str, Any]:
    """AutomaticMaskGenerator.initialize for crop_n_layers=0 with precomputed features [1,256,64,64]."""
    original_size = tuple(image.shape[:2])
    crop_boxes, layer_idxs = A.generate_crop_boxes(original_size, 0, 512 / 1500)
    point_grids = A.build_all_layer_point_grids(points_per_side, 0, 1)
    crop_box = crop_boxes[0]
    x0, y0, x1, y1 = crop_box
    cropped_im_size = (y1 - y0, x1 - x0)
    points_scale = np.array(cropped_im_size)[None, ::-1]
    points_for_image = point_grids[0] * points_scale
    data = A.MaskData()
    nb = 0
    for (points,) in A.batch_iterator(points_per_batch, points_for_image):
        t0 = time.perf_counter()
        transformed = S.apply_coords(points, cropped_im_size)
        in_points = torch.as_tensor(transformed, dtype=torch.float)
        in_labels = torch.ones(in_points.shape[0], dtype=torch.int)
        masks, iou_preds, _ = S.predict_torch(
            sd, features, input_size, original_size, in_points[:, None, :], in_labels[:, None],
            multimask_output=True, return_logits=True, precision=precision, low_res_fp16=os.environ.get("MSAM_AMG_LOW_RES", "fp32") == "fp16")
        t1 = time.perf_counter()
        batch = to_mask_data(masks, iou_preds, crop_box, original_size, points=points,
                             stability_score_offset=stability_score_offset)
        t2 = time.perf_counter()
        if timings is not None:
            timings["decode"] = timings.get("decode", 0.0) + (t1 - t0)
            timings["mask_data"] = timings.get("mask_data", 0.0) + (t2 - t1)
        data.cat(batch)
        nb += 1
        if max_batches is not None and nb >= max_batches:
            break
    return {"crop_list": [data], "crop_boxes": crop_boxes, "original_size": original_size}


@torch.no_grad()
def amg_process_crop(sd, features: torch.Tensor, input_size, crop_box, original_size, point_grid: np.ndarray,
                     points_per_batch: int = 64, precision: str = "fp32", stability_score_offset: float = 1.0) -> A.MaskData:
    """AMG._process_crop / _process_batch for one crop or tile with a precomputed embedding
    (instance_segmentation.py:356-401; tiles: :636-650).  crop_box = [x0, y0, x1, y1] in the full image of size
    original_size; the prediction runs at the crop's own resolution, masks are padded back by uncrop_masks."""
    x0, y0, x1, y1 = crop_box
    cropped_im_size = (y1 - y0, x1 - x0)
    points_for_image = point_grid * np.array(cropped_im_size)[None, ::-1]
    data = A.MaskData()
    for (points,) in A.batch_iterator(points_per_batch, points_for_image):
        in_points = torch.as_tensor(S.apply_coords(points, cropped_im_size), dtype=torch.float)
        in_labels = torch.ones(in_points.shape[0], dtype=torch.int)
        masks, iou_preds, _ = S.predict_torch(sd, features, input_size, cropped_im_size, in_points[:, None, :],
                                              in_labels[:, None], multimask_output=True, return_logits=True,
                                              precision=precision)
        data.cat(to_mask_data(masks, iou_preds, crop_box, original_size, points=points,
                              stability_score_offset=stability_score_offset))
    return data


def postprocess_batch(data: A.MaskData, crop_box, original_size, pred_iou_thresh, stability_score_thresh,
                      box_nms_thresh) -> A.MaskData:
    """AMGBase._postprocess_batch, instance_segmentation.py:99-144."""
    orig_h, orig_w = original_size
    if pred_iou_thresh > 0.0:
        data.filter(data["iou_preds"] > pred_iou_thresh)
    if stability_score_thresh > 0.0:
        data.filter(data["stability_score"] >= stability_score_thresh)
    keep_mask = ~A.is_box_near_crop_edge(data["boxes"], crop_box, [0, 0, orig_w, orig_h])
    if not torch.all(keep_mask):
        data.filter(keep_mask)
    keep = A.batched_nms(data["boxes"].float(), data["iou_preds"], torch.zeros_like(data["boxes"][:, 0]),
                         iou_threshold=box_nms_thresh)
    data.filter(keep)
    data["boxes"] = A.uncrop_boxes_xyxy(data["boxes"], crop_box)
    data["crop_boxes"] = torch.tensor([crop_box for _ in range(len(data["rles"]))])
    data["points"] = A.uncrop_points(data["points"], crop_box)
    return data


def postprocess_small_regions(mask_data: A.MaskData, min_area: float, nms_thresh: float) -> A.MaskData:
    """AMGBase._postprocess_small_regions, instance_segmentation.py:146-186."""
    if len(mask_data["rles"]) == 0:
        return mask_data
    new_masks, scores = [], []
    for rle in mask_data["rles"]:
        mask = A.rle_to_mask(rle)
        mask, changed = A.remove_small_regions(mask, min_area, mode="holes")
        unchanged = not changed
        mask, changed = A.remove_small_regions(mask, min_area, mode="islands")
        unchanged = unchanged and not changed
        new_masks.append(torch.as_tensor(mask, dtype=torch.int).unsqueeze(0))
        scores.append(float(unchanged))
    masks = torch.cat(new_masks, dim=0)
    boxes = A.batched_mask_to_box(masks.to(torch.bool))
    keep = A.batched_nms(boxes.float(), torch.as_tensor(scores, dtype=torch.float), torch.zeros_like(boxes[:, 0]),
                         iou_threshold=nms_thresh)
    for i_mask in keep:
        if scores[i_mask] == 0.0:
            mask_data["rles"][i_mask] = A.mask_to_rle(masks[i_mask].unsqueeze(0).to(torch.bool))[0]
            mask_data["boxes"][i_mask] = boxes[i_mask] if torch.is_tensor(mask_data["boxes"]) else boxes[i_mask].numpy()
    mask_data.filter(keep)
    return mask_data


def postprocess_masks(mask_data: A.MaskData, output_mode: str = "binary_mask", min_mask_region_area: int = 0,
                      box_nms_thresh: float = 0.7, crop_nms_thresh: float = 0.7) -> List[Dict[str, Any]]:
    """AMGBase._postprocess_masks, instance_segmentation.py:188-227."""
    if min_mask_region_area > 0:
        mask_data = postprocess_small_regions(mask_data, min_mask_region_area, max(box_nms_thresh, crop_nms_thresh))
    if output_mode in ("binary_mask", "instance_segmentation"):
        mask_data["segmentations"] = [A.rle_to_mask(rle) for rle in mask_data["rles"]]
    elif output_mode == "rle":
        mask_data["segmentations"] = mask_data["rles"]
    else:
        raise ValueError(f"Invalid output mode {output_mode}.")
    anns = []
    for idx in range(len(mask_data["segmentations"])):
        anns.append({
            "segmentation": mask_data["segmentations"][idx],
            "area": A.area_from_rle(mask_data["rles"][idx]),
            "bbox": A.box_xyxy_to_xywh(mask_data["boxes"][idx]).tolist(),
            "predicted_iou": mask_data["iou_preds"][idx].item(),
            "stability_score": mask_data["stability_score"][idx].item(),
            "crop_box": A.box_xyxy_to_xywh(mask_data["crop_boxes"][idx]).tolist(),
            "point_coords": [mask_data["points"][idx].tolist()],
        })
    return anns


def amg_generate(state: Dict[str, Any], pred_iou_thresh: float = 0.88, stability_score_thresh: float = 0.95,
                 box_nms_thresh: float = 0.7, output_mode: str = "instance_segmentation",
                 with_background: bool = True, crop_nms_thresh: float = 0.7, min_mask_region_area: int = 0):
    """AutomaticMaskGenerator.generate (crops / tiles included).  instance_segmentation.py:463-530."""
    data = A.MaskData()
    for data_, crop_box in zip(state["crop_list"], state["crop_boxes"]):
        data.cat(postprocess_batch(deepcopy(data_), crop_box, state["original_size"], pred_iou_thresh,
                                   stability_score_thresh, box_nms_thresh))
    if len(state["crop_boxes"]) > 1 and len(data["crop_boxes"]) > 0:
        # duplicates between crops: prefer masks from smaller crops (instance_segmentation.py:512-520)
        scores = 1 / A.box_area(data["crop_boxes"])
        keep = A.batched_nms(data["boxes"].float(), scores, torch.zeros_like(data["boxes"][:, 0]), iou_threshold=crop_nms_thresh)
        data.filter(keep)
    data.to_numpy()
    masks = postprocess_masks(data, output_mode, min_mask_region_area, box_nms_thresh, crop_nms_thresh)
    if output_mode == "instance_segmentation":
        shape = next(iter(masks))["segmentation"].shape if len(masks) > 0 else state["original_size"]
        masks = A.mask_data_to_segmentation(masks, shape=shape, with_background=with_background,
                                            merge_exclusively=False)
    return masks


@torch.no_grad()
def batched_inference(sd, features: torch.Tensor, input_size, original_size, batch_size: int, boxes=None, points=None,
                      point_labels=None, multimasking: bool = False, return_instance_segmentation: bool = True,
                      segmentation_ids=None, reduce_multimasking: bool = True, mask_threshold: float = 0.0,
                      precision: str = "fp32"):
    """inference.batched_inference (micro_sam/inference.py:154-286) on a precomputed embedding, float threshold.
    boxes [N,4] xyxy / points [N,Np,2] / point_labels [N,Np] in original image coordinates."""
    have_boxes, have_points = boxes is not None, points is not None
    n_prompts = boxes.shape[0] if have_boxes else points.shape[0]
    n_batches = int(np.ceil(float(n_prompts) / batch_size))
    if have_boxes:
        boxes = torch.tensor(S.apply_boxes(np.asarray(boxes), original_size), dtype=torch.float32)
    if have_points:
        points = torch.tensor(S.apply_coords(np.asarray(points), original_size), dtype=torch.float32)
        point_labels = torch.tensor(np.asarray(point_labels), dtype=torch.float32)
    cols = {"masks": [], "iou_preds": [], "stability_scores": [], "boxes": [], "logits": []}
    for b in range(n_batches):
        sl = slice(b * batch_size, min((b + 1) * batch_size, n_prompts))
        bm, bi, bl = S.predict_torch(sd, features, input_size, original_size, points[sl] if have_points else None,
                                     point_labels[sl] if have_points else None, boxes[sl] if have_boxes else None, None,
                                     multimask_output=multimasking, return_logits=True, precision=precision)
        if reduce_multimasking and multimasking:
            _, mx = bi.max(axis=1)
            bm = torch.cat([bm[i, m][None] for i, m in enumerate(mx)]).unsqueeze(1)
            bi = torch.cat([bi[i, m][None] for i, m in enumerate(mx)]).unsqueeze(1)
            bl = torch.cat([bl[i, m][None] for i, m in enumerate(mx)]).unsqueeze(1)
        masks = bm.flatten(0, 1)
        cols["stability_scores"].append(A.calculate_stability_score(masks, mask_threshold, 1.0))
        masks = masks > mask_threshold
        cols["masks"].append(masks); cols["iou_preds"].append(bi.flatten(0, 1)); cols["boxes"].append(A.batched_mask_to_box(masks))
        cols["logits"].append(bl)
    cat = {k: torch.cat(v) for k, v in cols.items()}
    records = [{"segmentation": cat["masks"][i], "area": cat["masks"][i].sum(),
                "bbox": A.box_xyxy_to_xywh(cat["boxes"][i]).tolist(), "predicted_iou": cat["iou_preds"][i].item(),
                "stability_score": cat["stability_scores"][i].item(),
                "seg_id": i + 1 if segmentation_ids is None else int(segmentation_ids[i]), "logits": cat["logits"][i]}
               for i in range(len(cat["masks"]))]
    if return_instance_segmentation:
        return A.mask_data_to_segmentation(records, min_object_size=0)
    return records
