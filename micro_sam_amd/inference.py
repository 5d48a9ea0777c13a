"""Prompt-based batched inference behind micro_sam's ``inference.batched_inference`` (reference
``micro_sam/inference.py:22-286``): boxes and / or points in chunks of ``batch_size`` ->
``msam_decoder_forward`` -> (optional best-of-3 by predicted IoU) -> fused device post-processing
(``msam_postprocess_masks``: upsample, threshold, stability counts, boxes, bit masks) -> records or a label image.

Same names, argument meaning and error behaviour as the reference (``embedding_path`` is the zarr v2 cache of
``util.precompute_image_embeddings``).  The tiled variant (``batched_tiled_inference``) is not provided.
"""
from typing import Any, Dict, List, Optional, Union

import numpy as np
import torch

from . import ops, util
from .predictor import SamPredictor
from .transforms import ResizeLongestSide


def _validate_inputs(boxes, points, point_labels, multimasking, return_instance_segmentation, segmentation_ids,
                     logits_masks):
    """Reference inference.py:22-72 (same checks, same exceptions)."""
    if multimasking and (segmentation_ids is not None) and (not return_instance_segmentation):
        raise NotImplementedError
    if (points is None) != (point_labels is None):
        raise ValueError("If you have point prompts both `points` and `point_labels` have to be passed, "
                         "but you passed only one of them.")
    have_points = points is not None
    have_boxes = boxes is not None
    have_logits = logits_masks is not None
    if (not have_points) and (not have_boxes):
        raise ValueError("Point and/or box prompts have to be passed, you passed neither.")
    if have_points and (len(point_labels) != len(points)):
        raise ValueError(f"The number of point coordinates and labels does not match: {len(point_labels)} != {len(points)}")
    if (have_points and have_boxes) and (len(points) != len(boxes)):
        raise ValueError(f"The number of point and box prompts does not match: {len(points)} != {len(boxes)}")
    if have_logits:
        if have_points and (len(logits_masks) != len(point_labels)):
            raise ValueError(f"The number of point and logits does not match: {len(points) != len(logits_masks)}")
        elif have_boxes and (len(logits_masks) != len(boxes)):
            raise ValueError(f"The number of boxes and logits does not match: {len(boxes)} != {len(logits_masks)}")
    n_prompts = boxes.shape[0] if have_boxes else points.shape[0]
    if (segmentation_ids is not None) and (len(segmentation_ids) != n_prompts):
        raise ValueError(f"The number of segmentation ids and prompts does not match: {len(segmentation_ids)} != {n_prompts}")
    return n_prompts, have_boxes, have_points, have_logits


def _local_otsu_threshold(images: torch.Tensor, window_size: int = 31, num_bins: int = 64, eps: float = 1e-6):
    """Per-image threshold = spatial maximum of a local (windowed) Otsu threshold of the low-res logits, clamped at 0
    (reference inference.py:75-134).  Small tensors ([B,1,256,256]); evaluated with torch on the logits' device."""
    x = images.to(torch.float32)
    b, _, h, w = x.shape
    x_flat = x.view(b, -1)
    x_min = x_flat.min(dim=1).values.view(b, 1, 1, 1)
    x_max = x_flat.max(dim=1).values.view(b, 1, 1, 1)
    x_range = (x_max - x_min).clamp_min(eps)
    x_norm = (x - x_min) / x_range
    patches = torch.nn.functional.unfold(x_norm, kernel_size=window_size, padding=window_size // 2)     # [B, P, L]
    bin_idx = (patches * (num_bins - 1)).long().clamp(0, num_bins - 1)
    n_pos = patches.shape[2]
    hist = torch.zeros(b, n_pos, num_bins, device=x.device, dtype=torch.float32)
    idx = bin_idx.transpose(1, 2)
    hist.scatter_add_(2, idx, torch.ones_like(idx, dtype=hist.dtype))
    hist = hist.permute(0, 2, 1)
    p = hist / hist.sum(dim=1, keepdim=True).clamp_min(eps)
    bins = torch.arange(num_bins, device=x.device, dtype=torch.float32).view(1, num_bins, 1)
    omega1 = torch.cumsum(p, dim=1)
    mu = torch.cumsum(p * bins, dim=1)
    mu_t = mu[:, -1:, :]
    omega2 = 1.0 - omega1
    mu1 = mu / omega1.clamp_min(eps)
    mu2 = (mu_t - mu) / omega2.clamp_min(eps)
    sigma_b2 = omega1 * omega2 * (mu1 - mu2) ** 2
    t_norm = torch.argmax(sigma_b2, dim=1).to(torch.float32) / (num_bins - 1)
    thr_vals = (x_min.view(b, 1) + t_norm * x_range.view(b, 1)).clamp_min(0.0)
    return torch.amax(thr_vals.view(b, h, w), dim=(1, 2), keepdim=True)


@torch.no_grad()
def batched_inference(predictor: SamPredictor, image: Optional[np.ndarray], batch_size: int,
                      boxes: Optional[np.ndarray] = None, points: Optional[np.ndarray] = None,
                      point_labels: Optional[np.ndarray] = None, multimasking: bool = False,
                      embedding_path=None, return_instance_segmentation: bool = True,
                      segmentation_ids: Optional[list] = None, reduce_multimasking: bool = True,
                      logits_masks: Optional[torch.Tensor] = None, verbose_embeddings: bool = True,
                      mask_threshold: Optional[Union[float, str]] = None, return_highres_logits: bool = False,
                      i: Optional[int] = None) -> Union[List[Dict[str, Any]], np.ndarray]:
    """Reference inference.py:154-286.  boxes [N,4] xyxy, points [N,Np,2] xy, point_labels [N,Np] in original image
    coordinates.  Returns the list of mask records (``segmentation`` bool [H,W] device tensor, ``area``, ``bbox`` xywh,
    ``predicted_iou``, ``stability_score``, ``seg_id``, ``logits``) or, with ``return_instance_segmentation``, the
    uint32 label image of ``util.mask_data_to_segmentation(records, min_object_size=0)``."""
    n_prompts, have_boxes, have_points, have_logits = _validate_inputs(
        boxes, points, point_labels, multimasking, return_instance_segmentation, segmentation_ids, logits_masks)
    if image is None:
        predictor.get_image_embedding()          # raises RuntimeError when no embedding is set (reference :212-214)
    else:
        input_ = image if i is None else image[i]
        image_embeddings = util.precompute_image_embeddings(predictor, input_, embedding_path, verbose=verbose_embeddings)
        util.set_precomputed(predictor, image_embeddings)

    n_batches = int(np.ceil(float(n_prompts) / batch_size))
    device = predictor.device
    transform_function = ResizeLongestSide(1024)
    image_shape = predictor.original_size
    if have_boxes:
        boxes = torch.tensor(transform_function.apply_boxes(np.asarray(boxes), image_shape), dtype=torch.float32).to(device)
    if have_points:
        points = torch.tensor(transform_function.apply_coords(np.asarray(points), image_shape), dtype=torch.float32).to(device)
        point_labels = torch.tensor(np.asarray(point_labels), dtype=torch.float32).to(device)

    mask_threshold = predictor.model.mask_threshold if mask_threshold is None else mask_threshold
    height, width = int(image_shape[0]), int(image_shape[1])
    cols = {k: [] for k in ("bits", "counts", "boxes", "iou", "logits")}
    for batch_idx in range(n_batches):
        sl = slice(batch_idx * batch_size, min((batch_idx + 1) * batch_size, n_prompts))
        low, iou = predictor.model.decode(predictor.features, points[sl] if have_points else None,
                                          point_labels[sl].to(torch.int32) if have_points else None,
                                          boxes[sl] if have_boxes else None,
                                          logits_masks[sl] if have_logits else None, multimasking)
        if reduce_multimasking and multimasking:          # most likely of the three masks (reference :256-263)
            best = iou.argmax(dim=1)
            sel = torch.arange(low.shape[0], device=low.device)
            low, iou = low[sel, best][:, None], iou[sel, best][:, None]
        b, c = low.shape[:2]
        flat = low.reshape(b * c, 256, 256)
        if mask_threshold == "auto":
            # bilinear interpolation reproduces constants, so thresholding the upsampled logits at t equals
            # thresholding (logits - t) at 0 (to fp32 rounding): one fused pass with a per-mask shift
            thr = _local_otsu_threshold(low.reshape(b * c, 1, 256, 256))
            post = ops.postprocess_masks(flat - thr, predictor.input_size, image_shape, 0.0, 1.0,
                                         want_logits=return_highres_logits)
            if return_highres_logits:
                post["logits"] = post["logits"] + thr
        else:
            post = ops.postprocess_masks(flat, predictor.input_size, image_shape, float(mask_threshold), 1.0,
                                         want_logits=return_highres_logits)
        cols["bits"].append(post["bits"]); cols["counts"].append(post["counts"]); cols["boxes"].append(post["boxes"])
        cols["iou"].append(iou.reshape(-1))
        cols["logits"].append(post["logits"].reshape(b, c, height, width) if return_highres_logits else low)

    bits = torch.cat(cols["bits"]); counts = torch.cat(cols["counts"]); bxs = torch.cat(cols["boxes"])
    ious = torch.cat(cols["iou"]); logits = torch.cat(cols["logits"])
    # calculate_stability_score: |{x > t + 1}| / |{x > t - 1}| (0/0 -> nan as in the reference's tensor division)
    stability = counts[:, 0].to(torch.float32) / counts[:, 1].to(torch.float32)

    if return_instance_segmentation and segmentation_ids is None:
        # label image straight from the bit masks (identical to mask_data_to_segmentation over the records below)
        return _records_to_segmentation_device(bits, counts[:, 2], (height, width))

    seg = ops.unpack_bits(bits, height)
    xywh = bxs.clone()                                  # box_xyxy_to_xywh per record (reference :272)
    xywh[:, 2] -= xywh[:, 0]; xywh[:, 3] -= xywh[:, 1]
    xywh = xywh.tolist()
    ious_h, stab_h = ious.tolist(), stability.tolist()
    masks = [{"segmentation": seg[idx], "area": counts[idx, 2], "bbox": xywh[idx], "predicted_iou": ious_h[idx],
              "stability_score": stab_h[idx],
              "seg_id": idx + 1 if segmentation_ids is None else int(segmentation_ids[idx]),
              "logits": logits[idx]} for idx in range(seg.shape[0])]
    if return_instance_segmentation:
        masks = util.mask_data_to_segmentation(masks, min_object_size=0)
    return masks


def _records_to_segmentation_device(bits: torch.Tensor, areas: torch.Tensor, shape) -> np.ndarray:
    """``util.mask_data_to_segmentation(records, min_object_size=0)`` (defaults: label_masks=True, with_background=False,
    merge_exclusively=True, seg_id = index + 1) from bit masks on the device.

    merge_exclusively paints in area-descending order and never overwrites: that is "first painter wins", i.e. the
    device painter's "last painter wins" rule applied to the reversed order.  Equal-value components are then labelled
    and renumbered consecutively exactly like the host function (ids only label connected pieces, so the seg_id values
    themselves do not survive ``label_masks=True``: any distinct ids give the same result)."""
    h, w = int(shape[0]), int(shape[1])
    dev = bits.device
    if int(bits.shape[0]) == 0:
        return np.zeros((h, w), dtype="uint32")
    order = torch.sort(areas.to(dev), descending=True, stable=True).indices
    order = torch.flip(order, dims=(0,)).to(torch.int32).contiguous()
    painted = ops.paint_label_image(bits, order, h, w)
    roots = ops.label_components(painted).to(torch.int64)
    fg = roots >= 0
    is_root = util._root_marks(roots)
    new_id = torch.cumsum(is_root.to(torch.int64), 0)
    labels = torch.where(fg, new_id[roots.clamp(min=0)], torch.zeros_like(roots))
    return labels.reshape(h, w).to(torch.int32).cpu().numpy().astype("uint32")


def _require_tiled_embeddings(predictor, image, image_embeddings, embedding_path, tile_shape, halo, verbose_embeddings):
    """Reference inference.py:288-311."""
    if image_embeddings is None:
        assert image is not None
        assert (tile_shape is not None) and (halo is not None)
        shape = image.shape[:2]
        image_embeddings = util.precompute_image_embeddings(predictor, image, embedding_path, ndim=2, tile_shape=tile_shape,
                                                            halo=halo, verbose=verbose_embeddings)
    else:
        attrs = image_embeddings["features"].attrs
        tile_shape_, halo_ = attrs["tile_shape"], attrs["halo"]
        shape = attrs["shape"]
        if tile_shape is None:
            tile_shape = tile_shape_
        elif any(ts != ts_ for ts, ts_ in zip(tile_shape, tile_shape_)):
            raise ValueError(f"Incompatible tile shapes: {tile_shape} != {tile_shape_}")
        if halo is None:
            halo = halo_
        elif any(ts != ts_ for ts, ts_ in zip(halo, halo_)):
            raise ValueError(f"Incompatible tile shapes: {halo} != {halo_}")
    return image_embeddings, tuple(shape), tuple(tile_shape), tuple(halo)


def _merge_segmentations(this_seg, prev_seg, overlap_threshold=0.75):
    """Reference inference.py:314-332: first come, first served - the earlier tiles' labels are preserved.  (The reference also
    collects the new ids that overlap earlier ones by more than ``overlap_threshold`` but never uses that list.)"""
    captured = prev_seg != 0
    this_seg[captured] = prev_seg[captured]
    return this_seg


def _stitch_segmentation(masks, tile_ids, tiling, halo, output_shape):
    """Reference inference.py:338-355."""
    segmentation = np.zeros(output_shape, dtype="uint32")
    for tile_id, this_seg in zip(tile_ids, masks):
        tile = tiling.get_block_with_halo(tile_id, list(halo)).outer_block
        bb = tuple(slice(b, e) for b, e in zip(tile.begin, tile.end))
        if tile_id == 0:
            segmentation[bb] = this_seg
        else:
            prev_seg = segmentation[bb]
            assert prev_seg.shape == this_seg.shape, f"{tile_id}: {prev_seg.shape}, {this_seg.shape}"
            segmentation[bb] = _merge_segmentations(this_seg, prev_seg)
    return segmentation


@torch.no_grad()
def batched_tiled_inference(predictor: SamPredictor, image: Optional[np.ndarray], batch_size: int, image_embeddings=None,
                            boxes: Optional[np.ndarray] = None, points: Optional[np.ndarray] = None,
                            point_labels: Optional[np.ndarray] = None, multimasking: bool = False, embedding_path=None,
                            return_instance_segmentation: bool = True, reduce_multimasking: bool = True,
                            logits_masks: Optional[torch.Tensor] = None, verbose_embeddings: bool = True,
                            mask_threshold: Optional[Union[float, str]] = None, tile_shape=None, halo=None,
                            optimize_memory: bool = False, i: Optional[int] = None, **nms_kwargs):
    """Reference ``inference.batched_tiled_inference`` (micro_sam/inference.py:358-538): prompts are assigned to the tile that
    contains their centre, every tile runs ``batched_inference`` on its own embedding, the per-tile records are merged through
    their ``global_bbox`` (or, with ``optimize_memory``, reduced per tile by ``util.apply_nms`` and stitched)."""
    from .tiling import Blocking
    segmentation_ids = None
    n_prompts, have_boxes, have_points, have_logits = _validate_inputs(
        boxes, points, point_labels, multimasking, return_instance_segmentation, segmentation_ids, logits_masks)
    if have_logits:
        raise NotImplementedError
    image_embeddings, shape, tile_shape, halo = _require_tiled_embeddings(
        predictor, image, image_embeddings, embedding_path, tile_shape, halo, verbose_embeddings)
    tiling = Blocking([0, 0], shape, tile_shape)
    box_to_tile, point_to_tile, label_to_tile = {}, {}, {}
    tile_ids = []
    for prompt_id in range(n_prompts):
        this_tile_id = None
        if have_boxes:
            box = boxes[prompt_id]
            center = np.array([(box[1] + box[3]) / 2, (box[0] + box[2]) / 2]).round().astype("int").tolist()
            this_tile_id = tiling.coordinates_to_block_id(center)
            tile = tiling.get_block_with_halo(this_tile_id, list(halo)).outer_block
            offset, this_tile_shape = tile.begin, tile.shape
            box_in_tile = np.array([max(box[1] - offset[0], 0), max(box[0] - offset[1], 0),
                                    min(box[3] - offset[0], this_tile_shape[0]), min(box[2] - offset[1], this_tile_shape[1])])[None]
            # (the reference stores the tile-local box in [MIN_Y, MIN_X, MAX_Y, MAX_X] order and hands it to batched_inference,
            # which reads boxes as [MIN_X, MIN_Y, MAX_X, MAX_Y]: mirrored as is)
            box_to_tile[this_tile_id] = np.concatenate([box_to_tile[this_tile_id], box_in_tile]) if this_tile_id in box_to_tile \
                else box_in_tile
        if have_points:
            point = points[prompt_id, 0][::-1].round().astype("int").tolist()
            if this_tile_id is None:
                this_tile_id = tiling.coordinates_to_block_id(point)
            else:
                assert this_tile_id == tiling.coordinates_to_block_id(point)
            tile = tiling.get_block_with_halo(this_tile_id, list(halo)).outer_block
            point_in_tile = (points[prompt_id, 0] - np.array(tile.begin)[::-1])[None, None]
            label_in_tile = point_labels[prompt_id][None]
            if this_tile_id in point_to_tile:
                point_to_tile[this_tile_id] = np.concatenate([point_to_tile[this_tile_id], point_in_tile])
                label_to_tile[this_tile_id] = np.concatenate([label_to_tile[this_tile_id], label_in_tile])
            else:
                point_to_tile[this_tile_id], label_to_tile[this_tile_id] = point_in_tile, label_in_tile
        tile_ids.append(this_tile_id)
    tile_ids = sorted(list(set(tile_ids)))
    masks = []
    id_offset = 0
    for tile_id in tile_ids:
        util.set_precomputed(predictor, image_embeddings, tile_id=tile_id, i=i)
        this_masks = batched_inference(
            predictor=predictor, image=None, batch_size=batch_size, boxes=box_to_tile.get(tile_id),
            points=point_to_tile.get(tile_id), point_labels=label_to_tile.get(tile_id), multimasking=multimasking,
            return_instance_segmentation=False, segmentation_ids=segmentation_ids, reduce_multimasking=reduce_multimasking,
            logits_masks=None, mask_threshold=mask_threshold)
        if optimize_memory:
            segmentation = util.apply_nms(this_masks, **nms_kwargs)
            fg_mask = segmentation != 0
            segmentation[fg_mask] += id_offset
            id_offset = segmentation.max()
            masks.append(segmentation)
        else:
            tile = tiling.get_block_with_halo(tile_id, list(halo)).outer_block
            offset = np.array(tile.begin[::-1] + [0, 0])
            masks.extend([{**mask, "global_bbox": (np.array(mask["bbox"]) + offset).tolist()} for mask in this_masks])
    if optimize_memory:
        return _stitch_segmentation(masks, tile_ids, tiling, halo, output_shape=shape)
    if return_instance_segmentation:
        masks = util.mask_data_to_segmentation(masks, shape=shape, min_object_size=0)
    return masks
