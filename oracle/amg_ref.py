"""CPU oracle for the AMG post-processing on micro_sam's hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this.

Restates, with numpy / torch on the CPU:

* ``segment_anything.utils.amg`` helpers the reference calls (third party, un-vendored, unpinned ->
  PyPI segment-anything 1.0; semantics listed in SURVEY.md Appendix B; call sites
  ``micro_sam/instance_segmentation.py:99-255,356-530``): point grids, crop boxes, batch iterator,
  stability score, box/crop-edge test, uncrop helpers, ``rle_to_mask``, ``area_from_rle``,
  ``box_xyxy_to_xywh``, ``MaskData``.
* ``micro_sam/_vendored.py:33-85``  ``batched_mask_to_box``
* ``micro_sam/_vendored.py:104-152`` ``mask_to_rle_pytorch`` (numpy back-end, which the reference's own
  ``test/test_vendored.py:63-78`` proves equal to the upstream and bioimage_cpp back-ends).
* ``torchvision.ops.batched_nms`` with all-zero category ids (= plain greedy NMS; third party).
* ``micro_sam/util.py:1773-1848`` ``mask_data_to_segmentation`` including the
  ``elf.parallel.{label,unique,isin,relabel_consecutive}`` calls.

Pinned by the reference's known-answer tests (tests/test_oracle_amg.py): box of ``mask[7:9,3:5]`` ==
``[3,7,4,8]`` (test/test_vendored.py:12-25); ``sum(counts) == H*W`` and RLE round trip
(test/test_vendored.py:63-78).

PARITY UNPINNED for one thing: the *numbering* of connected components produced by
``elf.parallel.label`` (block-wise labelling + union-find merge; elf / nifty are not in the container and
the reference's tests are permutation invariant).  This oracle numbers components in
raster order of their first pixel (what ``skimage.measure.label`` yields for a single block, i.e. for
every image up to 512x512 such as the reference's own test fixtures).  4-connectivity, components are
regions of equal non-zero value.
"""
from __future__ import annotations

import math
from copy import deepcopy
from itertools import product
from typing import Any, Dict, Generator, List, Optional, Tuple

import numpy as np
import torch


# ----------------------------------------------------------------------------------------------
# segment_anything.utils.amg subset
# ----------------------------------------------------------------------------------------------

class MaskData:
    """Dict-of-columns container (lists / ndarrays / tensors) with filter / cat / to_numpy."""

    def __init__(self, **kwargs) -> None:
        for v in kwargs.values():
            assert isinstance(v, (list, np.ndarray, torch.Tensor))
        self._stats = dict(**kwargs)

    def __setitem__(self, key: str, item: Any) -> None:
        assert isinstance(item, (list, np.ndarray, torch.Tensor))
        self._stats[key] = item

    def __delitem__(self, key: str) -> None:
        del self._stats[key]

    def __getitem__(self, key: str) -> Any:
        return self._stats[key]

    def items(self):
        return self._stats.items()

    def filter(self, keep: torch.Tensor) -> None:
        for k, v in self._stats.items():
            if v is None:
                self._stats[k] = None
            elif isinstance(v, torch.Tensor):
                self._stats[k] = v[torch.as_tensor(keep, device=v.device)]
            elif isinstance(v, np.ndarray):
                self._stats[k] = v[keep.detach().cpu().numpy()]
            elif isinstance(v, list) and keep.dtype == torch.bool:
                self._stats[k] = [a for i, a in enumerate(v) if keep[i]]
            elif isinstance(v, list):
                self._stats[k] = [v[i] for i in keep]
            else:
                raise TypeError(f"MaskData key {k} has an unsupported type {type(v)}.")

    def cat(self, new_stats: "MaskData") -> None:
        for k, v in new_stats.items():
            if k not in self._stats or self._stats[k] is None:
                self._stats[k] = deepcopy(v)
            elif isinstance(v, torch.Tensor):
                self._stats[k] = torch.cat([self._stats[k], v], dim=0)
            elif isinstance(v, np.ndarray):
                self._stats[k] = np.concatenate([self._stats[k], v], axis=0)
            elif isinstance(v, list):
                self._stats[k] = self._stats[k] + deepcopy(v)
            else:
                raise TypeError(f"MaskData key {k} has an unsupported type {type(v)}.")

    def to_numpy(self) -> None:
        for k, v in self._stats.items():
            if isinstance(v, torch.Tensor):
                self._stats[k] = v.float().detach().cpu().numpy() if v.dtype == torch.bfloat16 \
                    else v.detach().cpu().numpy()


def build_point_grid(n_per_side: int) -> np.ndarray:
    offset = 1 / (2 * n_per_side)
    points_one_side = np.linspace(offset, 1 - offset, n_per_side)
    points_x = np.tile(points_one_side[None, :], (n_per_side, 1))
    points_y = np.tile(points_one_side[:, None], (1, n_per_side))
    return np.stack([points_x, points_y], axis=-1).reshape(-1, 2)


def build_all_layer_point_grids(n_per_side: int, n_layers: int, scale_per_layer: int) -> List[np.ndarray]:
    return [build_point_grid(int(n_per_side / (scale_per_layer ** i))) for i in range(n_layers + 1)]


def generate_crop_boxes(im_size: Tuple[int, ...], n_layers: int, overlap_ratio: float):
    crop_boxes, layer_idxs = [], []
    im_h, im_w = im_size
    short_side = min(im_h, im_w)
    crop_boxes.append([0, 0, im_w, im_h])
    layer_idxs.append(0)

    def crop_len(orig_len, n_crops, overlap):
        return int(math.ceil((overlap * (n_crops - 1) + orig_len) / n_crops))

    for i_layer in range(n_layers):
        n_crops_per_side = 2 ** (i_layer + 1)
        overlap = int(overlap_ratio * short_side * (2 / n_crops_per_side))
        crop_w = crop_len(im_w, n_crops_per_side, overlap)
        crop_h = crop_len(im_h, n_crops_per_side, overlap)
        crop_box_x0 = [int((crop_w - overlap) * i) for i in range(n_crops_per_side)]
        crop_box_y0 = [int((crop_h - overlap) * i) for i in range(n_crops_per_side)]
        for x0, y0 in product(crop_box_x0, crop_box_y0):
            crop_boxes.append([x0, y0, min(x0 + crop_w, im_w), min(y0 + crop_h, im_h)])
            layer_idxs.append(i_layer + 1)
    return crop_boxes, layer_idxs


def batch_iterator(batch_size: int, *args) -> Generator[List[Any], None, None]:
    assert len(args) > 0 and all(len(a) == len(args[0]) for a in args)
    n_batches = len(args[0]) // batch_size + int(len(args[0]) % batch_size != 0)
    for b in range(n_batches):
        yield [arg[b * batch_size: (b + 1) * batch_size] for arg in args]


def calculate_stability_score(masks: torch.Tensor, mask_threshold: float, threshold_offset: float) -> torch.Tensor:
    inter = (masks > (mask_threshold + threshold_offset)).sum(-1, dtype=torch.int16).sum(-1, dtype=torch.int32)
    union = (masks > (mask_threshold - threshold_offset)).sum(-1, dtype=torch.int16).sum(-1, dtype=torch.int32)
    return inter / union


def is_box_near_crop_edge(boxes: torch.Tensor, crop_box: List[int], orig_box: List[int], atol: float = 20.0):
    crop_box_torch = torch.as_tensor(crop_box, dtype=torch.float, device=boxes.device)
    orig_box_torch = torch.as_tensor(orig_box, dtype=torch.float, device=boxes.device)
    boxes = uncrop_boxes_xyxy(boxes, crop_box).float()
    near_crop_edge = torch.isclose(boxes, crop_box_torch[None, :], atol=atol, rtol=0)
    near_image_edge = torch.isclose(boxes, orig_box_torch[None, :], atol=atol, rtol=0)
    near_crop_edge = torch.logical_and(near_crop_edge, ~near_image_edge)
    return torch.any(near_crop_edge, dim=1)


def uncrop_boxes_xyxy(boxes: torch.Tensor, crop_box: List[int]) -> torch.Tensor:
    x0, y0, _, _ = crop_box
    offset = torch.tensor([[x0, y0, x0, y0]], device=boxes.device)
    if len(boxes.shape) == 3:
        offset = offset.unsqueeze(1)
    return boxes + offset


def uncrop_points(points: torch.Tensor, crop_box: List[int]) -> torch.Tensor:
    x0, y0, _, _ = crop_box
    offset = torch.tensor([[x0, y0]], device=points.device)
    if len(points.shape) == 3:
        offset = offset.unsqueeze(1)
    return points + offset


def uncrop_masks(masks: torch.Tensor, crop_box: List[int], orig_h: int, orig_w: int) -> torch.Tensor:
    x0, y0, x1, y1 = crop_box
    if x0 == 0 and y0 == 0 and x1 == orig_w and y1 == orig_h:
        return masks
    pad_x, pad_y = orig_w - (x1 - x0), orig_h - (y1 - y0)
    pad = (x0, pad_x - x0, y0, pad_y - y0)
    return torch.nn.functional.pad(masks, pad, value=0)


def rle_to_mask(rle: Dict[str, Any]) -> np.ndarray:
    h, w = rle["size"]
    mask = np.empty(h * w, dtype=bool)
    idx = 0
    parity = False
    for count in rle["counts"]:
        mask[idx: idx + count] = parity
        idx += count
        parity ^= True
    mask = mask.reshape(w, h)
    return mask.transpose()


def area_from_rle(rle: Dict[str, Any]) -> int:
    return sum(rle["counts"][1::2])


def box_xyxy_to_xywh(box_xyxy: torch.Tensor) -> torch.Tensor:
    box_xywh = deepcopy(box_xyxy)
    box_xywh[2] = box_xywh[2] - box_xywh[0]
    box_xywh[3] = box_xywh[3] - box_xywh[1]
    return box_xywh


# ----------------------------------------------------------------------------------------------
# micro_sam/_vendored.py
# ----------------------------------------------------------------------------------------------

def batched_mask_to_box(masks: torch.Tensor) -> torch.Tensor:
    """XYXY (inclusive) box around each bool mask, [0,0,0,0] if empty.  _vendored.py:33-85."""
    assert masks.dtype == torch.bool, masks.dtype
    if torch.numel(masks) == 0:
        return torch.zeros(*masks.shape[:-2], 4, device=masks.device)
    shape = masks.shape
    h, w = shape[-2:]
    masks = masks.flatten(0, -3) if len(shape) > 2 else masks.unsqueeze(0)
    rows = masks.any(dim=-1)          # [N, h]
    cols = masks.any(dim=-2)          # [N, w]
    ar_h = torch.arange(h, dtype=torch.int)[None, :]
    ar_w = torch.arange(w, dtype=torch.int)[None, :]
    bottom = (rows * ar_h).max(dim=-1).values
    top = (rows * ar_h + h * (~rows)).to(torch.int).min(dim=-1).values
    right = (cols * ar_w).max(dim=-1).values
    left = (cols * ar_w + w * (~cols)).to(torch.int).min(dim=-1).values
    empty = (right < left) | (bottom < top)
    out = torch.stack([left, top, right, bottom], dim=-1) * (~empty).unsqueeze(-1)
    return out.reshape(*shape[:-2], 4) if len(shape) > 2 else out[0]


def compute_rle(mask_1d: np.ndarray) -> List[int]:
    """Run lengths of a 1-d 0/1 vector; first count is the number of leading zeros.  _vendored.py:104-111."""
    diffs = mask_1d[1:] != mask_1d[:-1]
    indices = np.append(np.where(diffs), len(mask_1d) - 1)
    counts = [] if mask_1d[0] == 0 else [0]
    counts += np.diff(np.append(-1, indices)).tolist()
    return counts


def mask_to_rle(tensor: torch.Tensor) -> List[Dict[str, Any]]:
    """Column-major uncompressed RLE of bool masks [N,H,W].  _vendored.py:114-152."""
    b, h, w = tensor.shape
    flat = tensor.permute(0, 2, 1).flatten(1).detach().cpu().numpy()
    return [{"size": [h, w], "counts": compute_rle(m)} for m in flat]


# ----------------------------------------------------------------------------------------------
# torchvision.ops.batched_nms with zero category ids == greedy NMS
# ----------------------------------------------------------------------------------------------

def box_area(boxes: torch.Tensor) -> torch.Tensor:
    return (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])


def nms(boxes: torch.Tensor, scores: torch.Tensor, iou_threshold: float) -> torch.Tensor:
    """Greedy NMS: stable sort by score descending, suppress IoU > threshold (no +1 in areas).

    Returns kept indices in score order, int64 (torchvision semantics)."""
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64)
    b = boxes.detach().cpu().to(torch.float32).numpy()
    s = scores.detach().cpu().to(torch.float32).numpy()
    order = np.argsort(-s, kind="stable")
    x1, y1, x2, y2 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    areas = (x2 - x1) * (y2 - y1)
    suppressed = np.zeros(len(b), dtype=bool)
    keep = []
    for _i, i in enumerate(order):
        if suppressed[i]:
            continue
        keep.append(i)
        rest = order[_i + 1:]
        xx1 = np.maximum(x1[i], x1[rest]); yy1 = np.maximum(y1[i], y1[rest])
        xx2 = np.minimum(x2[i], x2[rest]); yy2 = np.minimum(y2[i], y2[rest])
        w = np.maximum(np.float32(0), xx2 - xx1); h = np.maximum(np.float32(0), yy2 - yy1)
        inter = w * h
        iou = inter / (areas[i] + areas[rest] - inter)
        suppressed[rest[iou > iou_threshold]] = True
    return torch.as_tensor(np.array(keep, dtype=np.int64))


def batched_nms(boxes, scores, idxs, iou_threshold):
    assert not torch.any(idxs != 0), "oracle restates the zero-category call sites only"
    return nms(boxes, scores, iou_threshold)


# ----------------------------------------------------------------------------------------------
# elf.parallel.{label, unique, isin, relabel_consecutive} + util.mask_data_to_segmentation
# ----------------------------------------------------------------------------------------------

def label_components(seg: np.ndarray) -> np.ndarray:
    """Connected components (4-connectivity) of equal non-zero value, numbered 1.. in raster order of
    each component's first pixel.  (Numbering rule: see module docstring - parity unpinned.)"""
    h, w = seg.shape
    n = h * w
    flat = seg.reshape(-1)
    parent = np.arange(n, dtype=np.int64)
    # iterative min-label propagation with pointer jumping (vectorised union-find)
    idx = np.arange(n, dtype=np.int64).reshape(h, w)
    right = (seg[:, 1:] == seg[:, :-1]) & (seg[:, 1:] != 0)
    down = (seg[1:, :] == seg[:-1, :]) & (seg[1:, :] != 0)
    ea = np.concatenate([idx[:, :-1][right], idx[:-1, :][down]])
    eb = np.concatenate([idx[:, 1:][right], idx[1:, :][down]])
    while True:
        pa, pb = parent[ea], parent[eb]
        lo = np.minimum(pa, pb)
        changed = (pa != pb)
        if not changed.any():
            break
        np.minimum.at(parent, pa, lo)
        np.minimum.at(parent, pb, lo)
        while True:                      # pointer jumping
            pp = parent[parent]
            if (pp == parent).all():
                break
            parent = pp
    roots = parent
    out = np.zeros(n, dtype=seg.dtype)
    fg = flat != 0
    root_ids = np.unique(roots[fg])      # sorted == raster order of the first pixel
    out[fg] = (np.searchsorted(root_ids, roots[fg]) + 1).astype(seg.dtype)
    return out.reshape(h, w)


def mask_data_to_segmentation(masks: List[Dict[str, Any]], shape=None, min_object_size: int = 0,
                              max_object_size: Optional[int] = None, label_masks: bool = True,
                              with_background: bool = False, merge_exclusively: bool = True) -> np.ndarray:
    """micro_sam/util.py:1773-1848 (without the tiled 'global_bbox' branch, which AMG never takes)."""
    masks = sorted(masks, key=(lambda x: x["area"]), reverse=True)
    if shape is None:
        shape = next(iter(masks))["segmentation"].shape
    segmentation = np.zeros(shape, dtype="uint32")
    seg_id = 1
    for mask_data in masks:
        area = mask_data["area"]
        if (area < min_object_size) or (max_object_size is not None and area > max_object_size):
            continue
        this_mask = mask_data["segmentation"]
        this_mask = this_mask.cpu().numpy() if torch.is_tensor(this_mask) else this_mask
        this_seg_id = mask_data.get("seg_id", seg_id)
        if merge_exclusively:
            this_mask = np.logical_and(this_mask, segmentation == 0)
        segmentation[this_mask] = this_seg_id
        seg_id = this_seg_id + 1
    if label_masks:
        segmentation = label_components(segmentation)
    seg_ids, sizes = np.unique(segmentation, return_counts=True)
    filter_ids = seg_ids[sizes < min_object_size]
    if with_background:
        bg_id = seg_ids[np.argmax(sizes)]
        filter_ids = np.concatenate([filter_ids, [bg_id]])
    segmentation[np.isin(segmentation, filter_ids)] = 0
    # relabel_consecutive(keep_zeros=True, start_label=0): order preserving, zero stays zero
    ids = np.unique(segmentation)
    ids = ids[ids != 0]
    lut = np.zeros(int(segmentation.max()) + 1, dtype=segmentation.dtype)
    lut[ids] = np.arange(1, len(ids) + 1, dtype=segmentation.dtype)
    return lut[segmentation]


# ----------------------------------------------------------------------------------------------
# micro_sam/util.py:618-651
# ----------------------------------------------------------------------------------------------

def to_image(image: np.ndarray) -> np.ndarray:
    """util._to_image: any 2-d / HWC input -> uint8 RGB with per-channel min-max normalisation."""
    input_ = image
    ndim = input_.ndim
    n_channels = 1 if ndim == 2 else input_.shape[-1]
    if ndim == 2:
        input_ = np.concatenate([input_[..., None]] * 3, axis=-1)
    elif ndim == 3 and n_channels == 1:
        input_ = np.concatenate([input_] * 3, axis=-1)
    elif ndim == 3 and n_channels == 2:
        input_ = np.concatenate([input_, np.zeros(input_.shape[:2] + (1,), dtype=input_.dtype)], axis=-1)
    elif ndim == 3 and n_channels == 3:
        pass
    elif ndim == 3 and n_channels > 3:
        input_ = input_[..., :3]
    else:
        raise ValueError(f"Invalid input dimensionality {ndim}.")
    input_ = input_.astype("float32")
    input_ -= input_.min(axis=(0, 1))[None, None]
    input_ /= (input_.max(axis=(0, 1))[None, None] + 1e-7)
    return np.array((input_ * 255).astype("uint8"))
