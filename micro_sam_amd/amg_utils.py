"""Host-side helpers of automatic mask generation: the subset of ``segment_anything.utils.amg`` that micro_sam
calls (``micro_sam/instance_segmentation.py:99-255,356-530``; semantics in SURVEY.md Appendix B) plus greedy box NMS
(``torchvision.ops.batched_nms`` with one category).  These are host code in the reference as well."""
from __future__ import annotations

import math
from copy import deepcopy
from itertools import product
from typing import Any, Dict, Iterator, List, Tuple

import numpy as np
import torch


class MaskData:
    """Columnar store (lists / ndarrays / tensors of equal length) with ``filter`` / ``cat`` / ``to_numpy``."""

    _TYPES = (list, np.ndarray, torch.Tensor)

    def __init__(self, **kwargs) -> None:
        for v in kwargs.values():
            assert isinstance(v, self._TYPES), "MaskData only supports list, numpy arrays, and torch tensors."
        self._stats = dict(**kwargs)

    def __setitem__(self, key: str, item: Any) -> None:
        assert isinstance(item, self._TYPES), "MaskData only supports list, numpy arrays, and torch tensors."
        self._stats[key] = item

    def __delitem__(self, key: str) -> None:
        del self._stats[key]

    def __getitem__(self, key: str) -> Any:
        return self._stats[key]

    def __contains__(self, key: str) -> bool:
        return key in self._stats

    def items(self):
        return self._stats.items()

    def filter(self, keep: torch.Tensor) -> None:
        keep = torch.as_tensor(keep)
        for k, v in self._stats.items():
            if v is None:
                continue
            if isinstance(v, torch.Tensor):
                self._stats[k] = v[keep.to(v.device)]
            elif isinstance(v, np.ndarray):
                self._stats[k] = v[keep.detach().cpu().numpy()]
            elif keep.dtype == torch.bool:
                flags = keep.tolist()
                self._stats[k] = [a for a, f in zip(v, flags) if f]
            else:
                self._stats[k] = [v[i] for i in keep.tolist()]

    def cat(self, new_stats: "MaskData") -> None:
        for k, v in new_stats.items():
            cur = self._stats.get(k)
            if cur is None:
                self._stats[k] = deepcopy(v)
            elif isinstance(v, torch.Tensor):
                self._stats[k] = torch.cat([cur, v], dim=0)
            elif isinstance(v, np.ndarray):
                self._stats[k] = np.concatenate([cur, v], axis=0)
            elif isinstance(v, list):
                self._stats[k] = cur + deepcopy(v)
            else:
                raise TypeError(f"MaskData key {k} has an unsupported type {type(v)}.")

    def to_numpy(self) -> None:
        for k, v in self._stats.items():
            if isinstance(v, torch.Tensor):
                self._stats[k] = v.float().cpu().numpy() if v.dtype == torch.bfloat16 else v.detach().cpu().numpy()


def build_point_grid(n_per_side: int) -> np.ndarray:
    """Cell-centre grid in (0,1)^2, row-major over y, every row an (x, y) pair."""
    centres = (np.arange(n_per_side, dtype=np.float64) + 0.5) / n_per_side
    centres = np.linspace(centres[0], centres[-1], n_per_side) if n_per_side > 1 else centres
    xs, ys = np.meshgrid(centres, centres)
    return np.stack([xs, ys], axis=-1).reshape(-1, 2)


def build_all_layer_point_grids(n_per_side: int, n_layers: int, scale_per_layer: int) -> List[np.ndarray]:
    return [build_point_grid(int(n_per_side / (scale_per_layer ** i))) for i in range(n_layers + 1)]


def generate_crop_boxes(im_size: Tuple[int, ...], n_layers: int, overlap_ratio: float):
    im_h, im_w = im_size
    boxes, layers = [[0, 0, im_w, im_h]], [0]
    short = min(im_h, im_w)
    for layer in range(1, n_layers + 1):
        n = 2 ** layer
        overlap = int(overlap_ratio * short * (2 / n))
        cw = int(math.ceil((overlap * (n - 1) + im_w) / n))
        ch = int(math.ceil((overlap * (n - 1) + im_h) / n))
        x0s = [int((cw - overlap) * i) for i in range(n)]
        y0s = [int((ch - overlap) * i) for i in range(n)]
        for x0, y0 in product(x0s, y0s):
            boxes.append([x0, y0, min(x0 + cw, im_w), min(y0 + ch, im_h)])
            layers.append(layer)
    return boxes, layers


def batch_iterator(batch_size: int, *args) -> Iterator[List[Any]]:
    n = len(args[0])
    assert all(len(a) == n for a in args), "Batched iteration must have inputs of all the same size."
    for start in range(0, n, batch_size):
        yield [a[start:start + batch_size] for a in args]


_DEVICE_CONSTANTS: Dict[Any, torch.Tensor] = {}


def _device_constant(values, dtype: torch.dtype, device) -> torch.Tensor:
    """Small read-only row vector [1, n] on ``device``.  A host -> device copy of pageable memory waits for the stream
    to drain (measured: 5.6 ms per tile in ``generate``, the whole decode of the tile), so crop boxes / offsets are
    uploaded once per (values, dtype, device) and reused; callers never write to the result."""
    device = torch.device(device)
    if device.type != "cuda":
        return torch.tensor([list(values)], dtype=dtype, device=device)
    key = (tuple(int(v) for v in values), dtype, device.type, device.index)
    t = _DEVICE_CONSTANTS.get(key)
    if t is None:
        if len(_DEVICE_CONSTANTS) > 4096:
            _DEVICE_CONSTANTS.clear()
        t = torch.tensor([list(values)], dtype=dtype, device=device)
        _DEVICE_CONSTANTS[key] = t
    return t


def uncrop_boxes_xyxy(boxes: torch.Tensor, crop_box: List[int]) -> torch.Tensor:
    x0, y0 = crop_box[0], crop_box[1]
    offset = _device_constant([x0, y0, x0, y0], torch.int64, boxes.device)
    return boxes + (offset.unsqueeze(1) if boxes.dim() == 3 else offset)


def uncrop_points(points: torch.Tensor, crop_box: List[int]) -> torch.Tensor:
    offset = _device_constant([crop_box[0], crop_box[1]], torch.int64, points.device)
    return points + (offset.unsqueeze(1) if points.dim() == 3 else offset)


def is_box_near_crop_edge(boxes: torch.Tensor, crop_box: List[int], orig_box: List[int], atol: float = 20.0):
    crop_t = _device_constant(crop_box, torch.float, boxes.device)
    orig_t = _device_constant(orig_box, torch.float, boxes.device)
    b = uncrop_boxes_xyxy(boxes, crop_box).float()
    near_crop = torch.isclose(b, crop_t, atol=atol, rtol=0)
    near_img = torch.isclose(b, orig_t, atol=atol, rtol=0)
    return torch.any(near_crop & ~near_img, dim=1)


def rle_to_mask(rle: Dict[str, Any]) -> np.ndarray:
    """Uncompressed column-major RLE -> bool [h, w]."""
    h, w = rle["size"]
    counts = np.asarray(rle["counts"], dtype=np.int64)
    vals = (np.arange(len(counts)) & 1).astype(bool)
    return np.repeat(vals, counts).reshape(w, h).T


def area_from_rle(rle: Dict[str, Any]) -> int:
    return int(sum(rle["counts"][1::2]))


def box_xyxy_to_xywh(box_xyxy):
    b = deepcopy(box_xyxy)
    b[2] = b[2] - b[0]
    b[3] = b[3] - b[1]
    return b


def box_area(boxes: torch.Tensor) -> torch.Tensor:
    return (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])


def batched_nms(boxes: torch.Tensor, scores: torch.Tensor, idxs: torch.Tensor, iou_threshold: float) -> torch.Tensor:
    """Greedy NMS, torchvision semantics: stable descending score order, suppress IoU > threshold (areas without +1),
    boxes of different categories never suppress each other.  Returns kept indices in score order (int64)."""
    n = boxes.shape[0]
    if n == 0:
        return torch.empty((0,), dtype=torch.int64, device=boxes.device)
    if boxes.is_cuda and not bool((idxs != 0).any()):
        # one category (every call site of the hot path passes zeros): the device kernel (msam_box_nms: suppression bit matrix + sweep,
        # separately rounded fp32 operations in torchvision's order; tests/test_gpu_segment.py::test_box_nms_matches_oracle)
        from . import ops
        return ops.box_nms(boxes.detach().float(), scores.detach().float().to(boxes.device), iou_threshold)
    b = boxes.detach().float().cpu().numpy()
    s = scores.detach().float().cpu().numpy()
    cat = idxs.detach().cpu().numpy()
    order = np.argsort(-s, kind="stable")
    b, cat = b[order], cat[order]
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    alive = np.ones(n, dtype=bool)
    keep = []
    for i in range(n):
        if not alive[i]:
            continue
        keep.append(order[i])
        j = slice(i + 1, n)
        w = np.minimum(b[i, 2], b[j, 2]) - np.maximum(b[i, 0], b[j, 0])
        h = np.minimum(b[i, 3], b[j, 3]) - np.maximum(b[i, 1], b[j, 1])
        inter = np.maximum(w, np.float32(0)) * np.maximum(h, np.float32(0))
        iou = inter / (area[i] + area[j] - inter)
        alive[j] &= ~((iou > iou_threshold) & (cat[j] == cat[i]))
    return torch.as_tensor(np.asarray(keep, dtype=np.int64), device=boxes.device)


def remove_small_regions(mask: np.ndarray, area_thresh: float, mode: str):
    """``segment_anything.utils.amg.remove_small_regions`` (called at micro_sam/instance_segmentation.py:156-158): removes
    small disconnected regions (mode "islands") or fills small holes (mode "holes") of a binary mask; 8-connectivity.
    Returns (mask, changed).  Upstream labels with ``cv2.connectedComponentsWithStats`` (absent here); the result only
    depends on the component SETS and sizes, which ``scipy.ndimage.label`` with a full 3x3 structure reproduces."""
    from scipy import ndimage
    assert mode in ("holes", "islands")
    correct_holes = mode == "holes"
    working = np.logical_xor(correct_holes, mask)
    regions, n = ndimage.label(working, structure=np.ones((3, 3), dtype=np.uint8))
    sizes = np.bincount(regions.ravel(), minlength=n + 1)[1:]                      # label 0 = background of `working`
    small = [i + 1 for i, sz in enumerate(sizes) if sz < area_thresh]
    if len(small) == 0:
        return mask, False
    fill_labels = [0] + small
    if not correct_holes:
        fill_labels = [i for i in range(n + 1) if i not in fill_labels]
        if len(fill_labels) == 0:                                                   # every region below the threshold: keep the largest
            fill_labels = [int(np.argmax(sizes)) + 1]
    return np.isin(regions, fill_labels), True


def coco_encode_rle(uncompressed_rle: Dict[str, Any]) -> Dict[str, Any]:
    """``segment_anything.utils.amg.coco_encode_rle``: uncompressed column-major RLE -> COCO compressed RLE
    (``{"size": [h, w], "counts": str}``).  Upstream calls ``pycocotools.mask.frPyObjects`` (absent here); this is the
    published string coding of cocoapi ``rleToString`` (maskApi.c): counts are delta-coded against the count two places
    back from the fourth on, then written as little-endian 5-bit groups, bit 5 = continuation, + 48."""
    h, w = uncompressed_rle["size"]
    cnts = [int(c) for c in np.asarray(uncompressed_rle["counts"]).tolist()]
    out = []
    for i, x in enumerate(cnts):
        if i > 2:
            x -= cnts[i - 2]
        more = True
        while more:
            c = x & 0x1F
            x >>= 5
            more = (x != -1) if (c & 0x10) else (x != 0)
            if more:
                c |= 0x20
            out.append(chr(c + 48))
    return {"size": [int(h), int(w)], "counts": "".join(out)}


def coco_decode_rle(rle: Dict[str, Any]) -> Dict[str, Any]:
    """Inverse of ``coco_encode_rle`` (cocoapi ``rleFrString``): COCO compressed RLE -> uncompressed counts."""
    s = rle["counts"]
    s = s.decode("ascii") if isinstance(s, bytes) else s
    cnts, p = [], 0
    while p < len(s):
        x, k, more = 0, 0, True
        while more:
            c = ord(s[p]) - 48
            x |= (c & 0x1F) << (5 * k)
            more = bool(c & 0x20)
            p += 1; k += 1
            if not more and (c & 0x10):
                x |= -1 << (5 * k)
        if len(cnts) > 2:
            x += cnts[-2]
        cnts.append(x)
    return {"size": list(rle["size"]), "counts": cnts}


def mask_to_rle_numpy(mask: np.ndarray) -> Dict[str, Any]:
    """Uncompressed column-major RLE of ONE host mask (``micro_sam/_vendored.py:104-152`` semantics: counts start with the
    number of zeros, 0 if the mask starts with a one).  Host twin of the HIP RLE kernels for masks edited on the host."""
    h, w = mask.shape
    flat = np.asarray(mask, dtype=bool).reshape(-1, order="F")
    change = np.flatnonzero(flat[1:] != flat[:-1]) + 1
    bounds = np.concatenate([[0], change, [h * w]])
    counts = np.diff(bounds).tolist()
    if flat.size and flat[0]:
        counts = [0] + counts
    return {"size": [int(h), int(w)], "counts": counts}
