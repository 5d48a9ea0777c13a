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

The *numbering* of connected components follows ``elf.parallel.label`` (python-elf >= 0.9, ``setup.cfg:55``; absent from the
container, so restated from its published algorithm, ``elf/parallel/label.py``): (1) every block of ``block_shape`` - (512, 512) at
the call site ``micro_sam/util.py:1834-1838`` - is labelled on its own with ``skimage.measure.label`` (components of equal non-zero
value, 4-connectivity, numbered in raster order of their first pixel), (2) block b's labels are shifted by the running sum of the
maximal labels of the blocks before it (block raster order), (3) labels that touch across a block face with equal input value
are united (union-find), (4) the united labelling is made consecutive in order of FIRST OCCURRENCE over the ascending provisional
labels (``vigra.analysis.relabelConsecutive(ufd.find(arange), keep_zeros=True)``), which makes the outcome independent of the
union-find's choice of representatives: a component's id is the rank of its smallest provisional label.
``label_components_literal`` is that procedure step by step; ``label_components`` is its closed form (rank of the component's first
pixel in block-major order) and the two are compared on random label images incl. ragged block grids (tests/test_oracle_amg.py).
For an image of up to 512 x 512 (one block: the reference's own test fixtures) this is plain raster order.  elf / nifty / vigra are
not importable here and the reference's tests are permutation invariant: the scheme is pinned to the published
algorithm, not to a run of the library.
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


def remove_small_regions(mask: np.ndarray, area_thresh: float, mode: str) -> Tuple[np.ndarray, bool]:
    """segment_anything.utils.amg.remove_small_regions (call sites micro_sam/instance_segmentation.py:156-158): upstream
    labels ``correct_holes ^ mask`` with cv2.connectedComponentsWithStats(..., 8) - cv2 is absent, 8-connected labelling
    through scipy gives the same component sets and sizes (the label numbering only matters for the size tie-break of
    "keep the largest", raster order of the first pixel in both).  PARITY UNPINNED against cv2 itself."""
    from scipy import ndimage
    assert mode in ["holes", "islands"]
    correct_holes = mode == "holes"
    working_mask = (correct_holes ^ mask).astype(np.uint8)
    regions, n_labels = ndimage.label(working_mask, structure=np.ones((3, 3)))
    sizes = np.array([(regions == i).sum() for i in range(1, n_labels + 1)])
    small_regions = [i + 1 for i, s in enumerate(sizes) if s < area_thresh]
    if len(small_regions) == 0:
        return mask, False
    fill_labels = [0] + small_regions
    if not correct_holes:
        fill_labels = [i for i in range(n_labels + 1) if i not in fill_labels]
        if len(fill_labels) == 0:
            fill_labels = [int(np.argmax(sizes)) + 1]
    mask = np.isin(regions, fill_labels)
    return mask, True


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

LABEL_BLOCK = 512          # block_shape of the reference's elf.parallel.label call (micro_sam/util.py:1834)


def block_major_keys(h: int, w: int, block: int = LABEL_BLOCK) -> np.ndarray:
    """int64 [h, w]: position of every pixel in block-major order (blocks in raster order, raster order inside a block)."""
    yy, xx = np.mgrid[0:h, 0:w]
    by, bx = yy // block, xx // block
    bh = np.minimum(block, h - by * block)
    bw = np.minimum(block, w - bx * block)
    return (by * block) * w + (bx * block) * bh + (yy % block) * bw + (xx % block)


def label_components(seg: np.ndarray, block: int = LABEL_BLOCK) -> np.ndarray:
    """Connected components (4-connectivity) of equal non-zero value, numbered 1.. as ``elf.parallel.label(block_shape=(512, 512))``
    numbers them (module docstring): by the block-major position of each component's first pixel."""
    h, w = seg.shape
    n = h * w
    key = block_major_keys(h, w, block)
    parent = np.arange(n, dtype=np.int64)                  # union-find over KEYS: the smaller key is the root
    right = (seg[:, 1:] == seg[:, :-1]) & (seg[:, 1:] != 0)
    down = (seg[1:, :] == seg[:-1, :]) & (seg[1:, :] != 0)
    ea = np.concatenate([key[:, :-1][right], key[:-1, :][down]])
    eb = np.concatenate([key[:, 1:][right], key[1:, :][down]])
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
    roots = parent[key.reshape(-1)]      # per pixel: key of its component's first pixel
    flat = seg.reshape(-1)
    out = np.zeros(n, dtype=seg.dtype)
    fg = flat != 0
    root_ids = np.unique(roots[fg])      # ascending keys == the reference's numbering order
    out[fg] = (np.searchsorted(root_ids, roots[fg]) + 1).astype(seg.dtype)
    return out.reshape(h, w)


def label_components_literal(seg: np.ndarray, block: int = LABEL_BLOCK) -> np.ndarray:
    """``elf.parallel.label`` step by step (module docstring): per-block labels with running offsets, union across block faces,
    consecutive relabelling in order of first occurrence over the provisional labels.  Slow; the check of ``label_components``."""
    h, w = seg.shape
    prov = np.zeros((h, w), dtype=np.int64)
    offset = 0
    for y0 in range(0, h, block):                          # (1) + (2): block raster order
        for x0 in range(0, w, block):
            sub = label_components(seg[y0:y0 + block, x0:x0 + block], block=1 << 30).astype(np.int64)      # one block: raster order
            prov[y0:y0 + block, x0:x0 + block] = np.where(sub != 0, sub + offset, 0)
            offset += int(sub.max())
    parent = list(range(offset + 1))                       # (3): deliberately NOT union-by-minimum - any representative will do

    def find(a):
        while parent[a] != a:
            parent[a] = parent[parent[a]]
            a = parent[a]
        return a

    def faces(la, lb, va, vb):
        for a, b in {(int(a), int(b)) for a, b, x, y in zip(la, lb, va, vb) if a != 0 and b != 0 and x == y}:
            ra, rb = find(a), find(b)
            if ra != rb:
                parent[min(ra, rb) if (ra + rb) % 2 else max(ra, rb)] = max(ra, rb) if (ra + rb) % 2 else min(ra, rb)
    for y0 in range(block, h, block):
        faces(prov[y0 - 1], prov[y0], seg[y0 - 1], seg[y0])
    for x0 in range(block, w, block):
        faces(prov[:, x0 - 1], prov[:, x0], seg[:, x0 - 1], seg[:, x0])
    mapping, nxt = {0: 0}, 1                               # (4): first occurrence over ascending provisional labels
    lut = np.zeros(offset + 1, dtype=np.int64)
    for lab in range(1, offset + 1):
        r = find(lab)
        if r not in mapping:
            mapping[r] = nxt
            nxt += 1
        lut[lab] = mapping[r]
    return lut[prov].astype(seg.dtype)


def mask_data_to_segmentation(masks: List[Dict[str, Any]], shape=None, min_object_size: int = 0,
                              max_object_size: Optional[int] = None, label_masks: bool = True,
                              with_background: bool = False, merge_exclusively: bool = True) -> np.ndarray:
    """micro_sam/util.py:1773-1848 (the tiled 'global_bbox' branch is only taken by apply_nms on tiled predictions)."""
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
        if "global_bbox" in mask_data:                    # tile-local mask placed through its boxes (util.py:1816-1824)
            x, y, bw, bh = (int(v) for v in mask_data["bbox"])
            gx, gy, gw, gh = (int(v) for v in mask_data["global_bbox"])
            local = this_mask[y:y + bh, x:x + bw]
            window = segmentation[gy:gy + gh, gx:gx + gw]               # a view: assignments land in `segmentation`
            if merge_exclusively:
                local = np.logical_and(local, window == 0)
            window[local] = this_seg_id
        else:
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
# micro_sam/util.py:1589-1770, 1851-1957: mask NMS + merge (`apply_nms`), the post-processing of the prompt-based
# generators (APG, batched_tiled_inference: SURVEY.md 8(f) rank 1).  Oracle first: the device kernels of that row do not
# exist yet.  Pinned by the reference's known-answer test test/test_util.py:81-104 (tests/test_oracle_amg.py).
# ----------------------------------------------------------------------------------------------

def _boxes_overlap(boxes_xyxy: np.ndarray) -> np.ndarray:
    """[N,N] bool: the boxes share a region of positive area (util.py:1589-1599)."""
    b = np.asarray(boxes_xyxy, dtype=np.float64)
    w = np.minimum(b[:, None, 2], b[None, :, 2]) - np.maximum(b[:, None, 0], b[None, :, 0])
    h = np.minimum(b[:, None, 3], b[None, :, 3]) - np.maximum(b[:, None, 1], b[None, :, 1])
    return (np.clip(w, 0, None) * np.clip(h, 0, None)) > 0


def mask_overlap_matrix(masks: np.ndarray, boxes_xyxy: np.ndarray, intersection_over_min: bool) -> np.ndarray:
    """Pairwise IoU (diagonal 1; util.py:1602-1621) or intersection over the smaller area (diagonal = a / (a + 1e-6);
    util.py:1624-1648) of full-size masks [N,H,W]; pairs whose boxes do not overlap are 0."""
    m = np.asarray(masks).reshape(len(masks), -1).astype(np.float64)
    inter = m @ m.T
    area = m.sum(1)
    ov = _boxes_overlap(boxes_xyxy)
    if intersection_over_min:
        out = inter / (np.minimum(area[:, None], area[None, :]) + 1e-6)
        out[~ov] = 0
        return out
    union = area[:, None] + area[None, :] - inter
    with np.errstate(invalid="ignore", divide="ignore"):
        out = np.where(ov, inter / union, 0.0)
    np.fill_diagonal(out, 1.0)
    return out


def tiled_mask_overlap_matrix(masks, boxes_xywh, global_boxes_xywh, intersection_over_min: bool) -> np.ndarray:
    """Same for tile-local masks positioned by (bbox, global_bbox) pairs: only the overlap window of the two global boxes
    is compared (util.py:1670-1722)."""
    n = len(masks)
    lb = np.asarray(boxes_xywh, dtype=np.int64)
    gb = np.asarray(global_boxes_xywh, dtype=np.int64)
    gxyxy = gb.copy()
    gxyxy[:, 2] += gxyxy[:, 0]
    gxyxy[:, 3] += gxyxy[:, 1]
    ov = _boxes_overlap(gxyxy)
    ms = [np.asarray(m.cpu() if torch.is_tensor(m) else m).astype(bool) for m in masks]
    area = np.array([m.sum() for m in ms], dtype=np.float64)
    out = np.zeros((n, n))
    for i in range(n):
        oi = gb[i, :2] - lb[i, :2]
        for j in range(i + 1, n):
            if not ov[i, j]:
                continue
            oj = gb[j, :2] - lb[j, :2]
            x0, y0 = max(gxyxy[i, 0], gxyxy[j, 0]), max(gxyxy[i, 1], gxyxy[j, 1])
            x1, y1 = min(gxyxy[i, 2], gxyxy[j, 2]), min(gxyxy[i, 3], gxyxy[j, 3])
            a = ms[i][y0 - oi[1]:y1 - oi[1], x0 - oi[0]:x1 - oi[0]]
            b = ms[j][y0 - oj[1]:y1 - oj[1], x0 - oj[0]:x1 - oj[0]]
            inter = float(np.logical_and(a, b).sum())
            den = min(area[i], area[j]) if intersection_over_min else area[i] + area[j] - inter
            out[i, j] = out[j, i] = inter / den
    np.fill_diagonal(out, 1.0)
    return out


def greedy_matrix_nms(overlap: np.ndarray, scores: np.ndarray, thresh: float) -> np.ndarray:
    """Greedy suppression on a precomputed overlap matrix, candidates in descending score order; a candidate survives a
    kept mask when overlap <= thresh (util.py:1651-1668, 1725-1746).  Returns the kept indices in score order."""
    order = list(np.argsort(-np.asarray(scores, dtype=np.float64), kind="stable"))
    keep = []
    while order:
        i = order.pop(0)
        keep.append(int(i))
        order = [j for j in order if overlap[i, j] <= thresh]
    return np.asarray(keep, dtype=np.int64)


def infer_tiled_shape(predictions) -> Tuple[int, int]:
    """util.py:1757-1766."""
    h = w = 0
    for pred in predictions:
        bx, gb = pred["bbox"], pred["global_bbox"]
        mh, mw = pred["segmentation"].shape
        h = max(h, gb[1] - bx[1] + mh)
        w = max(w, gb[0] - bx[0] + mw)
    return int(h), int(w)


def apply_nms(predictions: List[Dict[str, Any]], min_size: int, shape=None, perform_box_nms: bool = False,
              nms_thresh: float = 0.9, max_size: Optional[int] = None, intersection_over_min: bool = False) -> np.ndarray:
    """util.py:1851-1957: size filters, NMS (boxes, masks, or tile-local masks) on score = predicted_iou * stability_score,
    merge of the survivors to a label image."""
    tiled = "global_bbox" in predictions[0]
    if tiled and shape is None:
        shape = infer_tiled_shape(predictions)
    preds = [dict(p, area=int(np.asarray(p["segmentation"].cpu() if torch.is_tensor(p["segmentation"]) else p["segmentation"]).sum()))
             for p in predictions]
    if min_size > 0:
        preds = [p for p in preds if p["area"] > min_size]
    if max_size is not None:
        preds = [p for p in preds if p["area"] < max_size]
    if shape is None:
        shape = predictions[0]["segmentation"].shape
    if not preds:
        return np.zeros(shape, dtype="uint32")
    scores = np.array([float(p["predicted_iou"]) * float(p["stability_score"]) for p in preds], dtype=np.float32)
    boxes = np.array([p["global_bbox"] if tiled else p["bbox"] for p in preds], dtype=np.float32)
    xyxy = boxes.copy()
    xyxy[:, 2] += xyxy[:, 0]
    xyxy[:, 3] += xyxy[:, 1]
    if perform_box_nms:
        assert not intersection_over_min
        keep = batched_nms(torch.as_tensor(xyxy), torch.as_tensor(scores), torch.zeros(len(preds)), nms_thresh).numpy()
    elif tiled:
        ovm = tiled_mask_overlap_matrix([p["segmentation"] for p in preds], [p["bbox"] for p in preds],
                                        [p["global_bbox"] for p in preds], intersection_over_min)
        keep = greedy_matrix_nms(ovm, scores, nms_thresh)
    else:
        stack = np.stack([np.asarray(p["segmentation"].cpu() if torch.is_tensor(p["segmentation"]) else p["segmentation"])
                          for p in preds])
        keep = greedy_matrix_nms(mask_overlap_matrix(stack, xyxy, intersection_over_min), scores, nms_thresh)
    kept = [preds[int(i)] for i in keep]
    records = [{k: p[k] for k in ("segmentation", "area", "bbox") + (("global_bbox",) if tiled else ())} for p in kept]
    return mask_data_to_segmentation(records, shape=shape, min_object_size=min_size)


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
