"""Prompt-based segmentation behind the reference's API (``micro_sam/prompt_based_segmentation.py``; SURVEY.md 8(f) rank 4,
the interactive path): ``segment_from_points / _mask / _box / _box_and_points`` turn annotator prompts - in image
coordinates, (y, x) / [y0, x0, y1, x1] - into ONE call of ``SamPredictor.predict`` (prompt encoder + mask decoder +
post-processing on libmsam_hip.so) on the precomputed embedding of the image or of the tile that holds the prompt.

All arithmetic in this module is host-side prompt bookkeeping on a handful of coordinates or one mask, as in the reference.
Restated third-party pieces (none is vendored in the reference): ``skimage.segmentation.find_boundaries`` (exact,
``_label_image_ops``), ``bioimage_cpp.distance.distance_transform`` (scipy's exact EDT), ``bioimage_cpp.filters
.gaussian_smoothing`` (scipy's ``gaussian_filter``; the window truncation of the original is not known here) and
``skimage.feature.peak_local_max`` (``_peak_local_max``) - the last two only feed ``use_points=True`` of
``segment_from_mask``, a non-default option, and are UNPINNED against the originals.
"""
from __future__ import annotations

import warnings
from typing import Optional, Tuple

import numpy as np
import torch

from . import util
from .predictor import SamPredictor
from .transforms import ResizeLongestSide


# ------------------------------------------------------------------------------------------ prompts from a mask

def _process_box(box, shape, original_size=None, box_extension=0):
    """[y0, x0, y1, x1] -> SAM's XYXY, grown by ``box_extension`` (pixels if >= 1, else a fraction of the side lengths),
    clipped to ``shape``, optionally rescaled from a 256^2 mask frame to ``original_size`` (reference :123-145)."""
    if box_extension == 0:
        ext_y = ext_x = 0
    elif box_extension >= 1:
        ext_y = ext_x = box_extension
    else:
        ext_y, ext_x = box_extension * (box[2] - box[0]), box_extension * (box[3] - box[1])
    xyxy = np.array([max(box[1] - ext_x, 0), max(box[0] - ext_y, 0),
                     min(box[3] + ext_x, shape[1]), min(box[2] + ext_y, shape[0])])
    if original_size is not None:
        xyxy = ResizeLongestSide(max(original_size)).apply_boxes(xyxy[None], (256, 256)).squeeze()
    return np.round(xyxy).astype(int)


def _compute_box_from_mask(mask, original_size=None, box_extension=0):
    """Half-open bounding box of ``mask == 1`` as a box prompt (reference :30-35)."""
    ys, xs = np.where(mask == 1)
    box = np.array([ys.min(), xs.min(), ys.max() + 1, xs.max() + 1])
    return _process_box(box, mask.shape, original_size=original_size, box_extension=box_extension)


def _peak_local_max(image: np.ndarray, min_distance: int) -> np.ndarray:
    """``skimage.feature.peak_local_max(image, min_distance=..., exclude_border=False)``: pixels that equal the maximum
    of their (2 d + 1)^2 window and exceed the image minimum, strongest first, thinned greedily so that no two peaks are
    within ``min_distance`` (Chebyshev)."""
    from scipy import ndimage as ndi
    size = 2 * min_distance + 1
    is_peak = image == ndi.maximum_filter(image, size=size, mode="nearest")
    if is_peak.all():                                           # a constant image has no peak
        is_peak[:] = False
    is_peak &= image > image.min()
    coords = np.argwhere(is_peak)
    coords = coords[np.argsort(-image[is_peak], kind="stable")]
    kept = []
    for c in coords:
        if all(np.abs(c - k).max() > min_distance for k in kept):
            kept.append(c)
    return np.array(kept, dtype=np.int64).reshape(-1, 2)


def _compute_points_from_mask(mask, original_size, box_extension, use_single_point=False):
    """Positive points at the maxima of the smoothed distance to the object boundary inside the object, negative points
    at the maxima outside of it, within the (extended) bounding box (reference :39-81).  Returns ([N,2] XY, [N] labels)."""
    from scipy import ndimage as ndi
    from ._label_image_ops import find_outer_boundaries
    box = _compute_box_from_mask(mask, box_extension=box_extension)
    window = (slice(box[1], box[3]), slice(box[0], box[2]))
    offset = np.array([box[1], box[0]])
    cropped = mask[window]
    distances = ndi.gaussian_filter(ndi.distance_transform_edt(~find_outer_boundaries(cropped)).astype("float32"), sigma=1.0)
    inside = cropped.astype(bool)
    inner = np.where(inside, distances, 0.0)
    if use_single_point:
        center = np.array(np.unravel_index(inner.argmax(), inner.shape))
        return (center + offset)[None][:, ::-1], np.ones(1, dtype="uint8")
    outer = np.where(inside, 0.0, distances)
    positives, negatives = _peak_local_max(inner, min_distance=3), _peak_local_max(outer, min_distance=5)
    coords = np.concatenate([positives, negatives]).astype("float64") + offset
    if original_size is not None:
        coords *= np.array([original_size[0] / float(mask.shape[0]), original_size[1] / float(mask.shape[1])])[None]
    labels = np.concatenate([np.ones(len(positives), dtype="uint8"), np.zeros(len(negatives), dtype="uint8")])
    return coords[:, ::-1], labels


def _compute_logits_from_mask(mask, eps=1e-3):
    """Binary mask -> SAM's [1, 256, 256] mask-prompt logits: the BINARY mask is resized (longest side -> 256, bilinear with
    antialiasing as ``ResizeLongestSide.apply_image_torch``), zero-padded to the square and re-binarised at 0.5; inside
    logit(1 - eps), outside logit(eps) (reference :84-115)."""
    assert mask.ndim == 2
    side = 256
    binary = (mask == 1).astype("float32")
    if binary.shape != (side, side):
        binary = ResizeLongestSide(side).apply_image_torch(torch.from_numpy(binary[None, None])).numpy().squeeze()
        if binary.shape != (side, side):
            h, w = binary.shape
            binary = np.pad(binary, ((0, side - h), (0, side - w)), mode="constant", constant_values=0)
    hi, lo = np.log((1 - eps) / eps), np.log(eps / (1 - eps))
    logits = np.where(binary > 0.5, hi, lo).astype("float32")[None]
    assert logits.shape == (1, side, side), f"{logits.shape}"
    return logits


# ------------------------------------------------------------------------------------------ prompts -> tile

def _outer_tile(center, shape, tile_shape, halo):
    from .tiling import Blocking
    tiling = Blocking([0, 0], shape, tile_shape)
    tile_id = tiling.coordinates_to_block_id(center)
    return tile_id, tiling.get_block_with_halo(tile_id, list(halo)).outer_block


def _points_to_tile(prompts, shape, tile_shape, halo):
    """Tile that contains the mean of the points; points in tile coordinates, points outside the tile are dropped with a
    warning (reference :150-178)."""
    points, labels = prompts
    tile_id, tile = _outer_tile(np.mean(points, axis=0).round().astype("int").tolist(), shape, tile_shape, halo)
    local = points - np.array(tile.begin)
    inside = ((local >= 0) & (local < np.array(tile.shape))).all(axis=1)
    if not inside.all():
        warnings.warn(f"{(~inside).sum()} points were not in the tile and are dropped")
        local, labels = local[inside], labels[inside]
    return tile_id, tile, (local, labels)


def _box_to_tile(box, shape, tile_shape, halo):
    """Tile that contains the box centre; the box clipped to the tile, in tile coordinates (reference :181-197)."""
    center = np.array([(box[0] + box[2]) / 2, (box[1] + box[3]) / 2]).round().astype("int").tolist()
    tile_id, tile = _outer_tile(center, shape, tile_shape, halo)
    (oy, ox), (th, tw) = tile.begin, tile.shape
    return tile_id, tile, np.array([max(box[0] - oy, 0), max(box[1] - ox, 0), min(box[2] - oy, th), min(box[3] - ox, tw)])


def _mask_to_tile(mask, shape, tile_shape, halo):
    """Tile that contains the mask's centre of mass; the mask cropped to it (reference :200-211)."""
    ys, xs = np.where(mask)
    tile_id, tile = _outer_tile(np.array([np.mean(ys), np.mean(xs)]).round().astype("int").tolist(), shape, tile_shape, halo)
    return tile_id, tile, mask[tuple(slice(b, e) for b, e in zip(tile.begin, tile.end))]


def _initialize_predictor(predictor, image_embeddings, i, prompts, to_tile):
    """Point the predictor at the embedding the prompts belong to (reference :214-232): the tile chosen by ``to_tile`` for
    tiled embeddings, the image / slice ``i`` otherwise, or whatever is already set when no embeddings are passed."""
    tile = None
    if image_embeddings is not None and image_embeddings["input_size"] is None:
        attrs = image_embeddings["features"].attrs
        shape = attrs["shape"]
        tile_id, tile, prompts = to_tile(prompts, shape, attrs["tile_shape"], attrs["halo"])
        util.set_precomputed(predictor, image_embeddings, i, tile_id=tile_id)
    elif image_embeddings is not None:
        shape = image_embeddings["original_size"]
        util.set_precomputed(predictor, image_embeddings, i)
    else:
        shape = predictor.original_size
    return predictor, tile, prompts, shape


def _tile_to_full_mask(mask, shape, tile):
    """[C, th, tw] tile prediction placed into zeros of the image shape (reference :235-239)."""
    full = np.zeros(mask.shape[0:1] + tuple(shape), dtype=mask.dtype)
    full[(slice(None),) + tuple(slice(b, e) for b, e in zip(tile.begin, tile.end))] = mask
    return full


def _finish(mask, scores, logits, tile, shape, return_all):
    if tile is not None:
        mask = _tile_to_full_mask(mask, shape, tile)
    return (mask, scores, logits) if return_all else mask


# ------------------------------------------------------------------------------------------ the four entry points

def segment_from_points(predictor: SamPredictor, points: np.ndarray, labels: np.ndarray, image_embeddings=None,
                        i: Optional[int] = None, multimask_output: bool = False, return_all: bool = False,
                        use_best_multimask: Optional[bool] = None):
    """Reference :251-305.  ``points`` [N,2] (y, x), ``labels`` [N]; a single positive point is decoded with three masks
    and the one with the highest predicted IoU is returned unless told otherwise.  Returns the mask [C,H,W] (bool)."""
    predictor, tile, (points, labels), shape = _initialize_predictor(predictor, image_embeddings, i, (points, labels),
                                                                     _points_to_tile)
    if use_best_multimask is None:
        use_best_multimask = len(points) == 1 and labels[0] == 1
    mask, scores, logits = predictor.predict(point_coords=points[:, ::-1], point_labels=labels,
                                             multimask_output=bool(multimask_output or use_best_multimask))
    if use_best_multimask:
        mask = mask[np.argmax(scores)][None]
    return _finish(mask, scores, logits, tile, shape, return_all)


def segment_from_mask(predictor: SamPredictor, mask: np.ndarray, image_embeddings=None, i: Optional[int] = None,
                      use_box: bool = True, use_mask: bool = True, use_points: bool = False,
                      original_size: Optional[Tuple[int, ...]] = None, multimask_output: bool = False,
                      return_all: bool = False, return_logits: bool = False, box_extension: float = 0.0,
                      box: Optional[np.ndarray] = None, points: Optional[np.ndarray] = None,
                      labels: Optional[np.ndarray] = None, use_single_point: bool = False):
    """Reference :308-407: any combination of the mask itself (as logits), its bounding box and points sampled from it
    (or the given ``box`` / ``points``), in one ``predict`` call."""
    def to_tile(prompts, shape, tile_shape, halo):
        mask, box, points, labels = prompts
        tile_id, tile, mask = _mask_to_tile(mask, shape, tile_shape, halo)
        if points is not None:
            tile_id_points, tile, (points, labels) = _points_to_tile((points, labels), shape, tile_shape, halo)
            if tile_id_points != tile_id:
                raise RuntimeError(f"Inconsistent tile ids for mask and point prompts: {tile_id_points} != {tile_id}.")
        if box is not None:
            tile_id_box, tile, box = _box_to_tile(box, shape, tile_shape, halo)
            if tile_id_box != tile_id:
                raise RuntimeError(f"Inconsistent tile ids for mask and box prompts: {tile_id_box} != {tile_id}.")
        return tile_id, tile, (mask, box, points, labels)

    predictor, tile, (mask, box, points, labels), shape = _initialize_predictor(
        predictor, image_embeddings, i, (mask, box, points, labels), to_tile)
    nonempty = mask.sum() != 0
    if points is not None:
        if labels is None:
            raise ValueError("If points are passed you also need to pass labels.")
        point_coords, point_labels = points, labels          # (passed on as they are, as in the reference)
    elif use_points and nonempty:
        point_coords, point_labels = _compute_points_from_mask(mask, original_size=original_size, box_extension=box_extension,
                                                               use_single_point=use_single_point)
    else:
        point_coords = point_labels = None
    if box is not None:
        box = _process_box(box, mask.shape, original_size=original_size, box_extension=box_extension)
    elif use_box and nonempty:
        box = _compute_box_from_mask(mask, original_size=original_size, box_extension=box_extension)
    mask_input = _compute_logits_from_mask(mask) if use_mask else None
    mask, scores, logits = predictor.predict(point_coords=point_coords, point_labels=point_labels, mask_input=mask_input,
                                             box=box, multimask_output=multimask_output, return_logits=return_logits)
    return _finish(mask, scores, logits, tile, shape, return_all)


def segment_from_box(predictor: SamPredictor, box: np.ndarray, image_embeddings=None, i: Optional[int] = None,
                     multimask_output: bool = False, return_all: bool = False, box_extension: float = 0.0):
    """Reference :410-449.  ``box`` = [y0, x0, y1, x1]."""
    predictor, tile, box, shape = _initialize_predictor(predictor, image_embeddings, i, box, _box_to_tile)
    mask, scores, logits = predictor.predict(box=_process_box(box, shape, box_extension=box_extension),
                                             multimask_output=multimask_output)
    return _finish(mask, scores, logits, tile, shape, return_all)


def segment_from_box_and_points(predictor: SamPredictor, box: np.ndarray, points: np.ndarray, labels: np.ndarray,
                                image_embeddings=None, i: Optional[int] = None, multimask_output: bool = False,
                                return_all: bool = False):
    """Reference :452-506."""
    def to_tile(prompts, shape, tile_shape, halo):
        box, points, labels = prompts
        tile_id, tile, (points, labels) = _points_to_tile((points, labels), shape, tile_shape, halo)
        tile_id_box, tile, box = _box_to_tile(box, shape, tile_shape, halo)
        if tile_id_box != tile_id:
            raise RuntimeError(f"Inconsistent tile ids for box and point annotations: {tile_id_box} != {tile_id}.")
        return tile_id, tile, (box, points, labels)

    predictor, tile, (box, points, labels), shape = _initialize_predictor(predictor, image_embeddings, i,
                                                                         (box, points, labels), to_tile)
    mask, scores, logits = predictor.predict(point_coords=points[:, ::-1], point_labels=labels, box=_process_box(box, shape),
                                             multimask_output=multimask_output)
    return _finish(mask, scores, logits, tile, shape, return_all)
