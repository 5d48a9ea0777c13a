"""Host-side label image helpers of the prompt derivation (``micro_sam/instance_segmentation.py:1322-1379``).

The reference takes these from scikit-image 0.2x (``skimage.segmentation.find_boundaries``, ``skimage.measure.regionprops``)
and python-elf (``elf.parallel.distance_transform``: scipy's exact Euclidean distance transform per 512^2 block with a
halo); neither package is vendored in the reference, so the published algorithms are restated here over numpy / scipy.
They run on the host in the reference as well: the inputs are the three decoder maps of ONE image, the output a few
hundred point prompts - the device work starts with ``inference.batched_inference`` on those prompts.
"""
from typing import List, Tuple

import numpy as np


def _neighbourhood_extrema(labels: np.ndarray, full: bool) -> Tuple[np.ndarray, np.ndarray]:
    """Max and min of every pixel's 3x3 cross (``full=False``) or 3x3 square neighbourhood; pixels outside the image do
    not take part (grey dilation / erosion with scikit-image's border handling)."""
    p = np.pad(labels, 1, mode="edge")
    h, w = labels.shape
    views = [p[1:h + 1, 1:w + 1], p[0:h, 1:w + 1], p[2:h + 2, 1:w + 1], p[1:h + 1, 0:w], p[1:h + 1, 2:w + 2]]
    if full:
        views += [p[0:h, 0:w], p[0:h, 2:w + 2], p[2:h + 2, 0:w], p[2:h + 2, 2:w + 2]]
    hi, lo = views[0].copy(), views[0].copy()
    for v in views[1:]:
        np.maximum(hi, v, out=hi)
        np.minimum(lo, v, out=lo)
    return hi, lo


def find_outer_boundaries(labels: np.ndarray) -> np.ndarray:
    """``skimage.segmentation.find_boundaries(labels, connectivity=1, mode="outer", background=0)``: background pixels
    that touch an object through an edge, plus object pixels that touch a DIFFERENT object (8-neighbourhood)."""
    labels = np.ascontiguousarray(labels)
    if labels.dtype == bool:
        labels = labels.astype(np.uint8)
    hi, lo = _neighbourhood_extrema(labels, full=False)
    boundaries = hi != lo
    background = labels == 0
    lifted = labels.copy()
    lifted[background] = np.iinfo(labels.dtype).max            # background must not count as "another object"
    hi8, _ = _neighbourhood_extrema(labels, full=True)
    _, lo8 = _neighbourhood_extrema(lifted, full=True)
    adjacent_objects = (hi8 != lo8) & ~background
    return boundaries & (background | adjacent_objects)


def blockwise_distance_transform(mask: np.ndarray, halo=(16, 16), block_shape=(512, 512)) -> np.ndarray:
    """``elf.parallel.distance_transform(mask, halo, block_shape=...)``: the Euclidean distance of every non-zero pixel
    to the nearest zero pixel, computed independently per block on the block grown by ``halo`` (so distances are exact up
    to the halo; one block - images up to 512^2 - is the plain transform)."""
    from scipy.ndimage import distance_transform_edt
    from .tiling import Blocking
    out = np.zeros(mask.shape, dtype="float32")
    blocking = Blocking([0, 0], list(mask.shape), list(block_shape))
    for block_id in range(blocking.number_of_blocks):
        block = blocking.get_block_with_halo(block_id, list(halo))
        outer = tuple(slice(b, e) for b, e in zip(block.outer_block.begin, block.outer_block.end))
        inner = tuple(slice(b, e) for b, e in zip(block.inner_block.begin, block.inner_block.end))
        local = tuple(slice(b, e) for b, e in zip(block.inner_block_local.begin, block.inner_block_local.end))
        out[inner] = distance_transform_edt(mask[outer])[local]
    return out


def label_regions(labels: np.ndarray) -> List[Tuple[int, Tuple[slice, slice], int]]:
    """(label, bounding box slices, area) of every label > 0 in ascending label order - the part of
    ``skimage.measure.regionprops`` the callers read (``prop.label``, ``prop.bbox`` = half-open min/max, ``prop.area``)."""
    from scipy.ndimage import find_objects
    labels = np.asarray(labels)
    if labels.size == 0 or int(labels.max()) == 0:
        return []
    areas = np.bincount(labels.ravel().astype(np.int64))
    return [(idx + 1, bb, int(areas[idx + 1])) for idx, bb in enumerate(find_objects(labels.astype(np.int64))) if bb is not None]
