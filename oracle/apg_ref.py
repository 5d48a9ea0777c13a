"""CPU oracle for the prompt derivation of the reference's AutomaticPromptGenerator.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this.

Restates ``micro_sam/instance_segmentation.py:1322-1391`` (``_get_centers``, ``_derive_point_prompts``,
``_derive_box_prompts``) and the generate recipe of ``AutomaticPromptGenerator`` (:1394-1505) on top of the
oracle's ``batched_inference`` / ``apply_nms``.

The reference leans on three un-vendored packages here, none of which is in this container:

* scikit-image (``find_boundaries(mode="outer")``, ``regionprops``) - restated from the published algorithm with
  scipy's grey dilation / erosion, i.e. through a different route than the product's shifted-view code;
* python-elf (``elf.parallel.label``: 4-connected, raster numbering for one block - see ``amg_ref``;
  ``elf.parallel.distance_transform``: scipy's exact EDT per block + halo) - the EDT is restated a second time as a
  brute-force nearest-zero search (``brute_force_edt``) that the tests use on small cases.

PARITY UNPINNED against those packages themselves; pinned by hand-derived known answers in
``tests/test_prompt_derivation_host.py`` (single-pixel object, touching objects, the reference's three-disk fixture
of ``test/test_instance_segmentation.py:20-39``, whose derived prompts must be the disk centres).
"""
from __future__ import annotations

import numpy as np
from scipy import ndimage as ndi

from . import amg_ref as A


def find_boundaries_outer(labels: np.ndarray) -> np.ndarray:
    """skimage.segmentation.find_boundaries(labels, connectivity=1, mode="outer", background=0)."""
    labels = np.asarray(labels)
    if labels.dtype == bool:
        labels = labels.astype(np.uint8)
    cross = ndi.generate_binary_structure(2, 1)
    square = ndi.generate_binary_structure(2, 2)
    boundaries = ndi.grey_dilation(labels, footprint=cross) != ndi.grey_erosion(labels, footprint=cross)
    background = labels == 0
    inverted = labels.copy()
    inverted[background] = np.iinfo(labels.dtype).max
    adjacent = (ndi.grey_dilation(labels, footprint=square) != ndi.grey_erosion(inverted, footprint=square)) & ~background
    return boundaries & (background | adjacent)


def brute_force_edt(mask: np.ndarray) -> np.ndarray:
    """Distance of every non-zero pixel to the nearest zero pixel, O(N * Z); small cases only."""
    mask = np.asarray(mask).astype(bool)
    zy, zx = np.nonzero(~mask)
    out = np.zeros(mask.shape, dtype=np.float64)
    if len(zy) == 0:
        raise ValueError("no zero pixel")
    for y, x in zip(*np.nonzero(mask)):
        out[y, x] = np.sqrt(((zy - y) ** 2 + (zx - x) ** 2).min())
    return out


def distance_transform_blockwise(mask: np.ndarray, halo=(16, 16), block_shape=(512, 512), edt=None) -> np.ndarray:
    """elf.parallel.distance_transform: per block, the transform of the block grown by the halo (clipped to the image)."""
    edt = ndi.distance_transform_edt if edt is None else edt
    h, w = mask.shape
    out = np.zeros((h, w), dtype="float32")
    for y0 in range(0, h, block_shape[0]):
        for x0 in range(0, w, block_shape[1]):
            y1, x1 = min(y0 + block_shape[0], h), min(x0 + block_shape[1], w)
            oy0, ox0 = max(y0 - halo[0], 0), max(x0 - halo[1], 0)
            oy1, ox1 = min(y1 + halo[0], h), min(x1 + halo[1], w)
            d = edt(mask[oy0:oy1, ox0:ox1])
            out[y0:y1, x0:x1] = d[y0 - oy0:y1 - oy0, x0 - ox0:x1 - ox0]
    return out


def get_centers(segmentation: np.ndarray, avoid_image_border: bool = True, edt=None) -> np.ndarray:
    """instance_segmentation.py:1322-1355."""
    boundaries = find_boundaries_outer(segmentation) == 0
    if avoid_image_border:
        boundaries[0, :] = False
        boundaries[:, 0] = False
        boundaries[-1, :] = False
        boundaries[:, -1] = False
    distances = distance_transform_blockwise(boundaries, edt=edt)
    centers = []
    for seg_id in np.unique(segmentation):
        if seg_id == 0:
            continue
        ys, xs = np.nonzero(segmentation == seg_id)
        bb = np.s_[ys.min():ys.max() + 1, xs.min():xs.max() + 1]
        dist = distances[bb].copy()
        dist[~(segmentation[bb] == seg_id)] = 0
        c = np.unravel_index(np.argmax(dist), dist.shape)
        centers.append((c[0] + bb[0].start, c[1] + bb[1].start))
    return np.array(centers)


def derive_point_prompts(foreground, center_distances, boundary_distances, foreground_threshold: float = 0.5,
                         center_distance_threshold: float = 0.5, boundary_distance_threshold: float = 0.5, edt=None):
    """instance_segmentation.py:1358-1379."""
    bg_mask = foreground < foreground_threshold
    hmap_cc = np.logical_and(center_distances < center_distance_threshold, boundary_distances < boundary_distance_threshold)
    hmap_cc[bg_mask] = 0
    cc = A.label_components(hmap_cc.astype("uint32"))
    prompts = get_centers(cc, edt=edt)
    if len(prompts) == 0:
        return None
    return {"points": prompts[:, None, ::-1], "point_labels": np.ones((len(prompts), 1))}


def derive_box_prompts(predictions, box_extension: float):
    """instance_segmentation.py:1382-1391."""
    shape = predictions[0]["segmentation"].shape
    bboxes = [pred["bbox"] for pred in predictions]
    return {"boxes": np.array([[max(x - w * box_extension, 0), max(y - h * box_extension, 0),
                                min(x + (1 + box_extension) * w, shape[0]), min(y + (1 + box_extension) * h, shape[1])]
                               for (x, y, w, h) in bboxes])}


def apg_generate(sd, features, input_size, original_size, foreground, center_distances, boundary_distances,
                 min_size: int = 25, center_distance_threshold: float = 0.5, boundary_distance_threshold: float = 0.5,
                 foreground_threshold: float = 0.5, multimasking: bool = False, batch_size: int = 32,
                 nms_threshold: float = 0.9, intersection_over_min: bool = False, refine_with_box_prompts: bool = False,
                 precision: str = "fp32", return_records: bool = False):
    """AutomaticPromptGenerator.generate (instance_segmentation.py:1452-1505), output_mode 'instance_segmentation'."""
    from . import pipeline_ref as R
    prompts = derive_point_prompts(foreground, center_distances, boundary_distances, foreground_threshold,
                                   center_distance_threshold, boundary_distance_threshold)
    if prompts is None:
        return np.zeros(foreground.shape, dtype="uint32")
    kw = dict(batch_size=batch_size, return_instance_segmentation=False, multimasking=multimasking, precision=precision)
    predictions = R.batched_inference(sd, features, input_size, original_size, **kw, **prompts)
    if refine_with_box_prompts:
        predictions = R.batched_inference(sd, features, input_size, original_size, **kw, **derive_box_prompts(predictions, 0.01))
    if return_records:
        return predictions
    return A.apply_nms(predictions, min_size=min_size, nms_thresh=nms_threshold, intersection_over_min=intersection_over_min)


def decoder_maps_from_labels(labels: np.ndarray):
    """Ideal (foreground, centre distance, boundary distance) maps of a label image, the targets the reference trains its
    UNETR decoder on (``torch_em.transform.label.PerObjectDistanceTransform``: per object, distances normalised to [0, 1],
    centre distance 0 at the centre, boundary distance 0 on the boundary ... inverted so that both are LOW inside the
    seed region the way ``_derive_point_prompts`` reads them).  A test input generator, not part of the parity claim."""
    labels = np.asarray(labels)
    fg = (labels > 0).astype("float32")
    center = np.ones(labels.shape, dtype="float32")
    boundary = np.ones(labels.shape, dtype="float32")
    for seg_id in np.unique(labels):
        if seg_id == 0:
            continue
        m = labels == seg_id
        d = ndi.distance_transform_edt(m)
        boundary[m] = 1.0 - (d[m] / d.max())                  # 1 on the boundary -> 0 at the innermost pixel
        cy, cx = np.unravel_index(np.argmax(d), d.shape)
        yy, xx = np.nonzero(m)
        dc = np.sqrt((yy - cy) ** 2 + (xx - cx) ** 2)
        center[m] = dc / max(dc.max(), 1e-6)                  # 0 at the centre -> 1 at the rim
    return fg, center, boundary
