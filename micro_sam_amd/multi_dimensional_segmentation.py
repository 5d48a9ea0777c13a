"""Per-slice automatic segmentation of a volume / time series (reference
``micro_sam/multi_dimensional_segmentation.py:385-416`` ``_segment_slices``).

``segment_slices`` is the reference's serial loop: embeddings of all slices once (batched), then per slice
``initialize(i=z)`` + ``generate`` with a running id offset.  ``segment_slices_sharded`` runs the same loop on a
contiguous block of slices per rank (one process per GPU) and assembles the volume with the two collectives of
``parallel.gather_label_tiles``; its result equals the serial loop's on every rank.

The merge of the per-slice segmentations across z (``merge_instance_segmentation_3d``: overlap graph + multicut from the
un-vendored ``elf`` / ``nifty``) is a host step on top of this output and is not part of this build.

``segment_mask_in_volume`` (reference :105-233) is the interactive 3-d path: an object annotated in a few slices is
carried through the volume slice by slice, each step one ``prompt_based_segmentation.segment_from_mask`` call (prompts
derived from the neighbouring slice's mask, one decoder pass on the slice's precomputed embedding).
"""
from typing import Optional, Tuple

import numpy as np
import torch

from . import parallel, util


def segment_slices(data: np.ndarray, predictor, segmentor, embedding_path=None, verbose: bool = False,
                   tile_shape: Optional[Tuple[int, int]] = None, halo: Optional[Tuple[int, int]] = None,
                   batch_size: int = 1, **kwargs):
    """Reference ``_segment_slices`` (:385-416).  Returns (uint32 [Z,Y,X] segmentation, image_embeddings)."""
    assert data.ndim == 3
    image_embeddings = util.precompute_image_embeddings(predictor=predictor, input_=data, save_path=embedding_path, ndim=3,
                                                        tile_shape=tile_shape, halo=halo, verbose=verbose,
                                                        batch_size=batch_size, keep_on_device=tile_shape is None)
    offset = 0
    segmentation = np.zeros(data.shape, dtype="uint32")
    for i in range(segmentation.shape[0]):
        segmentor.initialize(data[i], image_embeddings=image_embeddings, verbose=False, i=i)
        seg = segmentor.generate(**kwargs)
        max_z = int(seg.max())
        if max_z == 0:
            continue
        seg[seg != 0] += offset
        offset = max_z + offset
        segmentation[i] = seg
    return segmentation, image_embeddings


def segment_slices_sharded(data: np.ndarray, predictor, segmentor, verbose: bool = False, batch_size: int = 1,
                           **kwargs) -> np.ndarray:
    """``segment_slices`` with the slices block-partitioned over the ranks of the default process group (weights
    replicated, no collective until the assembly).  Every rank returns the full uint32 [Z,Y,X] volume."""
    import torch.distributed as dist
    assert data.ndim == 3
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    start, stop = parallel.shard_range(data.shape[0], rank, world)
    local = np.zeros((stop - start,) + data.shape[1:], dtype="int32")
    if stop > start:
        emb = util.precompute_image_embeddings(predictor=predictor, input_=data[start:stop], ndim=3, verbose=verbose,
                                               batch_size=batch_size, keep_on_device=True)
        for k in range(stop - start):
            segmentor.initialize(data[start + k], image_embeddings=emb, verbose=False, i=k)
            local[k] = segmentor.generate(**kwargs).astype("int32")
    dev = predictor.device if world > 1 and dist.get_backend() == "nccl" else "cpu"
    out = parallel.gather_label_tiles(torch.as_tensor(local, device=dev), data.shape[0])
    return out.cpu().numpy().astype("uint32")


# ------------------------------------------------------------------------------------------ interactive 3-d projection

PROJECTION_MODES = ("box", "mask", "points", "points_and_mask", "single_point")
_PROJECTIONS = {                      # (use_box, use_mask, use_points, use_single_point)
    "mask": (True, True, False, False), "points": (False, False, True, False), "box": (True, False, False, False),
    "points_and_mask": (False, True, True, False), "single_point": (False, False, True, True),
}


def _validate_projection(projection):
    """Reference :48-72."""
    if isinstance(projection, str):
        if projection not in _PROJECTIONS:
            raise ValueError("Choose projection method from 'mask' / 'points' / 'box' / 'points_and_mask' / 'single_point'. "
                             f"You have passed the invalid option {projection}.")
        return _PROJECTIONS[projection]
    if isinstance(projection, dict):
        assert len(projection.keys()) == 3, "There should be three parameters assigned for the projection method."
        return projection["use_box"], projection["use_mask"], projection["use_points"], False
    raise ValueError(f"{projection} is not a supported projection method.")


def segment_mask_in_volume(segmentation: np.ndarray, predictor, image_embeddings, segmented_slices: np.ndarray,
                           stop_lower: bool, stop_upper: bool, iou_threshold: float, projection,
                           update_progress: Optional[callable] = None, box_extension: float = 0.0,
                           verbose: bool = False) -> Tuple[np.ndarray, Tuple[int, int]]:
    """Reference ``segment_mask_in_volume`` (:105-233).  ``segmentation`` [Z,H,W] holds the object (value 1) in the slices
    ``segmented_slices``; it is extended in place: downwards from the lowest and upwards from the highest annotated slice
    until the IoU between consecutive slices drops below ``iou_threshold`` (unless ``stop_lower`` / ``stop_upper``), and
    between annotated slices from both ends towards the middle (a single middle slice is prompted with the union of its
    two neighbours).  Returns the volume and the (lowest, highest) slice that now holds the object."""
    from .prompt_based_segmentation import segment_from_mask
    use_box, use_mask, use_points, use_single_point = _validate_projection(projection)
    if update_progress is None:
        def update_progress(*args):
            pass
    prompt_kw = dict(image_embeddings=image_embeddings, use_mask=use_mask, use_box=use_box, use_points=use_points,
                     box_extension=box_extension)

    def walk(z_start, z_stop, step, done, threshold=None):
        """Carry the mask from ``z_start`` towards ``z_stop``; returns the last slice written."""
        z = z_start + step
        while True:
            if verbose:
                print(f"Segment {z_start} to {z_stop}: segmenting slice {z}")
            previous = segmentation[z - step]
            seg_z, _, _ = segment_from_mask(predictor, previous, i=z, return_all=True, use_single_point=use_single_point,
                                            **prompt_kw)
            if threshold is not None:
                iou = util.compute_iou(previous, seg_z)
                if iou < threshold:
                    if verbose:
                        print(f"Segmentation stopped at slice {z} due to IOU {iou} < {threshold}.")
                    break
            segmentation[z] = seg_z
            z += step
            if done(z, z_stop):
                if verbose:
                    print(f"Segment {z_start} to {z_stop}: stop at slice {z}")
                break
            update_progress(1)
        return z - step

    def fill_between(z, below, above):
        """One slice prompted with the union of the object in two other slices."""
        union = np.logical_or(segmentation[below] == 1, segmentation[above] == 1)
        segmentation[z] = segment_from_mask(predictor, union, i=z, **prompt_kw)
        update_progress(1)

    z0, z1 = int(segmented_slices.min()), int(segmented_slices.max())
    last = segmentation.shape[0] - 1
    z_min = walk(z0, 0, -1, np.less, iou_threshold) if (z0 > 0 and not stop_lower) else z0
    z_max = walk(z1, last, 1, np.greater, iou_threshold) if (z1 < last and not stop_upper) else z1
    if z0 != z1:
        for z_start, z_stop in zip(segmented_slices[:-1], segmented_slices[1:]):
            gap = z_stop - z_start
            z_mid = int((z_start + z_stop) // 2)
            if gap == 1:
                continue
            if z_start == z0 and stop_lower:
                walk(z_stop, z_start, -1, np.less_equal)
            elif z_stop == z1 and stop_upper:
                walk(z_start, z_stop, 1, np.greater_equal)
            elif gap == 2:
                fill_between(z_start + 1, z_start, z_stop)
            else:
                walk(z_start, z_mid, 1, np.greater_equal if gap % 2 == 0 else np.greater)
                walk(z_stop, z_mid, -1, np.less_equal)
                if gap % 2 == 0:          # the middle slice is equally far from both ends: union of its neighbours
                    fill_between(z_mid, z_mid - 1, z_mid + 1)
    return segmentation, (z_min, z_max)
