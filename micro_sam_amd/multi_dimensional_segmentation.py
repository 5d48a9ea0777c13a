"""Per-slice automatic segmentation of a volume / time series (reference
``micro_sam/multi_dimensional_segmentation.py:385-416`` ``_segment_slices``).

``segment_slices`` is the reference's serial loop: embeddings of all slices once (batched), then per slice
``initialize(i=z)`` + ``generate`` with a running id offset.  ``segment_slices_sharded`` runs the same loop on a
contiguous block of slices per rank (one process per GPU) and assembles the volume with the two collectives of
``parallel.gather_label_tiles``; its result equals the serial loop's on every rank.

The merge of the per-slice segmentations across z (``merge_instance_segmentation_3d``: overlap graph + multicut from the
un-vendored ``elf`` / ``nifty``) is a host step on top of this output and is not part of this build.
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
