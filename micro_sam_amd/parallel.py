"""Data-parallel sharding of independent tiles / slices over the GPUs of one node (one process per GPU,
``torch.distributed``: backend "nccl" = RCCL over xGMI on ROCm, "gloo" on CPU for tests).

The reference is single-device; its serial slice loop
(``micro_sam/multi_dimensional_segmentation.py:401-414``) gives slice ``z`` the id offset ``sum(max_id[:z])``.
Here every rank segments a contiguous block of items with local ids 1..K, then
  1. all_gather of the per-item max ids (int64 [n_items]) -> exclusive prefix sum = the serial loop's offsets,
  2. all_gather of the uint32 label tiles (4 MiB per 1024^2 tile) -> every rank holds the full stack.
No other collective is on the data path: tiles share nothing until this assembly step."""
from __future__ import annotations

from typing import List, Tuple

import torch
import torch.distributed as dist


def collectives_active() -> bool:
    """True when the collective code path has to run: a process group with more than one rank - or any initialised group when
    ``MSAM_FORCE_COLLECTIVES=1`` (a world-size-1 "nccl" group drives the same RCCL calls on one GPU: the smoke test of the N > 1 path
    that a single-GPU box can run, tests/test_gpu_rccl_smoke.py)."""
    import os
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get("MSAM_FORCE_COLLECTIVES") == "1"


def shard_range(n_items: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous block partition [start, stop) - keeps the z / tile order of the serial loop."""
    base, rem = divmod(n_items, world_size)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def gather_label_tiles(local_labels: torch.Tensor, n_items: int, relabel_globally: bool = True) -> torch.Tensor:
    """local_labels: int32/uint32-valued tensor [n_local, H, W] with ids 1..K per item (0 = background).

    Returns [n_items, H, W] (same dtype/device) on every rank; with ``relabel_globally`` ids of item i are shifted by the
    running offset sum(max_id[:i]) exactly like the reference's serial loop."""
    if not collectives_active():
        out = local_labels.clone()
        if relabel_globally:
            _apply_offsets(out, out.flatten(1).amax(dim=1).to(torch.int64))
        return out
    world, rank = dist.get_world_size(), dist.get_rank()
    counts = [shard_range(n_items, r, world)[1] - shard_range(n_items, r, world)[0] for r in range(world)]
    assert local_labels.shape[0] == counts[rank], (local_labels.shape, counts, rank)
    h, w = local_labels.shape[1:]
    n_max = max(counts)
    if min(counts) == n_max:
        # equal blocks (the usual case: tiles per rank fixed): gather straight into the output stack - no padding copy, no
        # list of per-rank buffers, no concatenation (each of those is a pass over world x n x 4 MiB)
        out = torch.empty((world * n_max, h, w), dtype=local_labels.dtype, device=local_labels.device)
        dist.all_gather_into_tensor(out, local_labels.contiguous())
        if relabel_globally:
            _apply_offsets(out, out.flatten(1).amax(dim=1).to(torch.int64))
        return out
    # all_gather needs equal shapes: pad the local block to n_max items
    padded = torch.zeros((n_max, h, w), dtype=local_labels.dtype, device=local_labels.device)
    padded[: counts[rank]] = local_labels
    gathered: List[torch.Tensor] = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(gathered, padded)
    out = torch.cat([g[:c] for g, c in zip(gathered, counts)], dim=0)
    if relabel_globally:
        # the max ids are a by-product of the gathered tiles; an explicit all_gather of them is only needed when the
        # tiles themselves stay sharded (gather-to-root variants)
        _apply_offsets(out, out.flatten(1).amax(dim=1).to(torch.int64))
    return out


def _apply_offsets(stack: torch.Tensor, max_ids: torch.Tensor) -> None:
    offsets = torch.cumsum(max_ids, 0) - max_ids
    off = offsets.view(-1, 1, 1).to(stack.dtype)
    stack += torch.where(stack != 0, off, torch.zeros_like(off))
