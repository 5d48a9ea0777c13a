"""Regular 2-D blocking with halos and the in-memory container of tiled embeddings.

``Blocking`` restates the subset of ``bioimage_cpp.utils.Blocking`` (un-vendored dependency of the reference,
``micro_sam/util.py:21``) that the tiled paths use (``util.py:765-803``, ``instance_segmentation.py:624-634``):
blocks of ``block_shape`` enumerated in C order over ``[roi_begin, roi_end)``, the last block per axis clipped;
``get_block_with_halo(block_id, halo)`` -> outer block = inner block grown by ``halo`` and clipped to the ROI.

``TiledFeatures`` / ``TileArray`` stand in for the zarr group the reference keeps tiled embeddings in (an in-memory
``zarr.group()`` when no ``save_path`` is given, ``util.py:1187-1189``): ``features[str(tile_id)]`` is an array-like with
``.attrs["original_size"]``, ``.attrs["input_size"]``, ``.ndim`` and indexing; the group carries ``.attrs["shape"]``,
``["tile_shape"]``, ``["halo"]`` and optionally ``["tiles_in_mask"]``.  The tensors stay in HBM.
"""
from typing import Dict, List, Sequence

import numpy as np
import torch


class Block:
    def __init__(self, begin: Sequence[int], end: Sequence[int]):
        self.begin, self.end = [int(b) for b in begin], [int(e) for e in end]

    @property
    def shape(self) -> List[int]:
        return [e - b for b, e in zip(self.begin, self.end)]

    def __repr__(self) -> str:
        return f"Block(begin={self.begin}, end={self.end})"


class BlockWithHalo:
    def __init__(self, inner: Block, outer: Block):
        self.inner_block, self.outer_block = inner, outer
        self.inner_block_local = Block([ib - ob for ib, ob in zip(inner.begin, outer.begin)],
                                       [ie - ob for ie, ob in zip(inner.end, outer.begin)])


class Blocking:
    def __init__(self, roi_begin: Sequence[int], roi_end: Sequence[int], block_shape: Sequence[int]):
        self.roi_begin = [int(x) for x in roi_begin]
        self.roi_end = [int(x) for x in roi_end]
        self.block_shape = [int(x) for x in block_shape]
        if len(self.roi_begin) != len(self.roi_end) or len(self.roi_begin) != len(self.block_shape):
            raise ValueError("roi_begin, roi_end and block_shape must have the same length")
        if any(bs <= 0 for bs in self.block_shape):
            raise ValueError("block_shape must be positive")
        self.blocks_per_axis = [max((e - b + bs - 1) // bs, 0) for b, e, bs in zip(self.roi_begin, self.roi_end, self.block_shape)]
        self.number_of_blocks = int(np.prod(self.blocks_per_axis))

    def _coords(self, block_id: int) -> List[int]:
        if not 0 <= block_id < self.number_of_blocks:
            raise IndexError(f"block id {block_id} out of range [0, {self.number_of_blocks})")
        coords = []
        for n in reversed(self.blocks_per_axis):
            coords.append(block_id % n)
            block_id //= n
        return coords[::-1]

    def coordinates_to_block_id(self, coordinates: Sequence[int]) -> int:
        """Id of the block that contains the point (bioimage_cpp ``Blocking.coordinates_to_block_id``; reference call sites
        micro_sam/inference.py:463,480); coordinates outside the roi are clamped to the border blocks."""
        block_id = 0
        for c, rb, bs, n in zip(coordinates, self.roi_begin, self.block_shape, self.blocks_per_axis):
            block_id = block_id * n + min(max((int(c) - rb) // bs, 0), n - 1)
        return int(block_id)

    def get_block(self, block_id: int) -> Block:
        c = self._coords(int(block_id))
        begin = [rb + ci * bs for rb, ci, bs in zip(self.roi_begin, c, self.block_shape)]
        end = [min(b + bs, re) for b, bs, re in zip(begin, self.block_shape, self.roi_end)]
        return Block(begin, end)

    def get_block_with_halo(self, block_id: int, halo: Sequence[int]) -> BlockWithHalo:
        inner = self.get_block(block_id)
        outer = Block([max(b - h, rb) for b, h, rb in zip(inner.begin, halo, self.roi_begin)],
                      [min(e + h, re) for e, h, re in zip(inner.end, halo, self.roi_end)])
        return BlockWithHalo(inner, outer)


class TileArray:
    """Embedding of one tile: tensor [1,256,64,64] (2-D input) or [Z,1,256,64,64] (3-D input) + zarr-like attrs."""

    def __init__(self, data: torch.Tensor, original_size, input_size):
        self.data = data
        self.attrs = {"original_size": tuple(int(x) for x in original_size), "input_size": tuple(int(x) for x in input_size)}

    @property
    def ndim(self) -> int:
        return self.data.ndim

    @property
    def shape(self):
        return tuple(self.data.shape)

    def __getitem__(self, idx):
        return self.data[idx]


class TiledFeatures:
    def __init__(self, shape, tile_shape, halo):
        self.attrs: Dict[str, object] = {"shape": tuple(int(x) for x in shape), "tile_shape": tuple(int(x) for x in tile_shape),
                                         "halo": tuple(int(x) for x in halo)}
        self._tiles: Dict[str, TileArray] = {}

    def __contains__(self, name) -> bool:
        return str(name) in self._tiles

    def __getitem__(self, name) -> TileArray:
        return self._tiles[str(name)]

    def __setitem__(self, name, value: TileArray) -> None:
        self._tiles[str(name)] = value

    def __len__(self) -> int:
        return len(self._tiles)

    def keys(self):
        return self._tiles.keys()
