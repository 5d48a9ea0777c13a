"""N > 1 path on CPU: world_size-2 gloo run of the tile sharding + label-tile gather (no GPU needed)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from micro_sam_amd import parallel


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n_items, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    start, stop = parallel.shard_range(n_items, rank, world)
    rng = np.random.default_rng(0)
    full = [rng.integers(0, 5 + i, size=(8, 8)).astype(np.int32) for i in range(n_items)]   # same on every rank
    local = torch.as_tensor(np.stack(full[start:stop])) if stop > start else torch.zeros((0, 8, 8), dtype=torch.int32)
    out = parallel.gather_label_tiles(local, n_items)
    q.put((rank, out.numpy()))
    dist.destroy_process_group()


def test_shard_range_is_a_contiguous_partition():
    for n in (1, 5, 8, 13):
        for world in (1, 2, 3, 8):
            ranges = [parallel.shard_range(n, r, world) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))


@pytest.mark.parametrize("n_items", [5, 4])          # unequal blocks (padded list all_gather) / equal blocks (all_gather_into_tensor)
def test_gather_label_tiles_matches_serial_offsets(n_items):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_items, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    # serial reference: micro_sam/multi_dimensional_segmentation.py:401-414
    rng = np.random.default_rng(0)
    full = [rng.integers(0, 5 + i, size=(8, 8)).astype(np.int32) for i in range(n_items)]
    offset, expect = 0, []
    for seg in full:
        seg = seg.copy()
        m = seg.max()
        seg[seg != 0] += offset
        offset += m
        expect.append(seg)
    expect = np.stack(expect)
    for r in range(world):
        assert np.array_equal(results[r], expect), r


@pytest.mark.parametrize("n_items", [256, 61])       # BASELINE configs[1]'s 256 tiles (32 per rank) / a 61-slice volume (5 ranks x 8 + 3 x 7)
def test_world_8_partitions(n_items):
    """The 8-rank partitions the driver's scaling run uses, on gloo: equal blocks and the uneven-remainder path."""
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_items, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
    rng = np.random.default_rng(0)
    full = [rng.integers(0, 5 + i, size=(8, 8)).astype(np.int32) for i in range(n_items)]
    offset, expect = 0, []
    for seg in full:
        seg = seg.copy()
        m = seg.max()
        seg[seg != 0] += offset
        offset += m
        expect.append(seg)
    expect = np.stack(expect)
    counts = [parallel.shard_range(n_items, r, world) for r in range(world)]
    assert sorted(b - a for a, b in counts)[0] >= n_items // world and sum(b - a for a, b in counts) == n_items
    for r in range(world):
        assert np.array_equal(results[r], expect), r


def test_single_process_path():
    x = torch.tensor([[[0, 1], [2, 0]], [[1, 1], [0, 3]]], dtype=torch.int32)
    out = parallel.gather_label_tiles(x, 2)
    assert out.tolist() == [[[0, 1], [2, 0]], [[3, 3], [0, 5]]]


# ---- segment_slices_sharded (reference _segment_slices, micro_sam/multi_dimensional_segmentation.py:385-416) under gloo
class _StubSegmentor:
    """Deterministic per-slice 'segmentation' (no GPU): labels 1..K on the slice's bright columns."""

    def initialize(self, image, image_embeddings=None, verbose=False, i=None):
        self._img = image

    def generate(self, **kwargs):
        seg = np.zeros(self._img.shape, dtype="uint32")
        k = int(self._img[0, 0]) % 4                       # K differs per slice, 0 for some
        for j in range(k):
            seg[:, 2 * j] = j + 1
        return seg


def _sharded_worker(rank, world, port, vol, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from micro_sam_amd import multi_dimensional_segmentation as M
    from micro_sam_amd import util

    class _P:
        device = "cpu"
    util.precompute_image_embeddings = lambda **kw: {"features": None, "input_size": None, "original_size": None}
    out = M.segment_slices_sharded(vol, _P(), _StubSegmentor())
    q.put((rank, out))
    dist.destroy_process_group()


def test_segment_slices_sharded_matches_serial_loop():
    from micro_sam_amd import multi_dimensional_segmentation as M
    from micro_sam_amd import util
    rng = np.random.default_rng(3)
    vol = rng.integers(0, 255, size=(5, 8, 8)).astype(np.uint8)
    vol[:, 0, 0] = [3, 0, 2, 1, 3]                          # K per slice: 3, 0, 2, 1, 3

    class _P:
        device = "cpu"
    orig = util.precompute_image_embeddings
    util.precompute_image_embeddings = lambda **kw: {"features": None, "input_size": None, "original_size": None}
    try:
        serial, _ = M.segment_slices(vol, _P(), _StubSegmentor())
    finally:
        util.precompute_image_embeddings = orig
    assert serial.max() == 9 and serial[1].max() == 0 and serial[2].max() == 5
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sharded_worker, args=(r, world, port, vol, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    for r in range(world):
        assert results[r].dtype == np.uint32 and np.array_equal(results[r], serial), r
