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


def test_single_process_path():
    x = torch.tensor([[[0, 1], [2, 0]], [[1, 1], [0, 3]]], dtype=torch.int32)
    out = parallel.gather_label_tiles(x, 2)
    assert out.tolist() == [[[0, 1], [2, 0]], [[3, 3], [0, 5]]]
