"""tiling.Blocking restates the blocking of the reference's un-vendored ``bioimage_cpp.utils.Blocking`` as used by the
tiled paths (micro_sam/util.py:765-803, instance_segmentation.py:624-634): C-order block ids, clipped last blocks, outer
blocks = inner + halo clipped to the image.  Checked against the brute-force definition."""
import numpy as np
import pytest

from micro_sam_amd.tiling import Blocking, TileArray, TiledFeatures


@pytest.mark.parametrize("shape,block,halo", [((600, 720), (384, 384), (64, 64)), ((1024, 1024), (512, 512), (0, 0)),
                                              ((1000, 300), (256, 512), (32, 100)), ((5, 7), (2, 3), (1, 1))])
def test_blocking_covers_image_once(shape, block, halo):
    b = Blocking([0, 0], shape, block)
    ny, nx = -(-shape[0] // block[0]), -(-shape[1] // block[1])
    assert b.number_of_blocks == ny * nx
    cover = np.zeros(shape, dtype=int)
    for bid in range(b.number_of_blocks):
        t = b.get_block_with_halo(bid, list(halo))
        iy, ix = bid // nx, bid % nx                                      # C order
        assert t.inner_block.begin == [iy * block[0], ix * block[1]]
        assert t.inner_block.end == [min((iy + 1) * block[0], shape[0]), min((ix + 1) * block[1], shape[1])]
        assert t.outer_block.begin == [max(t.inner_block.begin[0] - halo[0], 0), max(t.inner_block.begin[1] - halo[1], 0)]
        assert t.outer_block.end == [min(t.inner_block.end[0] + halo[0], shape[0]), min(t.inner_block.end[1] + halo[1], shape[1])]
        loc = t.inner_block_local
        assert [loc.begin[k] + t.outer_block.begin[k] for k in range(2)] == t.inner_block.begin
        cover[t.inner_block.begin[0]:t.inner_block.end[0], t.inner_block.begin[1]:t.inner_block.end[1]] += 1
    assert (cover == 1).all()
    with pytest.raises(IndexError):
        b.get_block(b.number_of_blocks)


def test_tiled_features_container():
    import torch
    f = TiledFeatures((600, 720), (384, 384), (64, 64))
    f[2] = TileArray(torch.zeros(1, 256, 64, 64), (280, 448), (640, 1024))
    assert "2" in f and 2 in f and len(f) == 1 and f["2"].ndim == 4 and f[2].attrs["input_size"] == (640, 1024)
    assert f.attrs["tile_shape"] == (384, 384) and f.attrs.get("tiles_in_mask", None) is None


def test_compute_iou_known_answers():
    """The reference's known-answer test test/test_util.py:66-79."""
    from micro_sam_amd.util import compute_iou
    x1, x2 = np.zeros((32, 32), dtype="uint32"), np.zeros((32, 32), dtype="uint32")
    x1[:16] = 1
    x2[16:] = 1
    assert np.isclose(compute_iou(x1, x1), 1.0) and np.isclose(compute_iou(x1, x2), 0.0)
    rng = np.random.default_rng(0)
    for _ in range(10):
        a, b = rng.random((32, 32)) > 0.5, rng.random((32, 32)) > 0.5
        assert 0.0 < compute_iou(a, b) < 1.0
