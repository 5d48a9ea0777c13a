"""InstanceSegmentationWithDecoder.generate (reference micro_sam/instance_segmentation.py:1083-1168: watershed_from_center_and_boundary_distances),
host half: the library's priority flood (csrc/watershed.hip = scikit-image's published algorithm) and the seed / mask logic around it.
scikit-image / vigra / torch_em are absent, so the checks are the flood's defining properties plus hand-computed small cases."""
import numpy as np
import pytest

from micro_sam_amd import instance_segmentation as IS


def test_flood_fills_the_mask_from_the_seeds_in_height_order():
    # a 1-d valley with two seeds: the ridge at x = 5 (height 9) is claimed by whoever reaches it first in (height, age) order
    h = np.array([[3, 2, 1, 2, 4, 9, 4, 2, 1, 2, 3]], dtype=np.float32)
    m = np.zeros_like(h, dtype=np.int32); m[0, 2] = 1; m[0, 8] = 2
    out = IS.seeded_watershed(h, m)
    assert out.tolist() == [[1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 2]]          # both neighbours of the ridge have height 4; label 1's push is older
    # mask: pixels outside stay 0 and block the flood
    mask = np.ones_like(h, dtype=bool); mask[0, 4] = False
    out = IS.seeded_watershed(h, m, mask)
    assert out.tolist() == [[1, 1, 1, 1, 0, 2, 2, 2, 2, 2, 2]]
    # a seed outside the mask is dropped
    mask = np.ones_like(h, dtype=bool); mask[0, 2] = False
    assert set(np.unique(IS.seeded_watershed(h, m, mask)).tolist()) == {0, 2}


def test_flood_properties_on_random_maps():
    rng = np.random.default_rng(0)
    from scipy import ndimage
    h = ndimage.gaussian_filter(rng.random((96, 128)).astype(np.float32), 3)
    fg = ndimage.gaussian_filter(rng.random((96, 128)), 6) > 0.49
    seeds = np.zeros((96, 128), dtype=np.int32)
    pts = rng.integers(0, [96, 128], size=(40, 2))
    for k, (y, x) in enumerate(pts, start=1):
        seeds[y, x] = k
    out = IS.seeded_watershed(h, seeds, fg)
    assert (out[~fg] == 0).all()
    inside = seeds * fg
    assert (out[inside > 0] == inside[inside > 0]).all()                  # seeds keep their label
    # every 4-connected foreground component that holds a seed is completely labelled, one without a seed stays 0
    comp, n = ndimage.label(fg)
    for c in range(1, n + 1):
        has_seed = (inside[comp == c] > 0).any()
        assert (out[comp == c] != 0).all() == has_seed and (has_seed or (out[comp == c] == 0).all())
    # every label region is 4-connected (a flood never jumps)
    for lab in np.unique(out)[1:]:
        assert ndimage.label(out == lab)[1] == 1


def test_generate_from_decoder_maps():
    """Three discs: foreground = disc, centre distance small at the centre, boundary distance small inside (as the UNETR heads are
    trained): one instance per disc, regenerate ==, state round trip ==, min_size filter, binary_mask records."""
    yy, xx = np.mgrid[0:200, 0:260]
    fg = np.zeros((200, 260), np.float32); cd = np.ones_like(fg); bd = np.ones_like(fg)
    for (cy, cx, r) in ((50, 60, 30), (140, 90, 35), (90, 200, 40)):
        d = np.sqrt((yy - cy) ** 2 + (xx - cx) ** 2)
        fg[d < r] = 1.0
        cd = np.minimum(cd, np.clip(d / r, 0, 1))
        bd = np.where(d < r, np.minimum(bd, np.clip(1.2 - (r - d) / 8.0, 0, 1)), bd)
    bd[fg == 0] = 1.0
    fg[8:15, 8:15] = 1.0; cd[8:15, 8:15] = 0.0; bd[8:15, 8:15] = 0.0                    # a 49-pixel speck with its own seed
    seg = IS.InstanceSegmentationWithDecoder(None, None)
    with pytest.raises(RuntimeError):
        seg.generate()
    seg.set_state({"foreground": fg, "center_distances": cd, "boundary_distances": bd})
    out = seg.generate(min_size=0)
    assert out.shape == fg.shape and out.dtype == np.uint32 and out.max() == 4
    assert len({int(out[50, 60]), int(out[140, 90]), int(out[90, 200]), int(out[11, 11])}) == 4
    assert np.array_equal(out, seg.generate(min_size=0))
    other = IS.InstanceSegmentationWithDecoder(None, None)
    other.set_state(seg.get_state())
    assert np.array_equal(out, other.generate(min_size=0))
    big = seg.generate(min_size=50)
    assert big.max() == 3 and big[11, 11] == 0
    recs = seg.generate(min_size=50, output_mode="binary_mask")
    assert len(recs) == 3 and all(r["segmentation"].sum() == r["area"] for r in recs)
