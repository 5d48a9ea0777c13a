"""Host half of micro_sam_amd.prompt_based_segmentation (reference micro_sam/prompt_based_segmentation.py:30-506): prompt
conversions, tile selection and what reaches ``SamPredictor.predict``.  Known answers derived by hand from the reference's
formulas; the device half is in tests/test_gpu_prompt_based_segmentation.py."""
import warnings

import numpy as np
import pytest

from micro_sam_amd import prompt_based_segmentation as PB
from micro_sam_amd.tiling import TiledFeatures


def _disk(shape, center, radius):
    yy, xx = np.mgrid[0:shape[0], 0:shape[1]]
    return ((yy - center[0]) ** 2 + (xx - center[1]) ** 2 < radius * radius).astype("uint8")


def test_process_box_known_answers():
    box = np.array([10, 20, 30, 60])                                         # y0, x0, y1, x1
    assert PB._process_box(box, (100, 100)).tolist() == [20, 10, 60, 30]     # -> XYXY
    assert PB._process_box(box, (100, 100), box_extension=0.1).tolist() == [16, 8, 64, 32]      # 10 % of 40 / 20
    assert PB._process_box(box, (100, 100), box_extension=5).tolist() == [15, 5, 65, 35]        # 5 px
    assert PB._process_box(box, (100, 62), box_extension=25).tolist() == [0, 0, 62, 55]         # clipped to the shape
    # a box of a 256^2 mask frame rescaled to the image: longest side 1024 -> x4
    assert PB._process_box(box, (256, 256), original_size=(512, 1024)).tolist() == [80, 40, 240, 120]
    out = PB._process_box(np.array([10.4, 20.6, 30.5, 60.5]), (100, 100))
    assert out.dtype.kind == "i" and out.tolist() == [21, 10, 60, 30]                            # np.round: half to even


def test_box_and_logits_from_mask():
    m = _disk((256, 256), (128, 100), 20)
    assert PB._compute_box_from_mask(m).tolist() == [81, 109, 120, 148]                         # half-open, XYXY
    assert PB._compute_box_from_mask(m, box_extension=4).tolist() == [77, 105, 124, 152]
    lg = PB._compute_logits_from_mask(m)
    hi = np.float32(np.log(0.999 / 0.001))
    assert lg.shape == (1, 256, 256) and lg.dtype == np.float32
    assert np.array_equal(lg[0] > 0, m == 1) and set(np.unique(lg).tolist()) == {float(-hi), float(hi)}
    # only label 1 is the object
    m2 = m.copy(); m2[:10, :10] = 2
    assert np.array_equal(PB._compute_logits_from_mask(m2), lg)
    # non-square: longest side -> 256, the short side is zero padded (and stays "outside")
    big = np.zeros((512, 1024), "uint8"); big[128:384, 256:768] = 1
    lg = PB._compute_logits_from_mask(big)
    assert lg.shape == (1, 256, 256) and (lg[0, 128:] < 0).all()
    inside = lg[0] > 0
    ys, xs = np.nonzero(inside)
    assert (ys.min(), ys.max() + 1, xs.min(), xs.max() + 1) == (32, 96, 64, 192)


def test_peak_local_max_known_answer():
    img = np.zeros((20, 20), "float32")
    img[5, 5], img[5, 7], img[12, 12], img[19, 0] = 3.0, 2.0, 1.0, 0.5
    assert PB._peak_local_max(img, 3).tolist() == [[5, 5], [12, 12], [19, 0]]       # (5,7) is within 3 of a stronger peak
    assert PB._peak_local_max(img, 1).tolist() == [[5, 5], [5, 7], [12, 12], [19, 0]]
    assert PB._peak_local_max(np.ones((6, 6), "float32"), 2).shape == (0, 2)         # constant image: no peak


def test_points_from_mask():
    m = _disk((200, 240), (90, 130), 30)
    pts, lbl = PB._compute_points_from_mask(m, None, box_extension=0, use_single_point=True)
    assert pts.tolist() == [[130, 90]] and lbl.tolist() == [1]                       # XY of the disk centre
    pts, lbl = PB._compute_points_from_mask(m, None, box_extension=10)
    assert lbl[0] == 1 and (lbl == 1).sum() >= 1 and (lbl == 0).sum() >= 1
    for (x, y), l in zip(pts, lbl):
        assert bool(m[int(y), int(x)]) == bool(l)                                    # positives inside, negatives outside
    # a downsampled mask: the points are scaled to the full frame
    pts2, _ = PB._compute_points_from_mask(m, (400, 480), box_extension=10)
    assert np.allclose(pts2, pts * 2)


def test_prompts_to_tile():
    shape, tile_shape, halo = (1024, 1024), (512, 512), (96, 96)
    pts, lbl = np.array([[510, 510], [400, 200], [200, 400]]), np.array([1, 0, 0])
    tile_id, tile, (p, l) = PB._points_to_tile((pts, lbl), shape, tile_shape, halo)
    assert tile_id == 0 and tile.begin == [0, 0] and tile.end == [608, 608] and np.array_equal(p, pts) and np.array_equal(l, lbl)
    pts = np.array([[300, 900], [200, 1000], [100, 100]])                            # mean (200, 667) -> block (0, 1)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        tile_id, tile, (p, l) = PB._points_to_tile((pts, np.array([1, 1, 0])), shape, tile_shape, halo)
    assert tile_id == 1 and tile.begin == [0, 416] and tile.end == [608, 1024]
    assert len(w) == 1 and "1 points were not in the tile" in str(w[0].message)      # (100, 100) lies left of the tile
    assert p.tolist() == [[300, 484], [200, 584]] and l.tolist() == [1, 1]
    tile_id, tile, b = PB._box_to_tile(np.array([500, 600, 700, 900]), shape, tile_shape, halo)     # centre (600, 750) -> block 3
    assert tile_id == 3 and tile.begin == [416, 416] and b.tolist() == [84, 184, 284, 484]
    tile_id, tile, b = PB._box_to_tile(np.array([300, 10, 700, 200]), shape, tile_shape, halo)      # centre (500, 105) -> block 0
    assert tile_id == 0 and b.tolist() == [300, 10, 608, 200]                                          # clipped to the outer tile
    m = np.zeros(shape, bool); m[600:700, 100:200] = True
    tile_id, tile, mt = PB._mask_to_tile(m, shape, tile_shape, halo)
    assert tile_id == 2 and tile.begin == [416, 0] and mt.shape == (608, 608) and mt.sum() == m.sum()
    full = PB._tile_to_full_mask(np.ones((3, 608, 608), bool), shape, tile)
    assert full.shape == (3, 1024, 1024) and full[:, 416:, :608].all() and full.sum() == 3 * 608 * 608


class _Recorder:
    """Stands in for SamPredictor: records what ``predict`` receives, answers with fixed arrays of the right shapes."""

    def __init__(self, original_size=(256, 256)):
        self.original_size, self.calls, self.set = original_size, [], []

    def predict(self, **kw):
        self.calls.append(kw)
        c = 3 if kw.get("multimask_output") else 1
        h, w = self.original_size
        masks = np.zeros((c, h, w), bool)
        for k in range(c):
            masks[k, :k + 1] = True
        return masks, np.array([0.2, 0.9, 0.5][:c]), np.zeros((c, 256, 256), "float32")


def test_segment_from_points_prompt_conversion():
    r = _Recorder()
    pts, lbl = np.array([[128, 100], [64, 32]]), np.array([1, 0])
    out = PB.segment_from_points(r, pts, lbl)
    kw = r.calls[-1]
    assert kw["point_coords"].tolist() == [[100, 128], [32, 64]] and kw["multimask_output"] is False      # (y,x) -> XY
    assert out.shape == (1, 256, 256)
    # one positive point: three masks are decoded, the one with the highest score (index 1) is returned
    out, scores, logits = PB.segment_from_points(r, pts[:1], lbl[:1], return_all=True)
    assert r.calls[-1]["multimask_output"] is True and out.shape == (1, 256, 256) and out[0, :2].all() and not out[0, 2:].any()
    assert scores.shape == (3,) and logits.shape == (3, 256, 256)
    assert PB.segment_from_points(r, pts[:1], lbl[:1], use_best_multimask=False).shape == (1, 256, 256)
    assert r.calls[-1]["multimask_output"] is False
    assert PB.segment_from_points(r, pts[:1], lbl[:1], multimask_output=True, use_best_multimask=False).shape == (3, 256, 256)
    # a single NEGATIVE point does not trigger the best-of-three
    PB.segment_from_points(r, pts[1:], lbl[1:])
    assert r.calls[-1]["multimask_output"] is False


def test_segment_from_mask_box_prompt_conversion():
    r = _Recorder()
    m = _disk((256, 256), (128, 100), 20)
    PB.segment_from_mask(r, m)                                                         # default: box + mask
    kw = r.calls[-1]
    assert kw["box"].tolist() == [81, 109, 120, 148] and kw["mask_input"].shape == (1, 256, 256)
    assert kw["point_coords"] is None and kw["point_labels"] is None and kw["return_logits"] is False
    PB.segment_from_mask(r, m, use_box=False, use_mask=False, use_points=True, use_single_point=True)
    kw = r.calls[-1]
    assert kw["box"] is None and kw["mask_input"] is None and kw["point_coords"].tolist() == [[100, 128]]
    PB.segment_from_mask(r, m, box=np.array([100, 70, 160, 130]), box_extension=0.5, use_mask=False)
    assert r.calls[-1]["box"].tolist() == [40, 70, 160, 190]
    with pytest.raises(ValueError):
        PB.segment_from_mask(r, m, points=np.array([[1, 2]]))
    # an empty mask gives no box and no points
    PB.segment_from_mask(r, np.zeros((256, 256), "uint8"), use_points=True)
    kw = r.calls[-1]
    assert kw["box"] is None and kw["point_coords"] is None and (kw["mask_input"] < 0).all()
    PB.segment_from_box(r, np.array([10, 20, 30, 60]), box_extension=0.1)
    assert r.calls[-1]["box"].tolist() == [16, 8, 64, 32]
    PB.segment_from_box_and_points(r, np.array([10, 20, 30, 60]), np.array([[20, 40]]), np.array([1]), multimask_output=True)
    kw = r.calls[-1]
    assert kw["box"].tolist() == [20, 10, 60, 30] and kw["point_coords"].tolist() == [[40, 20]] and kw["multimask_output"] is True


def test_tiled_embeddings_select_the_tile(monkeypatch):
    """With tiled embeddings the prompts are moved into the tile that holds them and the prediction is placed back."""
    feats = TiledFeatures((1024, 1024), (512, 512), (96, 96))
    emb = {"features": feats, "input_size": None, "original_size": None}
    picked = []

    def fake_set_precomputed(predictor, image_embeddings, i=None, tile_id=None):
        picked.append(tile_id)
        predictor.original_size = (608, 608)
        return predictor
    monkeypatch.setattr(PB.util, "set_precomputed", fake_set_precomputed)
    r = _Recorder()
    out = PB.segment_from_points(r, np.array([[900, 200], [800, 100]]), np.array([1, 0]), image_embeddings=emb)
    assert picked == [2] and r.calls[-1]["point_coords"].tolist() == [[200, 900 - 416], [100, 800 - 416]]
    assert out.shape == (1, 1024, 1024) and out[0, 416, :608].all() and not out[0, :416].any() and not out[0, :, 608:].any()
    out = PB.segment_from_box(r, np.array([500, 600, 700, 900]), image_embeddings=emb)
    assert picked[-1] == 3 and r.calls[-1]["box"].tolist() == [184, 84, 484, 284] and out.shape == (1, 1024, 1024)
    m = np.zeros((1024, 1024), "uint8"); m[100:200, 700:800] = 1
    out = PB.segment_from_mask(r, m, image_embeddings=emb)
    assert picked[-1] == 1 and r.calls[-1]["box"].tolist() == [700 - 416, 100, 800 - 416, 200]
    with pytest.raises(RuntimeError):                                                  # box and points in different tiles
        PB.segment_from_box_and_points(r, np.array([500, 600, 700, 900]), np.array([[100, 100]]), np.array([1]),
                                       image_embeddings=emb)


def _shrinking_segmenter(calls):
    """Stands in for segment_from_mask: answers with the prompt mask eroded by one pixel per side."""
    from scipy.ndimage import binary_erosion

    def fake(predictor, mask, image_embeddings=None, i=None, return_all=False, **kw):
        calls.append(dict(i=i, area=int((mask == 1).sum()), **kw))
        out = binary_erosion(mask == 1, structure=np.ones((3, 3), bool))[None]
        return (out, np.array([0.9]), None) if return_all else out
    return fake


def test_segment_mask_in_volume_control_flow(monkeypatch):
    """multi_dimensional_segmentation.segment_mask_in_volume (reference :105-233): which slices are visited, with which
    prompt, where the IoU criterion stops - with the per-slice segmentation replaced by a deterministic erosion."""
    from micro_sam_amd import multi_dimensional_segmentation as M
    calls = []
    monkeypatch.setattr(PB, "segment_from_mask", _shrinking_segmenter(calls))
    vol = np.zeros((12, 40, 40), "uint8")
    vol[4, 10:30, 10:30] = 1
    vol[9, 10:30, 10:30] = 1
    # consecutive IoUs going outwards: (18/20)^2 = .81, (16/18)^2 = .79, (14/16)^2 = .766, (12/14)^2 = .735 < 0.75
    seg, (z_min, z_max) = M.segment_mask_in_volume(vol, None, None, np.array([4, 9]), stop_lower=False, stop_upper=False,
                                                   iou_threshold=0.75, projection="mask", box_extension=0.05)
    assert seg is vol and (z_min, z_max) == (1, 11)
    areas = [int(vol[z].sum()) for z in range(12)]
    assert areas == [0, 14 * 14, 16 * 16, 18 * 18, 400, 18 * 18, 16 * 16, 16 * 16, 18 * 18, 400, 18 * 18, 16 * 16]
    assert [c["i"] for c in calls] == [3, 2, 1, 0, 10, 11, 5, 6, 8, 7]               # down, up, bottom half, top half
    assert all(c["use_box"] and c["use_mask"] and not c["use_points"] and c["box_extension"] == 0.05 for c in calls)
    # even gap: the middle slice is prompted with the union of its two neighbours; stop flags keep the ends fixed
    calls.clear()
    vol = np.zeros((9, 40, 40), "uint8")
    vol[2, 10:30, 10:30] = 1
    vol[6, 14:34, 10:30] = 1
    seg, rng = M.segment_mask_in_volume(vol, None, None, np.array([2, 6]), stop_lower=True, stop_upper=True,
                                        iou_threshold=0.75, projection="single_point")
    assert rng == (2, 6) and not vol[:2].any() and not vol[7:].any()
    assert [c["i"] for c in calls] == [5, 4, 3]                                     # lower end is a stop: walk down from the top
    assert all(c["use_points"] and not c["use_box"] and not c["use_mask"] for c in calls)
    assert calls[0]["use_single_point"] is True
    calls.clear()
    vol = np.zeros((9, 40, 40), "uint8")
    vol[2, 10:30, 10:30] = 1
    vol[6, 14:34, 10:30] = 1
    M.segment_mask_in_volume(vol, None, None, np.array([2, 6]), stop_lower=False, stop_upper=False, iou_threshold=0.99,
                             projection={"use_box": True, "use_mask": False, "use_points": True})
    # threshold 0.99: nothing is added below 2 / above 6 (the first step already fails); between: 3 from below, 5 from above,
    # then 4 from the union of 3 and 5 (rows 11..32 x cols 11..28 = 22 x 18, eroded to 20 x 16)
    assert [c["i"] for c in calls] == [1, 7, 3, 5, 4] and calls[-1]["area"] == 22 * 18
    assert not vol[1].any() and not vol[7].any() and int(vol[4].sum()) == 20 * 16
    # adjacent annotated slices: nothing to do in between
    calls.clear()
    vol = np.zeros((4, 20, 20), "uint8"); vol[1:3, 5:15, 5:15] = 1
    M.segment_mask_in_volume(vol, None, None, np.array([1, 2]), True, True, 0.5, "box")
    assert calls == []
    with pytest.raises(ValueError):
        M.segment_mask_in_volume(vol, None, None, np.array([1]), True, True, 0.5, "blob")
    assert M._validate_projection("points_and_mask") == (False, True, True, False)
