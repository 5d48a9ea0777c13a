"""Host half of the reference's AutomaticPromptGenerator (micro_sam/instance_segmentation.py:1322-1628, SURVEY.md 8(f)
rank 1): prompt derivation from decoder maps, state handling, the factory, tile-local ``apply_nms``.  CPU only - the
device half (``batched_inference`` on the derived prompts) is in tests/test_gpu_prompt_generator.py."""
import numpy as np
import pytest

from micro_sam_amd import instance_segmentation as IS
from micro_sam_amd import util
from micro_sam_amd._label_image_ops import blockwise_distance_transform, find_outer_boundaries, label_regions
from micro_sam_amd.synthetic import three_disk_fixture
from oracle import amg_ref as A
from oracle import apg_ref as G


def _random_labels(g, shape, n, max_side):
    lab = np.zeros(shape, np.uint32)
    for k in range(1, n + 1):
        y, x = g.integers(0, shape[0] - 1), g.integers(0, shape[1] - 1)
        lab[y:y + g.integers(1, max_side), x:x + g.integers(1, max_side)] = k
    return lab


def test_outer_boundaries_known_answers():
    # one pixel: its four edge neighbours (not the diagonal ones, not the pixel)
    lab = np.zeros((5, 5), np.uint32)
    lab[2, 2] = 1
    want = np.zeros((5, 5), bool)
    want[1, 2] = want[3, 2] = want[2, 1] = want[2, 3] = True
    assert np.array_equal(find_outer_boundaries(lab), want)
    # two objects that touch: the touching columns of BOTH objects are boundary, the far columns are not
    lab = np.zeros((5, 8), np.uint32)
    lab[1:4, 1:4] = 1
    lab[1:4, 4:7] = 2
    b = find_outer_boundaries(lab)
    assert b[1:4, 3].all() and b[1:4, 4].all() and not b[1:4, 1:3].any() and not b[1:4, 5:7].any()
    assert b[0, 1:7].all() and b[4, 1:7].all() and b[1:4, 0].all() and b[1:4, 7].all()
    assert not b[0, 0] and not b[4, 7]                                        # corners touch through a vertex only
    # an object on the image border has no boundary outside the image
    lab = np.zeros((4, 4), np.uint32)
    lab[0:2, 0:2] = 3
    want = np.zeros((4, 4), bool)
    want[2, 0:2] = want[0:2, 2] = True
    assert np.array_equal(find_outer_boundaries(lab), want)


@pytest.mark.parametrize("seed", range(6))
def test_outer_boundaries_match_the_morphological_restatement(seed):
    g = np.random.default_rng(seed)
    lab = _random_labels(g, (48, 61), 9, 12)
    assert np.array_equal(find_outer_boundaries(lab), G.find_boundaries_outer(lab))
    assert np.array_equal(find_outer_boundaries(lab > 0), G.find_boundaries_outer(lab > 0))


def test_label_regions():
    lab = np.zeros((6, 7), np.uint32)
    lab[1:3, 2:6] = 1
    lab[4, 0] = 3                                                              # label 2 is absent
    regs = label_regions(lab)
    assert [(r[0], r[2]) for r in regs] == [(1, 8), (3, 1)]
    assert regs[0][1] == (slice(1, 3), slice(2, 6)) and regs[1][1] == (slice(4, 5), slice(0, 1))
    assert label_regions(np.zeros((3, 3), np.uint32)) == []


def test_blockwise_distance_transform():
    g = np.random.default_rng(0)
    m = g.random((40, 52)) > 0.08
    # one block: the exact transform (brute force nearest zero)
    assert np.allclose(blockwise_distance_transform(m), G.brute_force_edt(m), atol=1e-5)
    # several blocks: each block sees itself + halo only; compare with the loop restatement on brute-force EDT
    got = blockwise_distance_transform(m, halo=(3, 4), block_shape=(16, 20))
    ref = G.distance_transform_blockwise(m, halo=(3, 4), block_shape=(16, 20), edt=G.brute_force_edt)
    assert np.allclose(got, ref, atol=1e-5)
    # distances are exact wherever the nearest zero lies inside the halo
    exact = G.brute_force_edt(m)
    near = exact <= 3
    assert np.allclose(got[near], exact[near], atol=1e-5)


def test_three_disk_fixture_prompts_are_the_disk_centres():
    """The reference's own fixture (test/test_instance_segmentation.py:20-39): ideal decoder maps of the three disks give
    one prompt per disk, at the disk centre, in label order; points are (x, y)."""
    mask, _ = three_disk_fixture(256)
    fg, center, boundary = G.decoder_maps_from_labels(mask)
    prompts = IS._derive_point_prompts(fg, center, boundary)
    assert prompts["points"].shape == (3, 1, 2) and prompts["point_labels"].shape == (3, 1)
    assert prompts["points"][:, 0].tolist() == [[64, 64], [128, 128], [192, 192]]
    assert (prompts["point_labels"] == 1).all()
    # nothing above the foreground threshold -> no prompts
    assert IS._derive_point_prompts(fg * 0.3, center, boundary) is None


@pytest.mark.parametrize("seed,shape", [(0, (64, 80)), (1, (96, 70)), (2, (130, 128)), (3, (600, 530))])
def test_derive_point_prompts_matches_oracle(seed, shape):
    g = np.random.default_rng(seed)
    # smooth random maps -> irregular, touching, border-hugging seed components
    def field():
        f = g.random((shape[0] // 8 + 2, shape[1] // 8 + 2))
        f = np.kron(f, np.ones((8, 8)))[:shape[0], :shape[1]]
        from scipy.ndimage import uniform_filter
        return uniform_filter(f, 7).astype("float32")
    fg, center, boundary = field(), field(), field()
    small = shape[0] * shape[1] <= 100 * 100
    kw = dict(foreground_threshold=0.45, center_distance_threshold=0.55, boundary_distance_threshold=0.55)
    got = IS._derive_point_prompts(fg, center, boundary, **kw)
    ref = G.derive_point_prompts(fg, center, boundary, **kw, edt=G.brute_force_edt if small else None)
    assert got is not None and len(got["points"]) >= 3
    assert np.array_equal(got["points"], ref["points"]) and np.array_equal(got["point_labels"], ref["point_labels"])
    # every prompt lies inside its own seed component, components and prompts are in one-to-one label order
    seeds = (center < 0.55) & (boundary < 0.55) & ~(fg < 0.45)
    cc = A.label_components(seeds.astype("uint32"))
    ids = [int(cc[y, x]) for x, y in got["points"][:, 0]]
    assert ids == list(range(1, int(cc.max()) + 1))


def test_derive_box_prompts_known_answer():
    preds = [{"segmentation": np.zeros((100, 200), bool), "bbox": [10, 20, 50, 40]},
             {"segmentation": np.zeros((100, 200), bool), "bbox": [0, 0, 150, 99]}]
    got = IS._derive_box_prompts(preds, 0.1)["boxes"]
    assert np.allclose(got, [[5.0, 16.0, 65.0, 64.0], [0.0, 0.0, 100.0, 108.9]])     # clipped at shape[0] / shape[1] as in the reference
    assert np.array_equal(got, G.derive_box_prompts(preds, 0.1)["boxes"])


class _NoPredictor:
    pass


def test_state_handling_and_empty_results():
    mask, _ = three_disk_fixture(128)
    fg, center, boundary = G.decoder_maps_from_labels(mask)
    apg = IS.AutomaticPromptGenerator(_NoPredictor(), decoder=None)
    assert not apg.is_initialized
    with pytest.raises(RuntimeError):
        apg.generate()
    with pytest.raises(RuntimeError):
        apg.get_state()
    apg.set_state({"foreground": fg, "center_distances": center, "boundary_distances": boundary})
    assert apg.is_initialized and set(apg.get_state()) == {"foreground", "center_distances", "boundary_distances"}
    # no prompt survives the thresholds: empty label image / empty list, the predictor is never touched
    out = apg.generate(foreground_threshold=2.0)
    assert out.shape == mask.shape and out.dtype == np.uint32 and not out.any()
    assert apg.generate(foreground_threshold=2.0, output_mode="binary_mask") == []
    assert apg.generate(prompt_function=lambda **kw: None).sum() == 0
    apg.clear_state()
    assert not apg.is_initialized
    tiled = IS.TiledAutomaticPromptGenerator(_NoPredictor(), decoder=None)
    with pytest.raises(RuntimeError):
        tiled.generate()
    with pytest.raises(NotImplementedError):
        tiled.get_state()
    tiled._foreground, tiled._center_distances, tiled._boundary_distances, tiled._is_initialized = fg, center, boundary, True
    assert not tiled.generate(foreground_threshold=2.0).any()
    with pytest.raises(ValueError):
        tiled.generate(optimize_memory=True, output_mode="binary_mask")


def test_to_masks_records():
    seg = np.zeros((20, 30), np.uint32)
    seg[2:6, 3:10] = 1
    seg[10:12, 20:21] = 4
    recs = IS.InstanceSegmentationWithDecoder(_NoPredictor(), None)._to_masks(seg, "binary_mask")
    assert [r["seg_id"] for r in recs] == [1, 4] and [r["area"] for r in recs] == [28, 2]
    assert recs[0]["bbox"] == [3, 7, 2, 4] and recs[0]["crop_box"] == [0, 30, 0, 20]          # [x0, w, y0, h]
    assert np.array_equal(recs[1]["segmentation"], seg == 4)
    with pytest.raises(ValueError):
        IS.InstanceSegmentationWithDecoder(_NoPredictor(), None)._to_masks(seg, "rle")


def test_generator_factory_modes():
    p = _NoPredictor()
    dec = object()
    assert isinstance(IS.get_instance_segmentation_generator(p, False, decoder=dec, segmentation_mode="apg"),
                      IS.AutomaticPromptGenerator)
    assert isinstance(IS.get_instance_segmentation_generator(p, True, decoder=dec, segmentation_mode="APG"),
                      IS.TiledAutomaticPromptGenerator)
    # the reference's default with a decoder is the watershed ("ais", instance_segmentation.py:44)
    assert type(IS.get_instance_segmentation_generator(p, False, decoder=dec)) is IS.InstanceSegmentationWithDecoder
    with pytest.raises(AssertionError):                        # reference: `assert decoder is not None`
        IS.get_instance_segmentation_generator(p, False, segmentation_mode="apg")
    with pytest.raises(ValueError):
        IS.get_instance_segmentation_generator(p, False, decoder=dec, segmentation_mode="xyz")


def _tiled_records(g, n, image_shape=(300, 420), tile=(160, 200)):
    """Tile-local records the way batched_tiled_inference emits them: masks of tile (outer block) shape, bbox in the tile,
    global_bbox = bbox + tile origin; neighbouring tiles overlap so objects meet across tiles."""
    origins = [(0, 0), (0, 180), (120, 0), (120, 180), (100, 220)]
    recs = []
    for _ in range(n):
        oy, ox = origins[g.integers(0, len(origins))]
        th, tw = min(tile[0], image_shape[0] - oy), min(tile[1], image_shape[1] - ox)
        m = np.zeros((th, tw), bool)
        cy, cx, r = g.integers(5, th - 5), g.integers(5, tw - 5), g.integers(4, 30)
        yy, xx = np.mgrid[0:th, 0:tw]
        m[(yy - cy) ** 2 + ((xx - cx) * g.uniform(0.6, 1.4)) ** 2 < r * r] = True
        ys, xs = np.nonzero(m)
        bbox = [int(xs.min()), int(ys.min()), int(xs.max() - xs.min()), int(ys.max() - ys.min())]
        recs.append({"segmentation": m, "bbox": bbox, "global_bbox": [bbox[0] + ox, bbox[1] + oy, bbox[2], bbox[3]],
                     "predicted_iou": float(g.uniform(0.5, 1.0)), "stability_score": float(g.uniform(0.5, 1.0))})
    return recs


@pytest.mark.parametrize("seed", range(4))
@pytest.mark.parametrize("iomin", [False, True])
def test_apply_nms_tile_local_records_match_oracle(seed, iomin):
    """util.apply_nms on records with a ``global_bbox`` (reference util.py:1769-1846, 1876-1957): window overlap of the
    global boxes, greedy suppression, merge through the global boxes.  Host arithmetic in the reference as well."""
    g = np.random.default_rng(seed)
    recs = _tiled_records(g, 60)
    for kw in (dict(min_size=0), dict(min_size=30, nms_thresh=0.5), dict(min_size=10, nms_thresh=0.3, max_size=1500)):
        got = util.apply_nms([dict(r) for r in recs], intersection_over_min=iomin, **kw)
        ref = A.apply_nms([dict(r) for r in recs], intersection_over_min=iomin, **kw)
        assert got.shape == ref.shape == A.infer_tiled_shape(recs) and got.dtype == np.uint32
        assert np.array_equal(got, ref)
        assert got.max() > 0
    # the NMS does something on this input: a low threshold removes records that a threshold of 1 keeps
    assert not np.array_equal(util.apply_nms([dict(r) for r in recs], min_size=0, nms_thresh=0.2, intersection_over_min=iomin),
                              util.apply_nms([dict(r) for r in recs], min_size=0, nms_thresh=1.0, intersection_over_min=iomin))
    assert np.array_equal(util.apply_nms([dict(r) for r in recs], min_size=0, shape=(310, 430)),
                          A.apply_nms([dict(r) for r in recs], min_size=0, shape=(310, 430)))
    # everything filtered out
    assert not util.apply_nms([dict(r) for r in recs], min_size=10 ** 6).any()


def test_tiled_overlap_uses_the_window_of_the_global_boxes():
    """Known answer: two 10x10 squares in different tiles that coincide in the image.  The window spans the xywh boxes,
    whose w / h are max - min (the last row / column is outside), so the intersection is 9*9 of area 100."""
    a = np.zeros((40, 40), bool); a[5:15, 5:15] = True
    b = np.zeros((40, 40), bool); b[25:35, 15:25] = True
    recs = [{"segmentation": a, "bbox": [5, 5, 9, 9], "global_bbox": [105, 25, 9, 9]},
            {"segmentation": b, "bbox": [15, 25, 9, 9], "global_bbox": [105, 25, 9, 9]}]
    sc = util._tiled_overlap_scores([a, b], np.array([r["bbox"] for r in recs]), np.array([r["global_bbox"] for r in recs]), False)
    assert sc == {(0, 1): np.float32(81.0) / np.float32(119.0)}
    ref = A.tiled_mask_overlap_matrix([a, b], [r["bbox"] for r in recs], [r["global_bbox"] for r in recs], False)
    assert abs(ref[0, 1] - 81.0 / 119.0) < 1e-12


def test_tiled_generator_composition_on_the_host(monkeypatch):
    """TiledAutomaticPromptGenerator.generate from the prompts on: with the device decode replaced by canned tile-local
    records, everything downstream (tile-local NMS, merge through the global boxes, record output) is host arithmetic."""
    from micro_sam_amd import inference
    g = np.random.default_rng(5)
    recs = _tiled_records(g, 40)
    shape = (300, 420)
    seen = {}

    def canned(predictor, image, batch_size, image_embeddings=None, points=None, point_labels=None, **kw):
        seen.update(kw, n=len(points), batch_size=batch_size)
        return [dict(r) for r in recs]
    monkeypatch.setattr(inference, "batched_tiled_inference", canned)
    mask = np.zeros(shape, np.uint32)
    mask[40:90, 50:120] = 1
    mask[150:220, 200:330] = 2
    fg, center, boundary = G.decoder_maps_from_labels(mask)
    tapg = IS.TiledAutomaticPromptGenerator(_NoPredictor(), decoder=None)
    tapg._foreground, tapg._center_distances, tapg._boundary_distances = fg, center, boundary
    tapg._image_embeddings, tapg._is_initialized = None, True
    seg = tapg.generate(min_size=10, nms_threshold=0.6)
    assert seen["n"] == 2 and seen["batch_size"] == 32 and seen["optimize_memory"] is False and seen["multimasking"] is False
    assert np.array_equal(seg, A.apply_nms([dict(r) for r in recs], shape=shape, min_size=10, nms_thresh=0.6))
    masks = tapg.generate(min_size=10, nms_threshold=0.6, output_mode="binary_mask")
    assert len(masks) == int(seg.max()) and all(np.array_equal(m["segmentation"], seg == m["seg_id"]) for m in masks)
