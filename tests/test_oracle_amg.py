"""Pin the integer post-processing oracle against the reference's known-answer tests."""
import numpy as np
import torch

from oracle import amg_ref as A


def test_batched_mask_to_box_known_answer():
    # reference: test/test_vendored.py:12-25
    mask = np.zeros((10, 10), dtype=bool)
    mask[7:9, 3:5] = True
    box = A.batched_mask_to_box(torch.as_tensor(mask))
    assert box.tolist() == [3, 7, 4, 8]
    assert A.batched_mask_to_box(torch.zeros(2, 5, 5, dtype=torch.bool)).tolist() == [[0, 0, 0, 0]] * 2


def _rle_python(mask_1d):
    counts, val, cnt = ([] if mask_1d[0] == 0 else [0]), mask_1d[0], 0
    for m in mask_1d:
        if m == val:
            cnt += 1
        else:
            counts.append(cnt); val = m; cnt = 1
    counts.append(cnt)
    return counts


def _random_shapes(rng, shape, n):
    h, w = shape
    out = np.zeros((n,) + shape, dtype=bool)
    yy, xx = np.mgrid[0:h, 0:w]
    for i in range(n):
        cy, cx, r = rng.integers(0, h), rng.integers(0, w), rng.integers(4, 40)
        if i % 2 == 0:
            out[i] = (yy - cy) ** 2 + (xx - cx) ** 2 < r * r
        else:
            out[i, max(cy - r, 0):cy + r, max(cx - r, 0):cx + r] = True
    return out


def test_rle_matches_independent_restatement_and_sums():
    # reference: test/test_vendored.py:63-78 (shape 128x256, 6 random shapes; sum(counts) == H*W; == upstream)
    rng = np.random.default_rng(0)
    masks = _random_shapes(rng, (128, 256), 6)
    masks[0, 0, 0] = True   # a mask that starts with 1 -> counts start with 0
    rles = A.mask_to_rle(torch.from_numpy(masks))
    for m, rle in zip(masks, rles):
        assert rle["size"] == [128, 256]
        assert sum(rle["counts"]) == 128 * 256
        assert rle["counts"] == _rle_python(m.T.reshape(-1))
        assert np.array_equal(A.rle_to_mask(rle), m)
        assert A.area_from_rle(rle) == int(m.sum())


def test_nms_known_cases():
    boxes = torch.tensor([[0, 0, 10, 10], [1, 1, 11, 11], [20, 20, 30, 30], [0, 0, 10, 10]], dtype=torch.float)
    scores = torch.tensor([0.9, 0.95, 0.5, 0.9])
    keep = A.nms(boxes, scores, 0.5)
    assert keep.tolist() == [1, 2]          # box 0 and 3 overlap box 1 with IoU 0.68
    keep = A.nms(boxes, scores, 0.7)
    assert keep.tolist() == [1, 0, 2]       # equal boxes 0/3: the first (stable order) wins


def test_point_grid_and_crop_boxes():
    g = A.build_all_layer_point_grids(32, 0, 1)[0]
    assert g.shape == (1024, 2)
    assert np.allclose(g[0], [1 / 64, 1 / 64]) and np.allclose(g[33], [3 / 64, 3 / 64])
    boxes, layers = A.generate_crop_boxes((480, 640), 0, 512 / 1500)
    assert boxes == [[0, 0, 640, 480]] and layers == [0]


def test_label_components_and_segmentation_merge():
    seg = np.zeros((8, 8), dtype="uint32")
    seg[0:2, 0:2] = 5
    seg[0:2, 2:4] = 7          # touches the first region but has another value -> separate component
    seg[5:8, 5:8] = 5          # same value, disconnected -> separate component
    lab = A.label_components(seg)
    assert lab[0, 0] == 1 and lab[0, 2] == 2 and lab[6, 6] == 3 and lab[4, 4] == 0
    m1 = np.zeros((8, 8), bool); m1[0:4, 0:4] = True
    m2 = np.zeros((8, 8), bool); m2[1:3, 1:3] = True
    masks = [{"segmentation": m1, "area": 16}, {"segmentation": m2, "area": 4}]
    out = A.mask_data_to_segmentation(masks, shape=(8, 8), with_background=True, merge_exclusively=False)
    # background (0, 48 px) is the largest -> removed (stays 0); ring of m1 -> 1, m2 -> 2
    assert out.dtype == np.uint32 and out.max() == 2 and out[0, 0] == 1 and out[1, 1] == 2 and out[7, 7] == 0


def test_to_image_matches_reference_formula():
    rng = np.random.default_rng(0)
    x = rng.random((64, 48)).astype("float64") * 1000
    y = A.to_image(x)
    assert y.shape == (64, 48, 3) and y.dtype == np.uint8
    xf = x.astype("float32"); xf = xf - xf.min(); xf = xf / (xf.max() + 1e-7)
    assert np.array_equal(y[..., 0], (xf * 255).astype("uint8"))
    assert y.max() == 254 or y.max() == 255


def test_stability_score():
    m = torch.tensor([[[2.0, 0.5], [-0.5, -2.0]]])
    assert abs(A.calculate_stability_score(m, 0.0, 1.0).item() - 1 / 3) < 1e-7


def test_apply_nms_tiled_known_answer_and_mask_nms():
    """The reference's known-answer test test/test_util.py:81-104 (two border masks of a tiled prediction -> shape (4, 7), two
    instances) plus self-consistency of the full-size mask NMS: IoU / IoMin suppression in score order."""
    import torch
    preds = [
        {"segmentation": torch.ones((4, 4), dtype=torch.bool), "bbox": [0, 0, 4, 4], "global_bbox": [0, 0, 4, 4],
         "predicted_iou": 1.0, "stability_score": 1.0},
        {"segmentation": torch.ones((4, 2), dtype=torch.bool), "bbox": [0, 0, 2, 4], "global_bbox": [5, 0, 2, 4],
         "predicted_iou": 1.0, "stability_score": 1.0},
    ]
    seg = A.apply_nms(preds, min_size=0)
    assert seg.shape == (4, 7) and seg.max() == 2 and seg.dtype == np.uint32
    assert (seg[:, :4] == 1).all() and (seg[:, 4] == 0).all() and (seg[:, 5:] == 2).all()
    # full-size masks: a duplicate with a lower score is suppressed, a disjoint one survives
    a = np.zeros((16, 16), bool); a[2:8, 2:8] = True
    b = np.zeros((16, 16), bool); b[3:8, 2:8] = True                      # IoU 30/36 with a
    c = np.zeros((16, 16), bool); c[10:14, 10:14] = True
    inner = np.zeros((16, 16), bool); inner[3:5, 3:5] = True              # inside a: IoU 4/36, IoMin 1
    def rec(m, score):
        ys, xs = np.nonzero(m)
        return {"segmentation": m, "bbox": [int(xs.min()), int(ys.min()), int(xs.max() - xs.min() + 1), int(ys.max() - ys.min() + 1)],
                "predicted_iou": score, "stability_score": 1.0}
    seg = A.apply_nms([rec(a, 0.9), rec(b, 0.8), rec(c, 0.7)], min_size=0, nms_thresh=0.7)
    assert seg.max() == 2 and (seg[a] == 1).all() and (seg[c] == 2).all()
    seg_keep = A.apply_nms([rec(a, 0.9), rec(b, 0.8), rec(c, 0.7)], min_size=0, nms_thresh=0.9)     # 30/36 <= 0.9: b survives
    assert seg_keep.max() == 2                       # b lies inside a: merged exclusively it adds no pixels, c is instance 2
    iou_seg = A.apply_nms([rec(a, 0.9), rec(inner, 0.8)], min_size=0, nms_thresh=0.5)               # IoU 0.11: both kept
    iomin_seg = A.apply_nms([rec(a, 0.9), rec(inner, 0.8)], min_size=0, nms_thresh=0.5, intersection_over_min=True)
    assert iou_seg.max() == 1 and iomin_seg.max() == 1                    # merged exclusively: the inner mask is covered either way
    m = A.mask_overlap_matrix(np.stack([a, b, c, inner]), np.array([[2, 2, 8, 8], [2, 3, 8, 8], [10, 10, 14, 14], [3, 3, 5, 5]]), False)
    assert np.isclose(m[0, 1], 30 / 36) and m[0, 2] == 0 and np.isclose(m[0, 3], 4 / 36) and np.allclose(np.diag(m), 1)
    mm = A.mask_overlap_matrix(np.stack([a, inner]), np.array([[2, 2, 8, 8], [3, 3, 5, 5]]), True)
    assert np.isclose(mm[0, 1], 1.0, atol=1e-5)
    assert list(A.greedy_matrix_nms(m, np.array([0.9, 0.8, 0.7, 0.6]), 0.7)) == [0, 2, 3]
    # box NMS variant goes through the same torchvision-style batched_nms as the AMG path
    seg_box = A.apply_nms([rec(a, 0.9), rec(b, 0.8), rec(c, 0.7)], min_size=0, perform_box_nms=True, nms_thresh=0.7)
    assert seg_box.max() == 2


def test_component_numbering_follows_the_blockwise_label_of_the_reference():
    """micro_sam/util.py:1834-1838 labels with elf.parallel.label(block_shape=(512, 512)): per-block labels with running offsets,
    union across faces, consecutive by first occurrence.  The oracle's closed form (rank of the first pixel in block-major order)
    equals that procedure restated step by step - on ragged block grids too - and the product's host function follows it; on images
    of more than one block it differs from plain raster numbering."""
    from micro_sam_amd import util
    rng = np.random.default_rng(0)
    differs = 0
    for (h, w, blk) in [(40, 52, 16), (33, 47, 16), (64, 64, 16), (20, 20, 32), (50, 30, 8), (17, 70, 16)]:
        for dens in (2, 4):
            seg = rng.integers(0, dens, size=(h, w)).astype(np.uint32)
            closed = A.label_components(seg, block=blk)
            assert np.array_equal(closed, A.label_components_literal(seg, block=blk)), (h, w, blk)
            raster = A.label_components(seg, block=1 << 30)
            assert closed.max() == raster.max() and np.array_equal(closed != 0, raster != 0)
            differs += int(not np.array_equal(closed, raster))
    assert differs >= 8
    # the reference's block size on an image of 2 x 3 blocks (ragged): the product's host function == the oracle
    seg = np.zeros((700, 1100), dtype=np.uint32)
    yy, xx = np.mgrid[0:700, 0:1100]
    for k in range(60):
        cy, cx, r = rng.integers(0, 700), rng.integers(0, 1100), rng.integers(5, 60)
        seg[(yy - cy) ** 2 + (xx - cx) ** 2 < r * r] = k + 1
    ref = A.label_components(seg)
    assert np.array_equal(util._label_equal_value_components(seg), ref)
    assert not np.array_equal(ref, A.label_components(seg, block=1 << 30))
    assert np.array_equal(A.block_major_keys(700, 1100), util._block_major_keys(700, 1100))
    assert np.array_equal(np.sort(A.block_major_keys(700, 1100).reshape(-1)), np.arange(700 * 1100))     # a permutation
