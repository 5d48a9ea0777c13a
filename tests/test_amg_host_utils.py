"""Host-side AMG utilities restated from segment_anything.utils.amg (absent here): remove_small_regions, COCO RLE string
coding, numpy RLE; checked against independent formulations and the oracle."""
import numpy as np
import torch


def test_mask_to_rle_numpy_matches_oracle():
    from micro_sam_amd import amg_utils
    from oracle import amg_ref as A
    rng = np.random.default_rng(1)
    for shape in ((7, 5), (64, 33), (1, 9)):
        for p in (0.0, 0.3, 1.0):
            m = rng.random(shape) < p
            ref = A.mask_to_rle(torch.as_tensor(m)[None])[0]
            got = amg_utils.mask_to_rle_numpy(m)
            assert got["size"] == list(shape) and list(got["counts"]) == list(ref["counts"])
            assert np.array_equal(amg_utils.rle_to_mask(got), m)


def test_coco_rle_round_trip_and_known_small_case():
    from micro_sam_amd import amg_utils
    rng = np.random.default_rng(2)
    for shape in ((10, 12), (300, 200)):
        m = rng.random(shape) < 0.4
        m[:, :3] = False
        rle = amg_utils.mask_to_rle_numpy(m)
        enc = amg_utils.coco_encode_rle(rle)
        assert isinstance(enc["counts"], str) and all(48 <= ord(c) < 48 + 64 for c in enc["counts"])
        assert amg_utils.coco_decode_rle(enc)["counts"] == list(rle["counts"])
    # cocoapi's coding of small counts is one character '0' + value: [5, 3, 2] -> "532"
    assert amg_utils.coco_encode_rle({"size": [2, 5], "counts": [5, 3, 2]})["counts"] == "532"
    # a count >= 16 needs the continuation bit: 40 = 0b01000 | (1 << 5) -> chars (8 | 32) + 48 = 'X', 1 + 48 = '1'
    assert amg_utils.coco_encode_rle({"size": [8, 5], "counts": [40]})["counts"] == "X1"


def test_remove_small_regions():
    from micro_sam_amd import amg_utils
    m = np.zeros((20, 20), dtype=bool)
    m[2:10, 2:10] = True            # 64-px island
    m[5, 5] = False                 # 1-px hole
    m[15, 15] = True                # 1-px island, 8-connected to nothing
    m[14, 14] = True                # diagonal neighbour: one 2-px island under 8-connectivity
    out, changed = amg_utils.remove_small_regions(m, 4, "holes")
    assert changed and out[5, 5] and out.sum() == m.sum() + 1
    out2, changed2 = amg_utils.remove_small_regions(out, 4, "islands")
    assert changed2 and not out2[15, 15] and not out2[14, 14] and out2[2:10, 2:10].all()
    same, changed3 = amg_utils.remove_small_regions(out2, 4, "islands")
    assert not changed3 and np.array_equal(same, out2)
    tiny = np.zeros((6, 6), dtype=bool); tiny[0, 0] = True; tiny[4, 4:6] = True
    kept, _ = amg_utils.remove_small_regions(tiny, 10, "islands")       # everything below the threshold: keep the largest
    assert kept.sum() == 2 and kept[4, 4]


def test_postprocess_batch_single_gather_equals_the_sequential_filters():
    """``AMGBase._postprocess_batch`` (reference instance_segmentation.py:99-144) filters the columns by predicted IoU, by
    stability, by the crop-edge test and by the NMS result, one after the other.  The product ANDs the three independent tests and
    gathers every column once with the final index list: same rows, same order, for tensor, numpy and list columns - checked here
    against the oracle's literal restatement on random candidates (crop inside the image: the edge test matters)."""
    from micro_sam_amd import amg_utils
    from micro_sam_amd.instance_segmentation import AutomaticMaskGenerator
    from oracle import amg_ref as A
    from oracle import pipeline_ref as PR
    g = torch.Generator().manual_seed(11)
    crop_box, original_size = [100, 50, 612, 562], (700, 800)
    for n in (0, 1, 257):
        x0 = torch.randint(0, 400, (n,), generator=g); y0 = torch.randint(0, 400, (n,), generator=g)
        w = torch.randint(1, 200, (n,), generator=g); h = torch.randint(1, 200, (n,), generator=g)
        boxes = torch.stack([x0, y0, torch.clamp(x0 + w, max=512), torch.clamp(y0 + h, max=512)], dim=1)
        boxes[::7, 0] = 0                                        # some touch the crop edge (and not the image edge)
        cols = {"iou_preds": torch.rand(n, generator=g) * 0.3 + 0.75, "stability_score": torch.rand(n, generator=g) * 0.2 + 0.85,
                "boxes": boxes, "points": torch.rand(n, 2, generator=g) * 512, "cand": torch.arange(n),
                "rles": [{"size": [512, 512], "counts": [i, 512 * 512 - i]} for i in range(n)]}
        ref = A.MaskData(**{k: (v.clone() if torch.is_tensor(v) else list(v)) for k, v in cols.items()})
        ref = PR.postprocess_batch(ref, crop_box, original_size, 0.88, 0.95, 0.7)
        got = amg_utils.MaskData(**{k: (v.clone() if torch.is_tensor(v) else list(v)) for k, v in cols.items()})
        got["area_np"] = np.arange(n) * 3                       # a numpy column as well
        got = AutomaticMaskGenerator._postprocess_batch(None, got, crop_box, original_size, 0.88, 0.95, 0.7)
        assert torch.equal(got["cand"], ref["cand"]), n
        for k in ("iou_preds", "stability_score", "boxes", "points", "crop_boxes"):
            assert torch.equal(torch.as_tensor(got[k]).double(), torch.as_tensor(ref[k]).double()), (n, k)
        assert got["rles"] == ref["rles"] and np.array_equal(got["area_np"], ref["cand"].numpy() * 3)
        if n == 257:
            assert 0 < len(ref["cand"]) < n
