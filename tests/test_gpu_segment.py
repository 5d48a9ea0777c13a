"""Device generate() stages (box NMS, paint, connected components, relabel): identical ids to the CPU oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


def _random_masks(rng, n, h, w):
    yy, xx = np.mgrid[0:h, 0:w]
    out = np.zeros((n, h, w), dtype=bool)
    for i in range(n):
        for _ in range(rng.integers(1, 4)):               # several blobs per mask -> several components per id
            cy, cx, r = rng.integers(0, h), rng.integers(0, w), rng.integers(3, max(4, min(h, w) // 4))
            out[i] |= (yy - cy) ** 2 + (xx - cx) ** 2 < r * r
    return out


@pytest.mark.parametrize("shape,n", [((64, 96), 5), ((300, 200), 40), ((1024, 1024), 200), ((1024, 1024), 0), ((700, 1100), 120)])
@pytest.mark.parametrize("with_background", [True, False])
def test_mask_data_to_segmentation_device(shape, n, with_background):
    _gpu()
    from micro_sam_amd import _vendored, util
    from oracle import amg_ref as A
    rng = np.random.default_rng(n + shape[0])
    masks = _random_masks(rng, n, *shape) if n else np.zeros((0,) + shape, dtype=bool)
    if n:
        masks[1] = masks[0]                                # equal areas: stable order matters
    areas = masks.reshape(n, shape[0] * shape[1]).sum(1)
    recs = [{"segmentation": m, "area": int(a)} for m, a in zip(masks, areas)]
    ref = A.mask_data_to_segmentation(recs, shape=shape, with_background=with_background, merge_exclusively=False)
    bits = _vendored.pack_bits(torch.as_tensor(masks).cuda()) if n else torch.zeros((0, (shape[0] + 31) // 32, shape[1]), dtype=torch.int32, device="cuda")
    got = util.mask_data_to_segmentation_device(bits, torch.as_tensor(areas, dtype=torch.int32).cuda(), shape,
                                                with_background=with_background)
    assert got.dtype == np.uint32 and got.shape == tuple(shape)
    assert np.array_equal(got, ref), f"{(got != ref).sum()} pixels differ, max ids {got.max()} vs {ref.max()}"
    if n:      # the fallback (union passes iterated under host control, torch-operator relabel) gives the same image
        assert np.array_equal(util._mask_data_to_segmentation_device_iterative(
            bits, torch.as_tensor(areas, dtype=torch.int32).cuda(), shape, with_background=with_background), ref)
    # the host implementation of the product agrees as well
    host = util.mask_data_to_segmentation(recs, shape=shape, with_background=with_background, merge_exclusively=False)
    assert np.array_equal(host, ref)


def test_label_components_worst_cases():
    _gpu()
    from micro_sam_amd import ops
    from oracle import amg_ref as A
    rng = np.random.default_rng(0)
    cases = [rng.integers(0, 3, size=(257, 130)), np.ones((64, 64), dtype=int), np.zeros((33, 17), dtype=int),
             rng.integers(0, 3, size=(600, 700)), np.ones((1030, 520), dtype=int)]          # more than one 512 x 512 block
    spiral = np.zeros((101, 101), dtype=int)                # one long snake: deep union-find chains
    spiral[::2, :] = 1; spiral[1::4, -1] = 1; spiral[3::4, 0] = 1
    cases.append(spiral)
    for seg in cases:
        roots = ops.label_components(torch.as_tensor(seg, dtype=torch.int32).cuda()).cpu().numpy()
        ref = A.label_components(seg.astype("uint32"))
        fg = seg.reshape(-1) != 0
        assert (roots[~fg] == -1).all()
        # same partition in the reference's numbering order (ascending root keys), and every root key is the smallest block-major
        # key of its component (csrc/common.h bm_key; = the linear index for a single 512 x 512 block)
        _, inv = np.unique(roots[fg], return_inverse=True)
        assert np.array_equal(inv + 1, ref.reshape(-1)[fg])
        keys = A.block_major_keys(*seg.shape).reshape(-1)
        pix_of_key = np.argsort(keys)
        assert (roots[fg] <= keys[fg]).all() and (roots[pix_of_key[roots[fg]]] == roots[fg]).all()


@pytest.mark.parametrize("k", [1, 63, 64, 65, 700, 3072])
def test_box_nms_matches_oracle(k):
    _gpu()
    from micro_sam_amd import ops
    from oracle import amg_ref as A
    rng = np.random.default_rng(k)
    xy = rng.integers(0, 900, size=(k, 2)).astype(np.float32)
    wh = rng.integers(0, 200, size=(k, 2)).astype(np.float32)           # includes zero-area boxes
    boxes = torch.as_tensor(np.concatenate([xy, xy + wh], axis=1))
    scores = torch.as_tensor(rng.integers(0, 50, size=k).astype(np.float32) / 50)   # many ties: stable order matters
    for thr in (0.7, 0.3):
        got = ops.box_nms(boxes.cuda(), scores.cuda(), thr).cpu()
        assert got.tolist() == A.nms(boxes, scores, thr).tolist()


@pytest.mark.parametrize("kw", [dict(), dict(pred_iou_thresh=0.5, stability_score_thresh=0.5, box_nms_thresh=0.9),
                                dict(pred_iou_thresh=0.0, stability_score_thresh=0.0, min_object_size=40),
                                dict(pred_iou_thresh=0.6, stability_score_thresh=0.7, with_background=False)])
def test_fused_generate_matches_the_operator_formulation(kw):
    """msam_amg_generate_labels (15 kernels, csrc/amgselect.hip) == the torch-operator formulation of generate_device (itself
    equal to generate() and to the oracle, tests/test_gpu_model.py) on a synthetic single-crop state with ties in scores and
    areas, boxes near the image border, empty masks (stability 0 / 0 = NaN) and tiny components."""
    _gpu()
    from micro_sam_amd._vendored import pack_bits
    from micro_sam_amd.instance_segmentation import AutomaticMaskGenerator, DeviceMaskData
    rng = np.random.default_rng(7)
    n, h, w = 600, 256, 320
    masks = np.zeros((n, h, w), dtype=bool)
    yy, xx = np.mgrid[0:h, 0:w]
    for i in range(n):
        if i % 37 == 0:
            continue                                                     # empty mask
        cy, cx, r = rng.integers(0, h), rng.integers(0, w), rng.integers(2, 30)
        masks[i] = (yy - cy) ** 2 + (xx - cx) ** 2 < r * r
        if i % 5 == 0:
            masks[i, rng.integers(0, h), rng.integers(0, w)] = True      # stray pixel: extra component
    m = torch.as_tensor(masks).cuda()
    area = m.flatten(1).sum(1).to(torch.int32)
    from micro_sam_amd._vendored import batched_mask_to_box
    data = DeviceMaskData(mask_size=(h, w), full_size=(h, w),
                          iou_preds=torch.as_tensor(rng.integers(40, 100, size=n).astype(np.float32) / 100).cuda(),   # many ties
                          points=torch.zeros(n, 2))
    num = (area.float() * torch.as_tensor(rng.integers(80, 101, size=n).astype(np.float32) / 100).cuda()).round()
    data["stability_score"] = num / area.float()                         # NaN for the empty masks
    data["boxes"] = batched_mask_to_box(m).to(torch.int32)
    data["area"] = area
    data["bits"] = pack_bits(m)
    amg = AutomaticMaskGenerator.__new__(AutomaticMaskGenerator)
    amg._is_initialized, amg._crop_list, amg._crop_boxes, amg._original_size = True, [data], [[0, 0, w, h]], (h, w)
    lab_f, flag_f = amg.generate_device(**kw)
    amg._torch_glue_generate = True
    lab_t, flag_t = amg.generate_device(**kw)
    assert int(flag_f.item()) == 0 and int(flag_t.item()) == 0
    assert lab_f.dtype == torch.int32 and torch.equal(lab_f, lab_t) and int(lab_f.max().item()) > 10


@pytest.mark.parametrize("mode", ["iou", "iomin", "box"])
def test_apply_nms_matches_oracle(mode):
    """util.apply_nms (reference micro_sam/util.py:1851-1957) with the device mask NMS (popcount of AND
    over bit masks) == the oracle's restatement (dense float matrix products), keep set and label image."""
    _gpu()
    from micro_sam_amd import util
    from oracle import amg_ref as A
    rng = np.random.default_rng(11)
    h, w = 200, 260
    yy, xx = np.mgrid[0:h, 0:w]
    preds = []
    for i in range(90):
        cy, cx = rng.integers(10, h - 10), rng.integers(10, w - 10)
        if i % 3 == 1 and preds:                                          # near-duplicate of an earlier mask (suppressed by NMS)
            q = preds[rng.integers(0, len(preds))]
            m = np.roll(q["segmentation"], (int(rng.integers(-2, 3)), int(rng.integers(-2, 3))), axis=(0, 1))
        elif i % 3 == 2 and preds:                                        # small mask inside an earlier one (IoMin ~ 1, IoU small)
            q = preds[rng.integers(0, len(preds))]
            ys, xs = np.where(q["segmentation"])
            k = rng.integers(0, len(ys))
            m = ((yy - ys[k]) ** 2 + (xx - xs[k]) ** 2 < 9) & q["segmentation"]
        else:
            r = rng.integers(4, 26)
            m = (yy - cy) ** 2 + (xx - cx) ** 2 < r * r
        if m.sum() == 0:
            continue
        ys, xs = np.where(m)
        preds.append({"segmentation": m, "bbox": [int(xs.min()), int(ys.min()), int(xs.max() - xs.min()), int(ys.max() - ys.min())],
                      "predicted_iou": float(rng.integers(50, 100)) / 100, "stability_score": float(rng.integers(80, 100)) / 100})
    kw = dict(min_size=5, nms_thresh=0.6 if mode != "iomin" else 0.8, perform_box_nms=(mode == "box"),
              intersection_over_min=(mode == "iomin"))
    got = util.apply_nms([dict(p) for p in preds], **kw)
    ref = A.apply_nms([dict(p) for p in preds], **kw)
    assert got.shape == (h, w) and got.dtype == np.uint32 and got.max() > 5
    assert np.array_equal(got, ref)
    assert np.array_equal(util.apply_nms([dict(p) for p in preds], max_size=400, **kw), A.apply_nms([dict(p) for p in preds], max_size=400, **kw))


def test_slice_overlaps_and_merge_3d():
    """ops.slice_overlaps (msam_slice_overlaps: scatter-add of consecutive-slice label pairs into a hash table in HBM) == the numpy
    contingency table, incl. pairs with the background and an object that spans two 512-row halves; then the reference's own merge
    tests (test/test_multi_dimensional_segmentation.py:15-66) through merge_instance_segmentation_3d with the device counter."""
    _gpu()
    from scipy import ndimage
    from micro_sam_amd import multi_dimensional_segmentation as M
    from micro_sam_amd import ops
    from test_merge_3d_host import _blobs, _stack, numpy_overlap_table      # tests/ is on sys.path (pytest prepend import mode)
    rng = np.random.default_rng(3)
    vol = np.zeros((6, 300, 520), dtype=np.int32)
    offset = 0
    for z in range(6):
        lab = ndimage.label(ndimage.gaussian_filter(rng.random((300, 520)), 3) > 0.52)[0]
        lab[lab != 0] += offset
        offset = max(offset, int(lab.max()))
        vol[z] = lab
    got = ops.slice_overlaps(torch.as_tensor(vol).cuda())
    ref = numpy_overlap_table(vol)
    assert got.shape == ref.shape and np.array_equal(got, ref) and (ref[:, 1] == 0).any()
    assert ops.slice_overlaps(torch.as_tensor(vol[:1]).cuda()).shape == (0, 3)            # a single slice: no pairs
    seg = _blobs(0, 512)
    stacked = _stack(seg, 5)
    merged = M.merge_instance_segmentation_3d(stacked)
    ids0 = np.unique(merged[0])
    assert len(ids0) > 10
    for z in range(1, 5):
        assert np.array_equal(ids0, np.unique(merged[z]))
    merged = M.merge_instance_segmentation_3d(_stack(seg, 5, blank=(2,)), gap_closing=1)
    for z in range(1, 5):
        assert np.array_equal(np.unique(merged[0]), np.unique(merged[z]))
