"""embed (oracle encoder) -> AutomaticMaskGenerator.initialize -> generate of the PRODUCT on the CPU: micro_sam_amd's own Python layer
(SamPredictor, AutomaticMaskGenerator, ops) over the library's C ABI, the kernel SOURCES running on host threads
(tests/hip_host_shim.build_library), against the oracle pipeline (fp32 CPU reference) on the same embedding: per-instance mask IoU,
keep set, scores, and the label image - the comparison tests/test_gpu_parity_iou.py makes on the device, on a 4 x 4 prompt grid.

TEST INFRASTRUCTURE: micro_sam_amd._lib (library handle, require_gpu, stream accessors) and torch.cuda.current_stream are patched for
the duration of the test; the product has no CPU path."""
import json
import os

import numpy as np
import pytest
import torch

from hip_host_shim import build_library

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GRID = 4


@pytest.fixture(scope="module")
def run(tmp_path_factory):
    from micro_sam_amd import _lib, modeling
    from micro_sam_amd.synthetic import synthetic_state_dict, synthetic_tile
    from oracle import amg_ref as A
    from oracle import parity as PT
    from oracle import pipeline_ref as PR
    os.environ["MSAM_EMU_CUS"] = "4"
    host = build_library(str(tmp_path_factory.mktemp("host_amg")), ROOT)
    for name, (res, args) in _lib._PROTOS.items():
        fn = getattr(host, name)
        fn.restype, fn.argtypes = res, args
    saved = (_lib._lib, _lib.require_gpu, _lib.stream_ptr, _lib.ptr, torch.cuda.current_stream)

    class _Stream:
        cuda_stream = 0
    _lib._lib = host
    _lib.require_gpu = lambda device=None: torch.device("cpu") if device is None else torch.device(device)
    _lib.stream_ptr = lambda: None
    _lib.ptr = lambda t: None if t is None else t.data_ptr()
    torch.cuda.current_stream = lambda *a, **k: _Stream()
    try:
        from micro_sam_amd import ops
        from micro_sam_amd.instance_segmentation import AutomaticMaskGenerator
        from micro_sam_amd.predictor import SamPredictor
        sd = synthetic_state_dict("vit_b", 0, variant="cells")
        tile = synthetic_tile(1000)
        feats, _, _ = PR.compute_embeddings(sd, [np.repeat(tile[..., None], 3, axis=2) if tile.ndim == 2 else tile])   # fp32 oracle encoder
        ref_state = PR.amg_initialize(sd, tile, feats, (1024, 1024), (1024, 1024), points_per_side=GRID, precision="fp32")
        ref_seg = np.asarray(PR.amg_generate(ref_state))
        sam = modeling.build_sam("vit_b")
        sam.load_state_dict(sd)
        sam.eval()
        amg = AutomaticMaskGenerator(SamPredictor(sam), points_per_side=GRID)
        amg.initialize(tile, {"features": feats.numpy(), "input_size": (1024, 1024), "original_size": (1024, 1024)})
        data = amg.crop_list[0]
        n = len(data)
        cand = data.shallow_copy()
        cand["cand"] = torch.arange(n)
        kept_test = amg._postprocess_batch(cand, amg.crop_boxes[0], amg.original_size, 0.88, 0.95, 0.7)["cand"].numpy()
        kept_ref = PT.kept_candidates(ref_state)
        bits = data["bits"]
        rep = PT.iou_report(kept_ref, kept_test, PT.oracle_mask_fn(ref_state), lambda i: ops.unpack_bits(bits[i:i + 1], 1024)[0].numpy(),
                            PT.oracle_scores(ref_state),
                            {"iou_pred": data["iou_preds"].float().numpy(), "stability": data["stability_score"].float().numpy()})
        seg = amg.generate()
        lab = PT.label_agreement(ref_seg.astype(np.uint32), np.asarray(seg).astype(np.uint32))
        print("\nhost-library product vs fp32 oracle:", json.dumps(PT.public(rep)), json.dumps(lab))
        yield dict(rep=rep, lab=lab, data=data, ref=ref_state, n=n, seg=seg, amg=amg, A=A, PR=PR)
    finally:
        _lib._lib, _lib.require_gpu, _lib.stream_ptr, _lib.ptr, torch.cuda.current_stream = saved
        os.environ.pop("MSAM_EMU_CUS", None)


def test_scores_and_boxes_close_to_the_reference(run):
    d, ref = run["data"], run["ref"]["crop_list"][0]
    assert run["n"] == 3 * GRID * GRID == len(ref["iou_preds"])
    di = np.abs(d["iou_preds"].float().numpy() - ref["iou_preds"].numpy())
    assert di.max() < 8e-3 and di.mean() < 2e-3
    st, sr = d["stability_score"].float().numpy(), ref["stability_score"].numpy()
    both = np.isfinite(st) & np.isfinite(sr)
    assert np.abs(st[both] - sr[both]).mean() < 3e-3
    assert (d["boxes"].numpy() == ref["boxes"].numpy()).all(axis=1).mean() > 0.9


def test_per_instance_iou_and_labels(run):
    rep, lab = run["rep"], run["lab"]
    assert rep["n_instances"] >= 4
    assert rep["min"] >= 0.99 and rep["frac_ge_0.99"] == 1.0, rep["worst"][:3]
    ks = rep["keep_set"]
    assert ks["ref_only"] + ks["test_only"] <= max(2, 0.15 * rep["n_instances"]), ks
    assert lab["foreground_agreement"] >= 0.995


def test_integer_stages_are_the_oracles_on_the_products_own_state(run):
    """generate() (filters, NMS, paint, components in the reference's numbering, relabel) of the PRODUCT's candidate state: identical
    to the oracle's generate on that state, ids included."""
    amg, A, PR = run["amg"], run["A"], run["PR"]
    d = amg.crop_list[0]
    from micro_sam_amd import ops
    masks = ops.unpack_bits(d["bits"], 1024)
    state = {"crop_list": [A.MaskData(iou_preds=d["iou_preds"].float(), stability_score=d["stability_score"].float(),
                                      boxes=d["boxes"].long(), points=torch.as_tensor(d["points"]), rles=A.mask_to_rle(masks))],
             "crop_boxes": amg.crop_boxes, "original_size": amg.original_size}
    assert np.array_equal(np.asarray(PR.amg_generate(state)).astype(np.uint32), np.asarray(run["seg"]).astype(np.uint32))


@pytest.mark.parametrize("thr", [(0.88, 0.95, 0.7), (0.0, 0.0, 0.7), (0.5, 0.9, 0.3)])
def test_one_synchronisation_form_of_postprocess_batch(run, thr):
    """generate() of tiled / cropped device states enqueues every crop's filters + NMS first (_postprocess_batch_prepare: no host
    synchronisation) and reads the survivor counts once; per crop it must leave the rows _postprocess_batch leaves, in its order."""
    amg, data = run["amg"], run["data"]
    a = data.shallow_copy()
    a["cand"] = torch.arange(len(data))
    b = data.shallow_copy()
    b["cand"] = torch.arange(len(data))
    want = amg._postprocess_batch(a, amg.crop_boxes[0], amg.original_size, *thr)
    order, count = amg._postprocess_batch_prepare(b, amg.crop_boxes[0], amg.original_size, *thr)
    got = amg._postprocess_batch_finish(b, amg.crop_boxes[0], order, int(count))
    assert torch.equal(got["cand"], want["cand"]) and len(want["cand"]) > 0
    assert torch.equal(got["boxes"], want["boxes"]) and torch.equal(got["iou_preds"], want["iou_preds"])
    assert torch.equal(torch.as_tensor(got["crop_boxes"]), torch.as_tensor(want["crop_boxes"]))
