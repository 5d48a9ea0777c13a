"""Per-instance mask IoU between two arithmetic paths of the AMG hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py`` (its ``cpu_baseline`` / ``mask_iou_vs_ref`` leg,
outside the timed region) may import this; the product never does.

``north_star`` (BASELINE.json): "mask IoU >= 0.999 per instance, identical instance ids for integer
post-processing" against the reference CPU path.  The instances of ``AutomaticMaskGenerator`` are the candidate masks
(3 per grid prompt) that survive ``AMGBase._postprocess_batch`` (``micro_sam/instance_segmentation.py:99-144``:
predicted-IoU and stability thresholds, crop-edge filter, box NMS).  Candidates of the two paths correspond one to
one (same prompt, same mask token), so no matching heuristic is needed:

* ``iou``: for every candidate the REFERENCE keeps, IoU of the reference mask and the tested path's mask of the same
  candidate (whether or not the tested path keeps it);
* ``keep_set``: candidates kept by both / only by the reference / only by the tested path (a predicted IoU or a
  stability score that crosses its threshold, or an NMS decision that flips);
* ``labels``: agreement of the two final label images (``util.mask_data_to_segmentation``,
  ``micro_sam/util.py:1773-1848``): fraction of pixels carrying the same instance id.
"""
from __future__ import annotations

from copy import deepcopy
from typing import Any, Callable, Dict, Optional, Sequence

import numpy as np
import torch

from . import amg_ref as A
from . import pipeline_ref as PR


def kept_candidates(state: Dict[str, Any], pred_iou_thresh: float = 0.88, stability_score_thresh: float = 0.95,
                    box_nms_thresh: float = 0.7) -> np.ndarray:
    """Candidate indices (into the single crop's MaskData) that survive ``_postprocess_batch`` of the oracle, in kept order.
    ``state``: an AMG state whose crop_list[0] holds iou_preds / stability_score / boxes / points (+ anything else)."""
    src = state["crop_list"][0]
    n = int(src["iou_preds"].shape[0])
    data = A.MaskData(iou_preds=torch.as_tensor(src["iou_preds"]).detach().cpu().float().clone(),
                      stability_score=torch.as_tensor(src["stability_score"]).detach().cpu().float().clone(),
                      boxes=torch.as_tensor(src["boxes"]).detach().cpu().clone(),
                      points=torch.as_tensor(src["points"]).detach().cpu().clone(),
                      rles=[None] * n, cand=torch.arange(n))
    out = PR.postprocess_batch(data, state["crop_boxes"][0], state["original_size"], pred_iou_thresh,
                               stability_score_thresh, box_nms_thresh)
    return out["cand"].numpy()


def _iou(a: np.ndarray, b: np.ndarray) -> float:
    union = int(np.count_nonzero(a | b))
    return 1.0 if union == 0 else float(np.count_nonzero(a & b)) / union


def iou_report(kept_ref: Sequence[int], kept_test: Sequence[int], ref_mask: Callable[[int], np.ndarray],
               test_mask: Callable[[int], np.ndarray], ref_scores: Optional[Dict[str, np.ndarray]] = None,
               test_scores: Optional[Dict[str, np.ndarray]] = None, worst: int = 10) -> Dict[str, Any]:
    """``ref_mask(i)`` / ``test_mask(i)``: dense bool mask of candidate i in either path."""
    kept_ref = np.asarray(kept_ref, dtype=np.int64)
    kept_test = np.asarray(kept_test, dtype=np.int64)
    ious, areas = np.ones(len(kept_ref)), np.zeros(len(kept_ref), dtype=np.int64)
    for j, i in enumerate(kept_ref):
        a, b = ref_mask(int(i)), test_mask(int(i))
        ious[j] = _iou(a, b)
        areas[j] = int(np.count_nonzero(a))
    sr, st = set(kept_ref.tolist()), set(kept_test.tolist())
    rep: Dict[str, Any] = {
        "n_instances": int(len(kept_ref)),
        "frac_ge_0.999": float(np.mean(ious >= 0.999)) if len(ious) else 1.0,
        "frac_ge_0.99": float(np.mean(ious >= 0.99)) if len(ious) else 1.0,
        "min": float(ious.min()) if len(ious) else 1.0,
        "p01": float(np.quantile(ious, 0.01)) if len(ious) else 1.0,
        "median": float(np.median(ious)) if len(ious) else 1.0,
        "mean": float(ious.mean()) if len(ious) else 1.0,
        "keep_set": {"both": len(sr & st), "ref_only": len(sr - st), "test_only": len(st - sr)},
    }
    order = np.argsort(ious)[:worst]
    rows = []
    for j in order:
        i = int(kept_ref[j])
        row = {"candidate": i, "iou": float(ious[j]), "area": int(areas[j])}
        if ref_scores is not None and test_scores is not None:
            for k in ref_scores:
                row[k] = [float(ref_scores[k][i]), float(test_scores[k][i])]
        rows.append(row)
    rep["worst"] = rows
    rep["_ious"] = ious
    rep["_areas"] = areas
    return rep


def label_agreement(seg_ref: np.ndarray, seg_test: np.ndarray) -> Dict[str, Any]:
    """Pixels carrying the same instance id in the two label images; same-id fraction over the reference's foreground."""
    same = seg_ref == seg_test
    fg = (seg_ref > 0) | (seg_test > 0)
    return {"instances_ref": int(seg_ref.max()), "instances_test": int(seg_test.max()),
            "identical_id_frac": float(same.mean()),
            "identical_id_frac_foreground": float(same[fg].mean()) if fg.any() else 1.0,
            "foreground_agreement": float(((seg_ref > 0) == (seg_test > 0)).mean())}


def oracle_mask_fn(state: Dict[str, Any]) -> Callable[[int], np.ndarray]:
    rles = state["crop_list"][0]["rles"]
    return lambda i: A.rle_to_mask(rles[i])


def oracle_scores(state: Dict[str, Any]) -> Dict[str, np.ndarray]:
    d = state["crop_list"][0]
    return {"iou_pred": torch.as_tensor(d["iou_preds"]).detach().cpu().float().numpy(),
            "stability": torch.as_tensor(d["stability_score"]).detach().cpu().float().numpy()}


def public(rep: Dict[str, Any]) -> Dict[str, Any]:
    """The report without its raw arrays (JSON-serialisable)."""
    return {k: v for k, v in rep.items() if not k.startswith("_")}


def merge_reports(reps: Sequence[Dict[str, Any]]) -> Dict[str, Any]:
    """Pool the per-tile reports (raw IoU arrays) into one distribution."""
    ious = np.concatenate([r["_ious"] for r in reps]) if reps else np.ones(0)
    ks = {k: sum(r["keep_set"][k] for r in reps) for k in ("both", "ref_only", "test_only")}
    worst = sorted((w for r in reps for w in r["worst"]), key=lambda w: w["iou"])[:10]
    return {"n_instances": int(len(ious)), "frac_ge_0.999": float(np.mean(ious >= 0.999)) if len(ious) else 1.0,
            "frac_ge_0.99": float(np.mean(ious >= 0.99)) if len(ious) else 1.0,
            "min": float(ious.min()) if len(ious) else 1.0, "p01": float(np.quantile(ious, 0.01)) if len(ious) else 1.0,
            "median": float(np.median(ious)) if len(ious) else 1.0, "mean": float(ious.mean()) if len(ious) else 1.0,
            "keep_set": ks, "worst": worst, "_ious": ious}
