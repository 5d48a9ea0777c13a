"""Calibrate the IoU head of the synthetic "cells" checkpoint (test-data tooling; uses the CPU oracle, never shipped
in the product path): the mean predicted IoU per output token over a 16x16 prompt grid of one synthetic tile, minus the
designed bias, is stored in micro_sam_amd/data/synthetic_calib.json under "<model_type>/<seed>/cells" and subtracted by
micro_sam_amd.synthetic.synthetic_state_dict.

Usage: python tools/calibrate_cells.py [model_type] [seed]
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from micro_sam_amd.synthetic import synthetic_state_dict, synthetic_tile  # noqa: E402
from oracle import amg_ref as A  # noqa: E402
from oracle import pipeline_ref as PR  # noqa: E402
from oracle import sam_ref as S  # noqa: E402


def main():
    model_type = sys.argv[1] if len(sys.argv) > 1 else "vit_b"
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    sd = synthetic_state_dict(model_type, seed, calibrated=False, variant="cells")
    bias = sd["mask_decoder.iou_prediction_head.layers.2.bias"].clone()
    img = A.to_image(synthetic_tile(9000))
    with torch.no_grad():
        f, _, _ = PR.compute_embeddings(sd, [img], model_type, "fp32")
        pts = torch.as_tensor(A.build_all_layer_point_grids(16, 0, 1)[0] * 1024, dtype=torch.float)[:, None, :]
        lbl = torch.ones(len(pts), 1, dtype=torch.int)
        means = torch.zeros(4)
        for multi in (True, False):
            _, iou, _ = S.predict_torch(sd, f, (1024, 1024), (1024, 1024), pts, lbl, multimask_output=multi, return_logits=True)
            if multi:
                means[1:] = iou.mean(0)
            else:
                means[0] = iou.mean()
    off = (means - bias).tolist()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "micro_sam_amd", "data", "synthetic_calib.json")
    with open(path) as fh:
        calib = json.load(fh)
    calib[f"{model_type[:5]}/{seed}/cells"] = {"iou_offset": [round(v, 6) for v in off]}
    with open(path, "w") as fh:
        json.dump(calib, fh, indent=1)
    print("iou offsets", off)


if __name__ == "__main__":
    main()
