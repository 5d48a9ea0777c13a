"""Per-instance mask parity on weights that are NOT hand-designed (VERDICT r3 item 1 ii): the "cells" checkpoint after 100 AdamW steps of
this package's own trainer on synthetic cell tiles, everything trainable (tools/trained_parity.py = the checkpoint generator + seed).
Fine-tuning on the GPU is not bit-reproducible, so both sides - the HIP path and the fp32 CPU oracle - are computed here from the same
freshly trained state_dict; no golden exists for it.

What the trained weights show (profiles/r04_experiments.md section 3): the designed checkpoint FLATTERS the 16-bit arithmetic - its weights
are exactly representable structures, and 100 steps (0.2 % relative weight change) remove that.  With the token MLP on plain fp16 operands
(rounds 1 - 3) only 26 - 29 % of the instances reach IoU >= 0.999 on the trained weights; with its operands as hi + lo pairs (round 4) 70 %.
The trained masks are soft (predicted IoU 0.3 - 0.6, stability 0.35 - 0.98: the default thresholds keep nothing), so the comparison
runs at the lower quartile of the reference's predicted IoUs (a fixed 0.5 keeps 36 or 190 instances depending on the run) / stability_score_thresh 0.8 - by the definition of the stability score a tenth of such a mask's pixels lies
within +-1 of the threshold, which is why single flipped pixels are frequent here although the logit error is 3e-4 of the logit scale."""
import json
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_parity_on_a_fine_tuned_checkpoint():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import trained_parity as TP
    from micro_sam_amd.synthetic import synthetic_state_dict
    from oracle import parity as PT
    sd, losses = TP.train_checkpoint(steps=100, seed=0, lr=1e-5)
    dist = TP.weight_distance(sd, synthetic_state_dict("vit_b", 0, variant="cells"))
    assert all(v > 1e-4 for v in dist.values()), dist                     # every part of the model moved
    assert sum(losses[-10:]) < sum(losses[:10])                           # and it trained
    rep, lab, extra = TP.compare(sd, tile_seed=1000, points_per_side=16, pred_iou_thresh=None, stability_score_thresh=0.8, ablations=True, strict=True)
    pub = PT.public(rep)
    pub.pop("worst", None)
    print("\ntrained checkpoint (100 steps):", json.dumps({"weights_moved": dist, "iou": pub, "labels": lab, **extra}))
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_trained.json"), "w") as fh:
            json.dump({"weights_moved": dist, "iou": pub, "labels": lab, **extra}, fh, indent=1)
    except OSError:
        pass
    # measured (round 4).  At a FIXED predicted-IoU threshold of 0.5 (four runs, tools/trained_parity.py --thresholds 0.5 0.8): 170 - 191 instances,
    # 70 - 75 % >= 0.999, 92 - 96 % >= 0.99, min 0.94 - 0.96, median 1.0 - and one run with 36 instances (58 %), the fine-tuned IoU head moves
    # from run to run.  At the lower quartile of the reference's predictions (what this test uses; 0.29 in the measured run): 280 instances incl.
    # the low-confidence ones, 49 % >= 0.999, 95 % >= 0.99, min 0.974, median 0.9989 (plain token MLP: 12 % / 39 % / min 0.68, median 0.985).
    # Floors leave room for the run-to-run spread of the (non-reproducible) training: round 5 measured 34.6 - 64 % >= 0.999, 88.8 - 98 % >= 0.99,
    # min 0.935 - 0.981, median 0.9979 - 0.9989 over six runs (the checkpoint differs every time: split-K atomics in the backward pass)
    assert rep["n_instances"] >= 100
    assert rep["frac_ge_0.999"] >= 0.25 and rep["frac_ge_0.99"] >= 0.80 and rep["median"] >= 0.995 and rep["min"] >= 0.85, pub
    ks = rep["keep_set"]
    assert ks["ref_only"] + ks["test_only"] <= 0.06 * rep["n_instances"], ks
    assert lab["foreground_agreement"] >= 0.99          # (0.9953 - 0.9999 over four runs: one kept mask more or less is its whole area)
    # the strict precision mode on the same (trained) weights: the reference's result up to fp32 rounding
    st = extra["strict"]
    assert st["frac_ge_0.999"] >= 0.95 and st["min"] >= 0.99, st
    assert st["keep_set"]["ref_only"] + st["keep_set"]["test_only"] <= 0.02 * st["n_instances"], st
    assert st["embedding_max_abs_err"] <= 2e-3 and st["iou_pred_max_abs_diff"] <= 1e-4, st
    abl = extra["ablations"]
    # the hi + lo token MLP is what carries it: the plain-operand decoder of rounds 1 - 3 on the same weights
    assert abl["product_with_plain_token_mlp"]["frac_ge_0.999"] <= rep["frac_ge_0.999"] - 0.15, abl
    # encoder and decoder now contribute alike (each alone: 36 - 74 % at this threshold, 70 - 86 % at 0.5)
    assert abl["hip_decoder_on_fp32_embedding"]["frac_ge_0.999"] >= 0.25 and abl["fp32_decoder_on_hip_embedding"]["frac_ge_0.999"] >= 0.5, abl
