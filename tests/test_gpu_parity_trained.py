"""Per-instance mask parity on weights that are NOT hand-designed (VERDICT r3 item 1 ii): the "cells" checkpoint after 100 AdamW steps of
this package's own trainer on synthetic cell tiles, everything trainable (tools/trained_parity.py = the checkpoint generator + seed).
Since round 6 fine-tuning is bit-reproducible (no atomics in any gradient), so the numbers below belong to ONE checkpoint; both sides - the HIP path and
the fp32 CPU oracle - are still computed here from the same freshly trained state_dict (a golden of the trained weights would be 360 MB).

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
    # Round 6: fine-tuning is reproducible (fixed-order reductions: csrc/train.hip msam_det_reduce), so these are the numbers of ONE checkpoint
    # (tools/trained_parity.checkpoint_digest: 48c4f8ff8fde61c7 with the 32-ary reduction tree; the first, sequential form of the round gave cda477f0e44ce875
    # with 40 % / 99.6 %) instead of floors under "six measured runs" (round 5: 0.25 / 0.80 / 0.85).  Measured: default mode 265 instances at the lower quartile
    # of the reference's predictions, 68.7 % >= 0.999, 95.1 % >= 0.99, min 0.944, median 1.0, keep set 261 / 4 / 7; split16 and strict: 265 / 265 >= 0.999
    # (min 0.9994 / 0.9995: ONE pixel of a 1 700-pixel mask, where the reference's own logit is 0.009 / 0.021 of a scale of 113 and the two fp32 evaluations
    # differ by 0.017 / 0.040 - `worst_instance`), keep set 265 / 0 / 0.  Floors a few instances under the measured values.
    print("checkpoint digest:", TP.checkpoint_digest(sd))
    assert rep["n_instances"] >= 200
    assert rep["frac_ge_0.999"] >= 0.60 and rep["frac_ge_0.99"] >= 0.90 and rep["median"] >= 0.999 and rep["min"] >= 0.92, pub
    ks = rep["keep_set"]
    assert ks["ref_only"] + ks["test_only"] <= 0.06 * rep["n_instances"], ks
    assert lab["foreground_agreement"] >= 0.99
    # the reference-formulation modes on the same (trained) weights: the north-star tolerance - split16 (fp16 operand pairs) and strict (fp32 kernels) alike.
    # (Instance IDS are not asserted here: one pixel can split a component of these soft masks - 893 vs 896 components - and every later id shifts.)
    for mode in ("split16", "strict"):
        st = extra[mode]
        assert st["frac_ge_0.999"] >= 0.99 and st["min"] >= 0.995, (mode, st)
        assert st["keep_set"]["ref_only"] + st["keep_set"]["test_only"] == 0, (mode, st)
        assert st["embedding_max_abs_err"] <= 2e-3 and st["iou_pred_max_abs_diff"] <= 1e-4, (mode, st)
        assert st["labels"]["foreground_agreement"] >= 0.9999, (mode, st["labels"])
    abl = extra["ablations"]
    # the hi + lo token MLP is what carries it: the plain-operand decoder of rounds 1 - 3 on the same weights
    assert abl["product_with_plain_token_mlp"]["frac_ge_0.999"] <= rep["frac_ge_0.999"] - 0.15, abl
    # encoder and decoder now contribute alike (each alone: 36 - 74 % at this threshold, 70 - 86 % at 0.5)
    assert abl["hip_decoder_on_fp32_embedding"]["frac_ge_0.999"] >= 0.25 and abl["fp32_decoder_on_hip_embedding"]["frac_ge_0.999"] >= 0.5, abl
