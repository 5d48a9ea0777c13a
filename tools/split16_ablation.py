"""Which fused stage of the split16 decoder carries its distance from the fp32 oracle: low-res logits of 5 point prompts on generic weights
(tests/test_gpu_strict.py's decoder case) with each fused kernel switched off in turn (micro_sam_amd.strict module flags).  One JSON line."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    from micro_sam_amd import strict, util
    from micro_sam_amd.synthetic import synthetic_state_dict
    from oracle import sam_ref as S
    sd = synthetic_state_dict("vit_b", 0, variant="cells")
    g = torch.Generator().manual_seed(77)
    for k in list(sd):
        if sd[k].dtype == torch.float32 and "gaussian" not in k and sd[k].dim() >= 1:
            sd[k] = sd[k] * (1 + 0.01 * torch.randn(sd[k].shape, generator=g))
    predictor = util.get_sam_model("vit_b", device="cuda", state_dict=sd)
    g = torch.Generator().manual_seed(6)
    feats = torch.randn(1, 256, 64, 64, generator=g) * 0.6
    P = 5
    pts = torch.rand(P, 1, 2, generator=g) * 1024
    lbl = torch.ones(P, 1, dtype=torch.int)
    with torch.no_grad():
        _, iou_r, low_r = S.predict_torch(sd, feats, (1024, 1024), (1024, 1024), pts, lbl, None, None, return_logits=True, precision="fp32")
    scale = low_r.abs().max().item()
    rec = {}

    def run(name, mode, **flags):
        saved = {k: getattr(strict, k) for k in flags}
        for k, v in flags.items():
            setattr(strict, k, v)
        predictor.set_precision(mode)
        low, iou = predictor.model.decode(feats.cuda(), pts.cuda(), lbl.cuda(), None, None)
        d = (low.cpu() - low_r).abs()
        rec[name] = {"max_rel": d.max().item() / scale, "mean_rel": d.mean().item() / scale, "iou_pred_max": (iou.cpu() - iou_r).abs().max().item()}
        for k, v in saved.items():
            setattr(strict, k, v)
    run("strict", "strict")
    run("split16", "split16")
    run("split16 without fused t2i", "split16", FUSED_T2I=False)
    run("split16 without fused up2", "split16", FUSED_UP2=False)
    run("split16 without fused i2t", "split16", FUSED_I2T=False)
    run("split16 without fused kv (and t2i)", "split16", FUSED_T2I=False, FUSED_KV_SPLIT=False)
    run("split16, products only (no fused kernel)", "split16", FUSED_T2I=False, FUSED_UP2=False, FUSED_I2T=False, FUSED_KV_SPLIT=False)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
