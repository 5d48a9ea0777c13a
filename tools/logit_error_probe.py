"""GPU-box probe (round 4): low-res logit error of the HIP mask decoder and of the oracle's emulation of its rounding policy against the fp32 oracle, on the designed
checkpoint, after 100 fine-tuning steps and after a 1 % random perturbation of every weight (profiles/r04_experiments.md section 3)."""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import trained_parity as TP
from micro_sam_amd import util, _lib
from micro_sam_amd.synthetic import synthetic_state_dict, synthetic_tile
from oracle import sam_ref as S, amg_ref as A, pipeline_ref as PR
S.DECODER_DTYPE = _lib.decoder_dtype()
torch.set_num_threads(32)
base = synthetic_state_dict("vit_b", 0, variant="cells")
trained, _ = TP.train_checkpoint(100, 0, 1e-5)
g = torch.Generator().manual_seed(1)
noisy = {k: (v + v.float().abs().mean() * 0.01 * torch.randn(v.shape, generator=g)).to(v.dtype) if v.is_floating_point() and v.dim() >= 1 else v for k, v in base.items()}
tile = synthetic_tile(1000); img = A.to_image(tile)
pts = torch.rand(48, 1, 2, generator=g) * 1000 + 12; lbl = torch.ones(48, 1, dtype=torch.int)
for name, sd in (("designed", base), ("trained100", trained), ("designed+1%noise", noisy)):
    feats, osz, isz = PR.compute_embeddings(sd, [img], "vit_b", "fp32")
    with torch.no_grad():
        _, iou32, low32 = S.predict_torch(sd, feats, (1024, 1024), (1024, 1024), pts, lbl, return_logits=True, precision="fp32")
        _, iouE, lowE = S.predict_torch(sd, feats, (1024, 1024), (1024, 1024), pts, lbl, return_logits=True, precision="bf16")
    p = util.get_sam_model("vit_b", device="cuda", state_dict=sd)
    low, iou = p.model.decode(feats.cuda(), pts.cuda(), lbl.cuda())
    low = low.float().cpu()
    def st(a, b):
        d = (a - b).abs(); return {"mean": round(float(d.mean()), 4), "p99": round(float(d.flatten().quantile(0.99)), 4), "max": round(float(d.max()), 3)}
    near = (low32.abs() < 1).float().mean()
    print(json.dumps({"weights": name, "logit_abs_mean": round(float(low32.abs().mean()), 2), "frac_abs_lt_1": round(float(near), 4),
                      "hip_vs_fp32": st(low, low32), "emu_vs_fp32": st(lowE, low32), "hip_vs_emu": st(low, lowE),
                      "sign_flips_hip": round(float(((low > 0) != (low32 > 0)).float().mean()), 5), "sign_flips_emu": round(float(((lowE > 0) != (low32 > 0)).float().mean()), 5)}), flush=True)
