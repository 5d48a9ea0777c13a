"""Which fp16 rounding site of the mask decoder carries the logit error on GENERIC weights (CPU oracle only; round 4).
The designed "cells" checkpoint understates the decoder's rounding error 12x (profiles/r04_experiments.md section 3): a 1 % random
perturbation of every weight - or 100 fine-tuning steps - removes its exactly-representable structure.  This tool rounds ONE decoder site
at a time (oracle Prec(only_sites=...)) on such weights and reports the low-res logit error against fp32.

    python tools/dec_site_ablation.py [noise_percent] [n_prompts] [--sites]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from micro_sam_amd.synthetic import synthetic_state_dict, synthetic_tile
from oracle import amg_ref as A
from oracle import pipeline_ref as PR
from oracle import sam_ref as S

S.DECODER_DTYPE = torch.float16
noise = float(sys.argv[1]) if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else 1.0
P = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 24
torch.set_num_threads(os.cpu_count() or 1)
base = synthetic_state_dict("vit_b", 0, variant="cells")
g = torch.Generator().manual_seed(1)
sd = {k: (v + v.float().abs().mean() * noise / 100 * torch.randn(v.shape, generator=g)).to(v.dtype) if v.is_floating_point() and v.dim() >= 1 else v
      for k, v in base.items()}
img = A.to_image(synthetic_tile(1000))
feats, osz, isz = PR.compute_embeddings(sd, [img], "vit_b", "fp32")
pts = torch.rand(P, 1, 2, generator=g) * 1000 + 12
lbl = torch.ones(P, 1, dtype=torch.int)


def run(prec):
    with torch.no_grad():
        return S.predict_torch(sd, feats, (1024, 1024), (1024, 1024), pts, lbl, return_logits=True, precision=prec)[2]


ref = run("fp32")
print(json.dumps({"noise_percent": noise, "prompts": P, "logit_abs_mean": round(float(ref.abs().mean()), 2)}), flush=True)
SITES = ("stream", "tok", "t2i0", "fold", "table", "probs", "foldv", "up", "head")


def with_dtype(**kw):
    pr = S.Prec("bf16")
    pr.site_dtype.update({k.replace("_", "."): v for k, v in kw.items()})
    return pr


def plain():
    pr = S.Prec("bf16")
    pr.site_dtype.pop("tok.mlp", None)
    return pr


cases = [("rounds 1 - 3: every site plain fp16", plain()),
         ("the product (round 4): token MLP on fp16 hi+lo operand pairs", S.Prec("bf16")),
         ("... + every token-side product on hi+lo pairs", with_dtype(tok_x="split16", tok_w="split16")),
         ("... + up-scaling operands on hi+lo pairs", with_dtype(tok_x="split16", tok_w="split16", up="split16")),
         ("all but tok", S.Prec("bf16", exact_sites=("tok",)))]
if "--sites" in sys.argv:            # one site at a time, and all but one (every site plain fp16 otherwise)
    def only(site_set, exact=()):
        pr = S.Prec("bf16", only_sites=site_set) if site_set else S.Prec("bf16", exact_sites=exact)
        pr.site_dtype.pop("tok.mlp", None)
        return pr
    cases = [(f"only {s_}", only((s_,))) for s_ in SITES] + [(f"all but {s_}", only(None, (s_,))) for s_ in SITES]
for name, prec in cases:
    d = (run(prec) - ref).abs()
    print(json.dumps({"case": name, "mean": round(float(d.mean()), 4), "p99": round(float(d.flatten().quantile(0.99)), 3), "max": round(float(d.max()), 2)}), flush=True)
