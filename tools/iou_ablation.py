"""CPU ablation behind DESIGN.md section 4 (test-data tooling; oracle only, nothing here is on the product path):

    python tools/iou_ablation.py sites          per decoder rounding site: flipped pixels / IoU of the kept candidates
    python tools/iou_ablation.py paths          oracle emulation of (encoder, decoder) precision combinations on the full
                                                32 x 32 grid of tile 1000 vs the fp32 reference (per-instance IoU report)
    python tools/iou_ablation.py encfp16        what-if: fp16 instead of bf16 MFMA operands in the image encoder

Embeddings / states are cached under /tmp/msam_ablation.
"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from micro_sam_amd.synthetic import synthetic_state_dict, synthetic_tile  # noqa: E402
from oracle import amg_ref as A  # noqa: E402
from oracle import parity as PT  # noqa: E402
from oracle import pipeline_ref as PR  # noqa: E402
from oracle import sam_ref as S  # noqa: E402

CACHE = "/tmp/msam_ablation"


def embedding(sd, img, prec):
    os.makedirs(CACHE, exist_ok=True)
    path = os.path.join(CACHE, f"emb_{prec}.pt")
    if not os.path.exists(path):
        f, _, _ = PR.compute_embeddings(sd, [img], "vit_b", prec)
        torch.save(f, path)
    return torch.load(path)


def sites(sd, img):
    f = embedding(sd, img, "fp32")
    g = A.build_all_layer_point_grids(32, 0, 1)[0] * 1024
    sel = np.arange(0, 1024, 4)
    pts = torch.as_tensor(g[sel], dtype=torch.float)[:, None, :]
    lbl = torch.ones(len(sel), 1, dtype=torch.int)

    def run(prec):
        outs = []
        with torch.no_grad():
            for s in range(0, len(sel), 64):
                m, iou, low = S.predict_torch(sd, f, (1024, 1024), (1024, 1024), pts[s:s + 64], lbl[s:s + 64], multimask_output=True,
                                              return_logits=True, precision=prec)
                mm = m.flatten(0, 1)
                outs.append(((mm > 0), iou.flatten(), A.calculate_stability_score(mm, 0.0, 1.0), low.flatten(0, 1)))
        return [torch.cat([o[i] for o in outs]) for i in range(4)]
    mr, ir, sr, lr = run("fp32")
    keep = (ir > 0.88) & (sr >= 0.95)
    print("kept candidates", int(keep.sum()))

    def cmp(name, prec):
        m, iou, stab, low = run(prec)
        inter = (mr & m).flatten(1).sum(1).float(); uni = (mr | m).flatten(1).sum(1).float()
        i = torch.where(uni > 0, inter / uni, torch.ones_like(uni))[keep]
        flips = (mr ^ m).flatten(1).sum(1)[keep]
        print(f"{name:34s} IoU>=.999 {float((i >= 0.999).float().mean()):.3f} min {float(i.min()):.4f} flips/mask "
              f"{float(flips.float().mean()):.2f} | logit mean|d| {float((low - lr).abs().mean()):.4f} | iou_pred max|d| "
              f"{float((iou - ir).abs().max()):.5f}", flush=True)

    def mk(dec=torch.bfloat16, **kw):
        p = S.Prec("bf16"); p.dec_dtype = dec; p.site_dtype.update(kw); return p
    cmp("all bf16", mk())
    for site in ("stream", "tok", "t2i0", "fold", "table", "probs", "foldv", "up", "head"):
        p = S.Prec("bf16", only_sites=[site]); p.dec_dtype = torch.bfloat16
        cmp("only " + site + " (bf16)", p)
    h = torch.float16
    cmp("stream fp16", mk(stream=h))
    cmp("stream+fold+probs+up+table fp16", mk(stream=h, fold=h, probs=h, up=h, table=h, foldv=h))
    cmp("+ tok / head as bf16 hi+lo", mk(stream=h, fold=h, probs=h, up=h, table=h, foldv=h, tok="split", head="split"))
    cmp("whole decoder fp16 (the default build)", mk(dec=h))


def paths(sd, img):
    def state(enc, dec):
        path = os.path.join(CACHE, f"state_{enc}_{dec}.pt")
        if not os.path.exists(path):
            S.DECODER_DTYPE = torch.float16 if dec == "fp16" else torch.bfloat16
            st = PR.amg_initialize(sd, img, embedding(sd, img, enc), (1024, 1024), (1024, 1024),
                                   precision="fp32" if dec == "fp32" else "bf16")
            torch.save(st, path)
        return torch.load(path, weights_only=False)
    ref = state("fp32", "fp32")
    kr = PT.kept_candidates(ref)
    seg_ref = PR.amg_generate(ref)
    for enc, dec in (("bf16", "bf16"), ("bf16", "fp16"), ("bf16", "fp32"), ("fp32", "bf16"), ("fp32", "fp16")):
        t0 = time.time()
        st = state(enc, dec)
        rep = PT.public(PT.iou_report(kr, PT.kept_candidates(st), PT.oracle_mask_fn(ref), PT.oracle_mask_fn(st)))
        rep.pop("worst")
        print(f"encoder {enc} decoder {dec}: {json.dumps(rep)} {PT.label_agreement(seg_ref, PR.amg_generate(st))} ({time.time() - t0:.0f}s)",
              flush=True)


def encfp16(sd, img):
    """What-if: the image encoder's MFMA operands in fp16 instead of bf16 (same MFMA rate, 11 instead of 8 significand bits), decoder
    fp16 as in the default build - the oracle's emulation on the full 32 x 32 grid of tile 1000 against the committed fp32 golden."""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "cells_vit_b_tile1000.npz"))
    kept_ref = g["kept"].astype(np.int64)
    off, cnt = g["rle_offsets"], g["rle_counts"]
    pos = {int(c): j for j, c in enumerate(kept_ref)}
    ref_mask = lambda i: A.rle_to_mask({"size": [1024, 1024], "counts": cnt[off[pos[i]]:off[pos[i] + 1]]})
    for name, edt in (("fp16", torch.float16), ("bf16", torch.bfloat16)):
        t0 = time.time()
        S.ENCODER_DTYPE = edt
        S.DECODER_DTYPE = torch.float16
        f, _, _ = PR.compute_embeddings(sd, [img], "vit_b", "bf16")
        emb_err = float((f - embedding(sd, img, "fp32")).abs().mean())
        st = PR.amg_initialize(sd, img, f, (1024, 1024), (1024, 1024), precision="bf16")
        rep = PT.public(PT.iou_report(kept_ref, PT.kept_candidates(st), ref_mask, PT.oracle_mask_fn(st)))
        rep.pop("worst")
        print(f"encoder operands {name}, decoder fp16 (oracle emulation): embedding mean |d| {emb_err:.5f}  {json.dumps(rep)} "
              f"({time.time() - t0:.0f}s)", flush=True)
    S.ENCODER_DTYPE = torch.bfloat16


if __name__ == "__main__":
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    sd_ = synthetic_state_dict("vit_b", 0, variant="cells")
    img_ = A.to_image(synthetic_tile(1000))
    {"sites": sites, "paths": paths, "encfp16": encfp16}[sys.argv[1] if len(sys.argv) > 1 else "sites"](sd_, img_)
