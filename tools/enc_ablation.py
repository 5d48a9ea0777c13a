"""Encoder rounding-site ablation (CPU, oracle only - test-data tooling, nothing here is on the product path).

VERDICT r2 item 1: which 16-bit rounding sites of the image encoder cost the per-instance IoU against the fp32 reference?
For every configuration the oracle's image encoder runs with bf16 rounding at exactly the named sites (``Prec.enc_only`` /
``enc_blocks`` / ``enc_dtype``, oracle/sam_ref.py), the mask decoder stays fp32, and the masks of the candidates the fp32 reference
keeps (predicted IoU > 0.88, stability >= 0.95) are compared with the reference's: flipped pixels per kept 1024 x 1024 mask and the
share of kept masks with IoU >= 0.999.

    python tools/enc_ablation.py [tile ...]            (default tiles 1000 1001 1002; every 4th grid prompt = 256 prompts per tile)

Log: profiles/r03_enc_ablation.txt
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from micro_sam_amd.synthetic import synthetic_state_dict, synthetic_tile  # noqa: E402
from oracle import amg_ref as A  # noqa: E402
from oracle import sam_ref as S  # noqa: E402

LIN = ("qkv", "proj", "lin1", "lin2")
GLOBAL = (2, 5, 8, 11)


def embed(sd, img, prec):
    t = torch.as_tensor(S.apply_image(img)).permute(2, 0, 1).contiguous()[None]
    with torch.no_grad():
        return S.image_encoder(sd, S.preprocess(t), "vit_b", prec)


def decode(sd, f, sel_pts):
    lbl = torch.ones(len(sel_pts), 1, dtype=torch.int)
    outs = []
    with torch.no_grad():
        for s in range(0, len(sel_pts), 64):
            m, iou, low = S.predict_torch(sd, f, (1024, 1024), (1024, 1024), sel_pts[s:s + 64], lbl[s:s + 64], multimask_output=True,
                                          return_logits=True, precision="fp32")
            mm = m.flatten(0, 1)
            outs.append(((mm > 0), iou.flatten(), A.calculate_stability_score(mm, 0.0, 1.0)))
    return [torch.cat([o[i] for o in outs]) for i in range(3)]


def configs():
    def mk(only=None, blocks=None, dtype=None, default=torch.bfloat16):
        p = S.Prec("bf16")
        p.enc_dtype = {}              # start from the plain 16-bit policy (the default policy splits patch + neck)
        p.enc_only = None if only is None else set(only)
        p.enc_blocks = None if blocks is None else set(blocks)
        if dtype:
            p.enc_dtype.update(dtype)
        return p
    allx = [f"{l}.x" for l in LIN]
    allw = [f"{l}.w" for l in LIN]
    out = [("all sites bf16 (the product)", mk())]
    out.append(("all sites fp16", mk(dtype={s: torch.float16 for s in
                                            ["patch", "qkvstore", "relpos", "probs", "neck"] + allx + allw})))
    for s in ("patch", "qkvstore", "relpos", "probs", "neck"):
        out.append((f"only {s}", mk(only=[s])))
    for l in LIN:
        out.append((f"only {l} activations", mk(only=[l + ".x"])))
        out.append((f"only {l} weights", mk(only=[l + ".w"])))
    out.append(("only the 4 linears' activations", mk(only=allx)))
    out.append(("only the 4 linears' weights", mk(only=allw)))
    out.append(("only attention internals (qkvstore+relpos+probs)", mk(only=["qkvstore", "relpos", "probs"])))
    out.append(("only windowed blocks", mk(blocks=[b for b in range(12) if b not in GLOBAL])))
    out.append(("only global blocks", mk(blocks=GLOBAL)))
    for lo in (0, 3, 6, 9):
        out.append((f"only blocks {lo}-{lo + 2}", mk(blocks=range(lo, lo + 3))))
    out.append(("only patch + neck", mk(blocks=[-1, 12])))
    # candidate fixes: everything bf16 except ...
    sp = lambda names: {n: "split" for n in names}
    out.append(("fix: linears' activations hi+lo", mk(dtype=sp(allx))))
    out.append(("fix: linears' weights hi+lo", mk(dtype=sp(allw))))
    out.append(("fix: linears' act + weights hi+lo", mk(dtype=sp(allx + allw))))
    out.append(("fix: linears hi+lo, attention internals fp16", mk(dtype={**sp(allx + allw), "qkvstore": torch.float16,
                                                                          "relpos": torch.float16, "probs": torch.float16})))
    out.append(("fix: lin1+lin2 hi+lo (act + weights)", mk(dtype=sp(["lin1.x", "lin1.w", "lin2.x", "lin2.w"]))))
    out.append(("fix: qkv+proj hi+lo (act + weights)", mk(dtype=sp(["qkv.x", "qkv.w", "proj.x", "proj.w"]))))
    out.append(("fix: neck + patch hi+lo", mk(dtype=sp(["neck", "patch"]))))
    if os.environ.get("ABL_ONLY"):                     # e.g. ABL_ONLY="fix: neck" python tools/enc_ablation.py : a subset by name prefix
        out = [c for c in out if any(c[0].startswith(pre) for pre in os.environ["ABL_ONLY"].split("|"))]
    return out


def main():
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    tiles = [int(a) for a in sys.argv[1:]] or [1000, 1001, 1002]
    sd = synthetic_state_dict("vit_b", 0, variant="cells")
    g = A.build_all_layer_point_grids(32, 0, 1)[0] * 1024
    sel = np.arange(0, 1024, 4)
    pts = torch.as_tensor(g[sel], dtype=torch.float)[:, None, :]
    log = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "r03_enc_ablation.txt"), "a")

    def say(s):
        print(s, flush=True)
        log.write(s + "\n"); log.flush()
    say(f"# encoder rounding-site ablation, tiles {tiles}, 256 prompts per tile, fp32 decoder, cells checkpoint seed 0")
    refs = []
    for t in tiles:
        img = A.to_image(synthetic_tile(t))
        f = embed(sd, img, "fp32")
        m, iou, stab = decode(sd, f, pts)
        keep = (iou > 0.88) & (stab >= 0.95)
        refs.append((img, f, m, keep))
        say(f"# tile {t}: kept candidates {int(keep.sum())}")
    say(f"{'configuration':52s} {'emb mean|d|':>11s} {'flips/mask':>10s} {'IoU>=.999':>9s} {'min IoU':>8s}")
    for name, prec in configs():
        t0 = time.time()
        errs, flips, ious = [], [], []
        for img, f, mr, keep in refs:
            fe = embed(sd, img, prec)
            errs.append(float((fe - f).abs().mean()))
            m, _, _ = decode(sd, fe, pts)
            inter = (mr & m).flatten(1).sum(1).float(); uni = (mr | m).flatten(1).sum(1).float()
            ious.append(torch.where(uni > 0, inter / uni, torch.ones_like(uni))[keep])
            flips.append((mr ^ m).flatten(1).sum(1)[keep].float())
        i = torch.cat(ious); fl = torch.cat(flips)
        say(f"{name:52s} {np.mean(errs):11.5f} {float(fl.mean()):10.2f} {float((i >= 0.999).float().mean()):9.3f} {float(i.min()):8.4f}"
            f"   ({time.time() - t0:.0f}s)")


if __name__ == "__main__":
    main()
