"""Calibrate the synthetic checkpoint (test-data tooling; uses the CPU oracle, never shipped in the product path).

Computes the spatial-mean direction ``ubar`` (32 floats) of the mask decoder's up-scaled feature map on two
synthetic tiles and a few point prompts and stores it in ``micro_sam_amd/data/synthetic_calib.json``;
``micro_sam_amd.synthetic.synthetic_state_dict`` projects it out of the hyper-network output layer so that
mask logits are zero-mean fields following the image content.

Usage: python tools/calibrate_synthetic.py [model_type] [seed] [gain] [variant: field|blobs]
"""
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from micro_sam_amd.synthetic import synthetic_state_dict, synthetic_tile  # noqa: E402
from oracle import amg_ref as A  # noqa: E402
from oracle import sam_ref as S  # noqa: E402


def upscaled_features(sd, f, pts):
    p = S.Prec("fp32")
    m = "mask_decoder."
    lbl = torch.ones(len(pts), 1, dtype=torch.int)
    sparse, dense = S.prompt_encoder(sd, (pts, lbl), None, None)
    out_tok = torch.cat([sd[m + "iou_token.weight"], sd[m + "mask_tokens.weight"]], dim=0)
    tokens = torch.cat((out_tok.unsqueeze(0).expand(sparse.size(0), -1, -1), sparse), dim=1)
    src = torch.repeat_interleave(f, tokens.shape[0], dim=0) + dense
    pos = torch.repeat_interleave(S.get_dense_pe(sd), tokens.shape[0], dim=0)
    _, src2 = S.two_way_transformer(sd, src, pos, tokens, p)
    src2 = src2.transpose(1, 2).view(tokens.shape[0], 256, 64, 64)
    up = F.conv_transpose2d(src2, sd[m + "output_upscaling.0.weight"], sd[m + "output_upscaling.0.bias"], stride=2)
    up = F.gelu(S.layer_norm_2d(up, sd[m + "output_upscaling.1.weight"], sd[m + "output_upscaling.1.bias"]))
    return F.gelu(F.conv_transpose2d(up, sd[m + "output_upscaling.3.weight"], sd[m + "output_upscaling.3.bias"],
                                     stride=2))


def main():
    model_type = sys.argv[1] if len(sys.argv) > 1 else "vit_b"
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    gain = float(sys.argv[3]) if len(sys.argv) > 3 else 8.0
    variant = sys.argv[4] if len(sys.argv) > 4 else "field"
    sd = synthetic_state_dict(model_type, seed, calibrated=False, variant=variant)
    g = torch.Generator().manual_seed(1234)
    far_acc, near_acc, ups, all_pts = [], [], [], []
    yy, xx = torch.meshgrid(torch.arange(256.0), torch.arange(256.0), indexing="ij")
    with torch.no_grad():
        for tile_seed in (9000, 9001):
            img = A.to_image(synthetic_tile(tile_seed))
            x = S.preprocess(torch.as_tensor(img).permute(2, 0, 1)[None])
            f = S.image_encoder(sd, x, model_type=model_type)
            pts = 150 + torch.rand(8, 1, 2, generator=g) * 724
            up = upscaled_features(sd, f, pts)                      # [8,32,256,256]
            ups.append(up); all_pts.append(pts)
            for i in range(8):
                d2 = (xx * 4 + 2 - pts[i, 0, 0]) ** 2 + (yy * 4 + 2 - pts[i, 0, 1]) ** 2
                near_acc.append(up[i][:, d2 < 24 ** 2].mean(dim=1))
                far_acc.append(up[i][:, d2 > 300 ** 2].mean(dim=1))
    ubar = torch.stack(far_acc).mean(0)
    u = ubar / ubar.norm()
    du = torch.stack(near_acc).mean(0) - ubar
    du = du - (du @ u) * u
    du_n = du / du.norm()
    # field statistics of the projected random hyper-network output (what the mask logits look like far away)
    sd_c = sd
    proj = torch.eye(32) - torch.outer(u, u)
    w2 = gain * proj @ sd_c["mask_decoder.output_hypernetworks_mlps.1.layers.2.weight"]
    hvec = w2 @ torch.randn(256, 64, generator=g) * 1.0             # typical hidden activations are O(1)
    sig = float(torch.einsum("ck,bchw->bkhw", hvec, ups[0][:2]).std())
    beta = 2.5 * sig / float(ubar.norm())
    gamma = 7.0 * sig / float(du.norm())
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "micro_sam_amd", "data",
                        "synthetic_calib.json")
    db = {}
    if os.path.exists(path):
        with open(path) as fh:
            db = json.load(fh)
    key = f"{model_type}/{seed}/{variant}"
    entry = {"ubar": [float(v) for v in ubar], "gain": gain, "field_sigma": sig, "ubar_norm": float(ubar.norm())}
    if variant == "blobs":
        entry.update({"du": [float(v) for v in du_n], "beta": beta, "gamma": gamma, "du_norm": float(du.norm())})
    db = {k: v for k, v in db.items() if k.count("/") == 2}
    db[key] = entry
    with open(path, "w") as fh:
        json.dump(db, fh, indent=1)
    print("wrote", path, key, {k: v for k, v in entry.items() if k not in ("ubar", "du")})


if __name__ == "__main__":
    main()
