"""Calibrate the synthetic checkpoint (test-data tooling; uses the CPU oracle, never shipped in the product path).

Computes the spatial-mean direction ``ubar`` (32 floats) of the mask decoder's up-scaled feature map on two
synthetic tiles and a few point prompts and stores it in ``micro_sam_amd/data/synthetic_calib.json``;
``micro_sam_amd.synthetic.synthetic_state_dict`` projects it out of the hyper-network output layer so that
mask logits are zero-mean fields following the image content.

Usage: python tools/calibrate_synthetic.py [model_type] [seed] [gain]
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
    sd = synthetic_state_dict(model_type, seed, calibrated=False)
    g = torch.Generator().manual_seed(1234)
    acc = []
    with torch.no_grad():
        for tile_seed in (9000, 9001):
            img = A.to_image(synthetic_tile(tile_seed))
            x = S.preprocess(torch.as_tensor(img).permute(2, 0, 1)[None])
            f = S.image_encoder(sd, x, model_type=model_type)
            pts = torch.rand(8, 1, 2, generator=g) * 1024
            acc.append(upscaled_features(sd, f, pts).mean(dim=(0, 2, 3)))
    ubar = torch.stack(acc).mean(0)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "micro_sam_amd", "data",
                        "synthetic_calib.json")
    db = {}
    if os.path.exists(path):
        with open(path) as fh:
            db = json.load(fh)
    db[f"{model_type}/{seed}"] = {"ubar": [float(v) for v in ubar], "gain": gain,
                                  "cos_between_tiles": float(F.cosine_similarity(acc[0], acc[1], dim=0))}
    with open(path, "w") as fh:
        json.dump(db, fh, indent=1)
    print("wrote", path, db[f"{model_type}/{seed}"]["cos_between_tiles"])


if __name__ == "__main__":
    main()
