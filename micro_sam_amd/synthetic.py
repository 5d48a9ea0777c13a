"""Seeded synthetic weights and microscopy-like tiles.

No SAM checkpoint can be downloaded in the build / GPU environments, so tests, ``smoke()`` and ``bench.py``
run on a deterministic random-init ``state_dict`` that uses the upstream parameter names (the names
micro_sam relies on: ``micro_sam/models/build_sam.py:26``, ``micro_sam/util.py:578-598``, SURVEY.md
Appendix C) and on synthetic tiles shaped like the reference's own synthetic fixtures
(``micro_sam/sample_data.py:342-357``).

The init scales are chosen so that every code path carries signal: attention logits have O(1) spread
(softmax neither uniform nor one-hot), relative-position tables are non-zero, mask logits are large
enough that stability scores spread over (0, 1] and predicted IoUs straddle the default thresholds.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Tuple

import numpy as np
import torch

VIT_CONFIGS = {
    # micro_sam/models/build_sam.py:40-84
    "vit_b": dict(embed_dim=768, depth=12, num_heads=12, global_attn_indexes=(2, 5, 8, 11)),
    "vit_l": dict(embed_dim=1024, depth=24, num_heads=16, global_attn_indexes=(5, 11, 17, 23)),
    "vit_h": dict(embed_dim=1280, depth=32, num_heads=16, global_attn_indexes=(7, 15, 23, 31)),
}


def _load_calibration() -> Dict[str, dict]:
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "synthetic_calib.json")
    if not os.path.exists(path):
        return {}
    with open(path) as f:
        return json.load(f)


def synthetic_state_dict(model_type: str = "vit_b", seed: int = 0, calibrated: bool = True,
                         variant: str = "field") -> "OrderedDict[str, torch.Tensor]":
    """fp32 CPU state_dict with upstream SAM key names for ``model_type`` in {vit_b, vit_l, vit_h}.

    ``variant``:
      "field"  mask logits are zero-mean fields that follow the image content (masks span the image; numerically
               benign: used by the parity tests);
      "blobs"  additionally gives the image->token attention a prompt-locality kernel and the hyper-network a
               negative far-field offset, so every grid prompt yields a compact blob near its point: realistic AMG
               workload (distinct boxes, non-trivial NMS, hundreds of instances, short RLEs) - used by bench.py.
    ``calibrated``: apply the hyper-network calibration stored in ``data/synthetic_calib.json`` (written by
    ``tools/calibrate_synthetic.py``) when one exists for (model_type, seed, variant)."""
    assert variant in ("field", "blobs"), variant
    cfg = VIT_CONFIGS[model_type[:5]]
    D, depth, heads = cfg["embed_dim"], cfg["depth"], cfg["num_heads"]
    hd = D // heads
    g = torch.Generator().manual_seed(seed)
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    calib = _load_calibration().get(f"{model_type[:5]}/{seed}/{variant}") if calibrated else None

    def n(*shape, std=0.02, mean=0.0):
        return torch.randn(*shape, generator=g) * std + mean

    e = "image_encoder."
    sd[e + "pos_embed"] = n(1, 64, 64, D, std=0.02)
    sd[e + "patch_embed.proj.weight"] = n(D, 3, 16, 16, std=0.03)
    sd[e + "patch_embed.proj.bias"] = n(D, std=0.02)
    for i in range(depth):
        b = f"{e}blocks.{i}."
        S = 64 if i in cfg["global_attn_indexes"] else 14
        sd[b + "norm1.weight"] = n(D, std=0.05, mean=1.0)
        sd[b + "norm1.bias"] = n(D, std=0.02)
        sd[b + "attn.rel_pos_h"] = n(2 * S - 1, hd, std=0.08)
        sd[b + "attn.rel_pos_w"] = n(2 * S - 1, hd, std=0.08)
        sd[b + "attn.qkv.weight"] = n(3 * D, D, std=1.3 / D ** 0.5)
        sd[b + "attn.qkv.bias"] = n(3 * D, std=0.05)
        sd[b + "attn.proj.weight"] = n(D, D, std=0.6 / D ** 0.5)
        sd[b + "attn.proj.bias"] = n(D, std=0.02)
        sd[b + "norm2.weight"] = n(D, std=0.05, mean=1.0)
        sd[b + "norm2.bias"] = n(D, std=0.02)
        sd[b + "mlp.lin1.weight"] = n(4 * D, D, std=1.0 / D ** 0.5)
        sd[b + "mlp.lin1.bias"] = n(4 * D, std=0.02)
        sd[b + "mlp.lin2.weight"] = n(D, 4 * D, std=0.6 / (4 * D) ** 0.5)
        sd[b + "mlp.lin2.bias"] = n(D, std=0.02)
    sd[e + "neck.0.weight"] = n(256, D, 1, 1, std=1.0 / D ** 0.5)
    sd[e + "neck.1.weight"] = n(256, std=0.05, mean=1.0)
    sd[e + "neck.1.bias"] = n(256, std=0.02)
    sd[e + "neck.2.weight"] = n(256, 256, 3, 3, std=1.0 / (9 * 256) ** 0.5)
    sd[e + "neck.3.weight"] = n(256, std=0.05, mean=1.0)
    sd[e + "neck.3.bias"] = n(256, std=0.02)

    p = "prompt_encoder."
    sd[p + "pe_layer.positional_encoding_gaussian_matrix"] = n(2, 128, std=1.0)
    for i in range(4):
        sd[p + f"point_embeddings.{i}.weight"] = n(1, 256, std=0.5)
    sd[p + "not_a_point_embed.weight"] = n(1, 256, std=0.5)
    sd[p + "mask_downscaling.0.weight"] = n(4, 1, 2, 2, std=0.5)
    sd[p + "mask_downscaling.0.bias"] = n(4, std=0.1)
    sd[p + "mask_downscaling.1.weight"] = n(4, std=0.05, mean=1.0)
    sd[p + "mask_downscaling.1.bias"] = n(4, std=0.02)
    sd[p + "mask_downscaling.3.weight"] = n(16, 4, 2, 2, std=0.25)
    sd[p + "mask_downscaling.3.bias"] = n(16, std=0.1)
    sd[p + "mask_downscaling.4.weight"] = n(16, std=0.05, mean=1.0)
    sd[p + "mask_downscaling.4.bias"] = n(16, std=0.02)
    sd[p + "mask_downscaling.6.weight"] = n(256, 16, 1, 1, std=0.25)
    sd[p + "mask_downscaling.6.bias"] = n(256, std=0.05)
    sd[p + "no_mask_embed.weight"] = n(1, 256, std=0.3)

    m = "mask_decoder."

    def attn(prefix, internal):
        for nm in ("q_proj", "k_proj", "v_proj"):
            sd[prefix + nm + ".weight"] = n(internal, 256, std=1.2 / 16)
            sd[prefix + nm + ".bias"] = n(internal, std=0.05)
        sd[prefix + "out_proj.weight"] = n(256, internal, std=0.8 / internal ** 0.5)
        sd[prefix + "out_proj.bias"] = n(256, std=0.02)

    for i in range(2):
        lp = f"{m}transformer.layers.{i}."
        attn(lp + "self_attn.", 256)
        sd[lp + "norm1.weight"] = n(256, std=0.05, mean=1.0); sd[lp + "norm1.bias"] = n(256, std=0.02)
        attn(lp + "cross_attn_token_to_image.", 128)
        sd[lp + "norm2.weight"] = n(256, std=0.05, mean=1.0); sd[lp + "norm2.bias"] = n(256, std=0.02)
        sd[lp + "mlp.lin1.weight"] = n(2048, 256, std=1.0 / 16); sd[lp + "mlp.lin1.bias"] = n(2048, std=0.02)
        sd[lp + "mlp.lin2.weight"] = n(256, 2048, std=0.7 / 2048 ** 0.5); sd[lp + "mlp.lin2.bias"] = n(256, std=0.02)
        sd[lp + "norm3.weight"] = n(256, std=0.05, mean=1.0); sd[lp + "norm3.bias"] = n(256, std=0.02)
        sd[lp + "norm4.weight"] = n(256, std=0.05, mean=1.0); sd[lp + "norm4.bias"] = n(256, std=0.02)
        attn(lp + "cross_attn_image_to_token.", 128)
        # prompt locality: aligned q / k projections (same orthonormal 128x256 map U, no bias) make the image->token
        # attention logit contain pos(x)^T U^T U pe(point) - the random-Fourier kernel, peaked within ~80 px of
        # the prompt - so image tokens near the prompt pick up the point token's value (a local "marker")
        u_mat = torch.linalg.qr(n(256, 128, std=1.0))[0].t().contiguous()          # [128,256], orthonormal rows
        if variant == "blobs":
            ip = lp + "cross_attn_image_to_token."
            sd[ip + "q_proj.weight"] = 1.7 * u_mat
            sd[ip + "k_proj.weight"] = 1.7 * u_mat.clone()
            sd[ip + "q_proj.bias"] = torch.zeros(128)
            sd[ip + "k_proj.bias"] = torch.zeros(128)
            sd[ip + "out_proj.weight"] = sd[ip + "out_proj.weight"] * 12.0
    attn(m + "transformer.final_attn_token_to_image.", 128)
    sd[m + "transformer.norm_final_attn.weight"] = n(256, std=0.05, mean=1.0)
    sd[m + "transformer.norm_final_attn.bias"] = n(256, std=0.02)
    sd[m + "iou_token.weight"] = n(1, 256, std=0.5)
    sd[m + "mask_tokens.weight"] = n(4, 256, std=0.5)
    # nearly sub-pixel-consistent transposed convs (+-3 %: still distinct per sub-pixel, so indexing bugs show):
    # masks are smooth at the 4x4 sub-pixel level instead of pixel noise -> saner run-length counts
    sd[m + "output_upscaling.0.weight"] = n(256, 64, 1, 1, std=1.0 / 16) * n(256, 64, 2, 2, std=0.03, mean=1.0)
    sd[m + "output_upscaling.0.bias"] = n(64, std=0.02)
    sd[m + "output_upscaling.1.weight"] = n(64, std=0.05, mean=1.0)
    sd[m + "output_upscaling.1.bias"] = n(64, std=0.02)
    sd[m + "output_upscaling.3.weight"] = n(64, 32, 1, 1, std=1.0 / 8) * n(64, 32, 2, 2, std=0.03, mean=1.0)
    sd[m + "output_upscaling.3.bias"] = n(32, std=0.02)
    for i in range(4):
        hp = f"{m}output_hypernetworks_mlps.{i}."
        sd[hp + "layers.0.weight"] = n(256, 256, std=1.0 / 16); sd[hp + "layers.0.bias"] = n(256, std=0.02)
        sd[hp + "layers.1.weight"] = n(256, 256, std=1.4 / 16); sd[hp + "layers.1.bias"] = n(256, std=0.02)
        # large output scale -> |logit| >> 1 away from the mask boundary (sharp masks, high stability)
        w2, b2 = n(32, 256, std=6.0 / 16), n(32, std=0.5)
        if calib is not None:
            # project out the calibrated spatial-mean direction of the up-scaled features: mask logits become
            # zero-mean fields that follow the image content (cells) instead of all-or-nothing masks
            u = torch.tensor(calib["ubar"], dtype=torch.float32)
            u = u / u.norm()
            proj = torch.eye(32) - torch.outer(u, u)
            w2, b2 = float(calib["gain"]) * proj @ w2, float(calib["gain"]) * proj @ b2
            if "du" in calib:
                # negative far-field offset (along ubar) + positive response to the prompt marker (along du)
                b2 = b2 - float(calib["beta"]) * u + float(calib["gamma"]) * torch.tensor(calib["du"], dtype=torch.float32)
        sd[hp + "layers.2.weight"] = w2; sd[hp + "layers.2.bias"] = b2
    ip = m + "iou_prediction_head."
    sd[ip + "layers.0.weight"] = n(256, 256, std=1.0 / 16); sd[ip + "layers.0.bias"] = n(256, std=0.02)
    sd[ip + "layers.1.weight"] = n(256, 256, std=1.4 / 16); sd[ip + "layers.1.bias"] = n(256, std=0.02)
    sd[ip + "layers.2.weight"] = n(4, 256, std=0.08 / 16)
    sd[ip + "layers.2.bias"] = torch.full((4,), 0.9)
    return sd


def synthetic_tile(seed: int, shape: Tuple[int, int] = (1024, 1024)) -> np.ndarray:
    """uint8 [H,W] cell-like tile: noisy background + blurred random ellipses."""
    from scipy.ndimage import gaussian_filter
    rng = np.random.default_rng(seed)
    h, w = shape
    img = np.clip(rng.normal(40.0, 10.0, size=shape), 0, 255).astype(np.float32)
    n_obj = int(rng.integers(40, 121) * (h * w) / (1024 * 1024)) or 1
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    for _ in range(n_obj):
        cy, cx = rng.uniform(0, h), rng.uniform(0, w)
        a, b = rng.uniform(8, 40, size=2)
        th = rng.uniform(0, np.pi)
        val = rng.uniform(120, 255)
        y0, y1 = int(max(0, cy - 42)), int(min(h, cy + 43))
        x0, x1 = int(max(0, cx - 42)), int(min(w, cx + 43))
        dy, dx = yy[y0:y1, x0:x1] - cy, xx[y0:y1, x0:x1] - cx
        u = dx * np.cos(th) + dy * np.sin(th)
        v = -dx * np.sin(th) + dy * np.cos(th)
        inside = (u / a) ** 2 + (v / b) ** 2 < 1.0
        img[y0:y1, x0:x1][inside] = val
    img = gaussian_filter(img, 1.5)
    img = img + rng.normal(0.0, 6.0, size=shape).astype(np.float32)
    return np.clip(img, 0, 255).astype(np.uint8)


def three_disk_fixture(size: int = 256) -> Tuple[np.ndarray, np.ndarray]:
    """The reference's test fixture (test/test_instance_segmentation.py:20-39): three filled disks of radius
    29/33/35 (scaled with ``size``/256) at 1/4, 1/2, 3/4 of the canvas; image = mask * 255."""
    s = size / 256.0
    yy, xx = np.mgrid[0:size, 0:size]
    mask = np.zeros((size, size), dtype=np.uint8)
    for k, (c, r) in enumerate(((size // 4, 29 * s), (size // 2, 33 * s), (3 * size // 4, 35 * s)), start=1):
        mask[(yy - c) ** 2 + (xx - c) ** 2 < r * r] = k
    image = (mask > 0).astype(np.uint8) * 255
    return mask, image
