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
    assert variant in ("field", "blobs", "cells"), variant
    if model_type[:5] == "vit_t":
        return _synthetic_vit_t(seed, variant)
    cfg = VIT_CONFIGS[model_type[:5]]
    D, depth, heads = cfg["embed_dim"], cfg["depth"], cfg["num_heads"]
    hd = D // heads
    g = torch.Generator().manual_seed(seed)
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    calib = _load_calibration().get(f"{model_type[:5]}/{seed}/{variant}") if (calibrated and variant != "cells") else None

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
    if variant == "cells":
        cal = _load_calibration().get(f"{model_type[:5]}/{seed}/cells") if calibrated else None
        _design_cells(sd, D, depth, g, iou_offset=None if cal is None else cal["iou_offset"])
    return sd


def _synthetic_vit_t(seed: int, variant: str):
    """MobileSAM-shaped checkpoint: prompt encoder / mask decoder of the vit_b checkpoint of the same seed ("field" or "blobs"; the
    designed "cells" variant is tied to the ViT encoder), TinyViT-5M encoder with seeded random weights (BatchNorm running statistics
    away from (0, 1) and non-zero attention offset biases, so that those paths are exercised) + its unused classification head."""
    from .models.tiny_vit import TinyViT
    base = synthetic_state_dict("vit_b", seed, variant="field" if variant == "cells" else variant)
    sd = OrderedDict((k, v) for k, v in base.items() if not k.startswith("image_encoder."))
    g = torch.Generator().manual_seed(seed + 7919)
    with torch.no_grad():
        enc = TinyViT()
        for name, t in enc.state_dict().items():
            if name.endswith("num_batches_tracked"):
                v = t.clone()
            elif name.endswith("running_var"):
                v = torch.rand(t.shape, generator=g) * 0.5 + 0.75
            elif name.endswith("running_mean"):
                v = torch.randn(t.shape, generator=g) * 0.1
            elif name.endswith("attention_biases"):
                v = torch.randn(t.shape, generator=g) * 0.5
            elif t.dim() == 1 and ("bn.weight" in name or "norm" in name or name.endswith(".1.weight") or name.endswith(".3.weight")) \
                    and name.endswith("weight"):
                v = torch.randn(t.shape, generator=g) * 0.05 + 1.0
            elif t.dim() == 1:
                v = torch.randn(t.shape, generator=g) * 0.02
            else:
                fan_in = t[0].numel()
                v = torch.randn(t.shape, generator=g) * (1.0 / fan_in ** 0.5)
            sd["image_encoder." + name] = v
    return sd


def scale_mask_logits(sd, scale: float):
    """Multiply every mask logit of the checkpoint by ``scale`` (in place, returns ``sd``): the last layer of the four
    hyper-network MLPs is linear in the logits.  ``scale`` = 0.25 gives the softer mask boundaries (|logit| ~ 3 - 12 instead of
    10 - 50) of the parity sensitivity runs (tests/test_gpu_parity_iou.py)."""
    for i in range(4):
        hp = f"mask_decoder.output_hypernetworks_mlps.{i}.layers.2."
        sd[hp + "weight"] = sd[hp + "weight"] * scale
        sd[hp + "bias"] = sd[hp + "bias"] * scale
    return sd


def _synthetic_tile(seed: int, shape: Tuple[int, int], with_labels: bool):
    from scipy.ndimage import gaussian_filter
    rng = np.random.default_rng(seed)
    h, w = shape
    img = np.clip(rng.normal(40.0, 10.0, size=shape), 0, 255).astype(np.float32)
    labels = np.zeros(shape, dtype=np.int32) if with_labels else None
    n_obj = int(rng.integers(40, 121) * (h * w) / (1024 * 1024)) or 1
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    for k in range(n_obj):
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
        if with_labels:
            labels[y0:y1, x0:x1][inside] = k + 1                 # later ellipses cover earlier ones, as in the image
    img = gaussian_filter(img, 1.5)
    img = img + rng.normal(0.0, 6.0, size=shape).astype(np.float32)
    return np.clip(img, 0, 255).astype(np.uint8), labels


def synthetic_tile(seed: int, shape: Tuple[int, int] = (1024, 1024)) -> np.ndarray:
    """uint8 [H,W] cell-like tile: noisy background + blurred random ellipses."""
    return _synthetic_tile(seed, shape, False)[0]


def synthetic_tile_with_labels(seed: int, shape: Tuple[int, int] = (1024, 1024)) -> Tuple[np.ndarray, np.ndarray]:
    """``synthetic_tile(seed, shape)`` (the same pixels) and the instance label image of its ellipses (int32, 0 = background; an ellipse
    covered completely by later ones has no pixel left): training pairs for the fine-tuning path (tools/trained_parity.py)."""
    return _synthetic_tile(seed, shape, True)


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


# ------------------------------------------------------------------------------------------------ variant "cells"
# A checkpoint with upstream names whose masks behave like a trained SAM's: compact regions around the prompt whose
# boundaries follow IMAGE EDGES (steep logit slopes, |logit| of order 10-50), near-duplicate masks for neighbouring
# prompts and for the three multimask outputs (so the box NMS has real work), a few hundred instances per tile.  No
# checkpoint can be downloaded here; per-instance IoU between two arithmetic paths (fp32 CPU reference vs bf16 MFMA
# kernels) is only meaningful on masks of that kind.  Everything not named below stays the seeded random init, and the
# random channels still reach the logits through a random "texture" term of ~1.5 logits, so an error in any kernel
# still moves mask boundaries.
#
# Mechanism (all of it ordinary SAM parameters):
#  * encoder: 16 residual channels carry the 4x4-sub-block means of each 16x16 patch (patch-embedding rows), three carry
#    intensity thresholds, eight carry the +-1 bits of the token's 64-px lattice block (pos_embed); the blocks do not write
#    to them (zero rows in proj / lin2); the neck turns them into embedding channels  sub-block mean - threshold_t  and
#    the block bits, next to an all-zero reference channel EREF.  Every LayerNorm on the way subtracts a per-token mean
#    and divides by a per-token deviation: the NEXT linear layer always takes (channel - reference channel), so zero
#    crossings stay exactly at  sub-block mean == threshold;
#  * prompt: eight designed PE frequencies are square-wave-like in the prompt position; layer 0's token MLP saturates
#    them to the +-1 bits of the prompt's block; head 0 of layer 0's self attention lets every token keep its own id /
#    bit channels (each token attends to itself);
#  * locality: heads 0 / 1 of layer 0's image->token attention compare the x / y block bits of image token and prompt:
#    on a match the image token attends to the point token, else to the padding token, whose value writes -V into a
#    marker channel (0 inside the prompt's 64-px block, -V outside);
#  * up-scaling: +x / -x channel pairs (GELU(x) - GELU(-x) = x) carry the sub-block channels to the 4x4 sub-pixels of
#    every token linearly; the hyper-network output (last-layer bias + small random weights) combines the intensity
#    term of the mask token's threshold, the two markers and the random texture.
CELLS = dict(
    EI0=32, EREF=80, EM0=81, C0=84, CONE=92, TREF=94, ID0=96, PAD_K=102, PAD_V=103, P0=104, B0=112,   # embedding / token channels
    thr_grey=(80.0, 92.0, 104.0),   # intensity thresholds of the three multimask outputs (background N(40,10), cells 120..255)
    id_amp=4.0, unit=8.0, kappa_min=0.4, bit_token=3.0,
)


def _block_bits(coord_px: torch.Tensor) -> torch.Tensor:
    """+-1 bits (4 per axis) of the 64-px lattice block of pixel coordinate(s): sign of sin(2 pi x / P), P = 128..1024."""
    import math
    return torch.stack([torch.sign(torch.sin(2.0 * math.pi * coord_px / P)) for P in (128.0, 256.0, 512.0, 1024.0)], dim=-1)


def _design_cells(sd, D: int, depth: int, g: torch.Generator, iou_offset=None) -> None:
    import math
    c = CELLS
    EI0, EREF, EM0, C0, ID0, PAD_K, PAD_V, P0, B0 = (c[k] for k in ("EI0", "EREF", "EM0", "C0", "ID0", "PAD_K", "PAD_V", "P0", "B0"))
    CONE, TREF = c["CONE"], c["TREF"]
    ei = list(range(EI0, EI0 + 48))
    em = [EM0, EM0 + 1]
    cch = list(range(C0, C0 + 8)) + [CONE]
    ids = list(range(ID0, ID0 + 8))
    pch = list(range(P0, P0 + 8))
    bch = list(range(B0, B0 + 8))
    img_designed = ei + [EREF] + em + cch            # image-side designed channels
    tok_designed = ids + pch + bch + [TREF]          # token-side designed channels (TREF: zero before every LayerNorm)
    all_designed = img_designed + tok_designed

    def n(*shape, std=0.02):
        return torch.randn(*shape, generator=g) * std

    # ---------------- encoder: residual channels 0..15 sub-block means, 16..18 thresholds, 19..26 block bits
    NR = 28
    e = "image_encoder."
    w = sd[e + "patch_embed.proj.weight"]
    w[:NR] = 0.0
    for sy in range(4):
        for sx in range(4):
            w[sy * 4 + sx, :, 4 * sy:4 * sy + 4, 4 * sx:4 * sx + 4] = 1.0 / 48.0
    mean = torch.tensor([123.675, 116.28, 103.53]); std = torch.tensor([58.395, 57.12, 57.375])
    sd[e + "patch_embed.proj.bias"][:NR] = 0.0
    for t, thr in enumerate(c["thr_grey"]):
        sd[e + "patch_embed.proj.bias"][16 + t] = float(((thr - mean) / std).mean())   # grey replicated to RGB, normalised
    pos = sd[e + "pos_embed"]                                          # [1, 64, 64, D]
    pos[..., :NR] = 0.0
    centre = torch.arange(64, dtype=torch.float32) * 16.0 + 8.0
    bits = _block_bits(centre)                                        # [64, 4]
    pos[0, :, :, 19:23] = bits[None, :, :]                            # x bits: vary along the column index
    pos[0, :, :, 23:27] = bits[:, None, :]                            # y bits: vary along the row index
    pos[0, :, :, 27] = 1.0                                            # constant code (scales like the bits)
    for i in range(depth):
        b = f"{e}blocks.{i}."
        for nm in ("attn.proj", "mlp.lin2"):
            sd[b + nm + ".weight"][:NR] = 0.0
            sd[b + nm + ".bias"][:NR] = 0.0
    w0 = sd[e + "neck.0.weight"]                                      # [256, D, 1, 1]
    w0[all_designed] = 0.0
    for t in range(3):
        for s in range(16):
            w0[EI0 + 16 * t + s, s, 0, 0] = 1.0
            w0[EI0 + 16 * t + s, 16 + t, 0, 0] = -1.0
    for k in range(8):
        w0[C0 + k, 19 + k, 0, 0] = 1.0
    w0[CONE, 27, 0, 0] = 1.0
    for nm in ("neck.1", "neck.3"):
        sd[e + nm + ".weight"][img_designed] = 1.0
        sd[e + nm + ".bias"][img_designed] = 0.0
    w2 = sd[e + "neck.2.weight"]                                      # [256, 256, 3, 3]
    w2[all_designed] = 0.0
    for ch in ei + cch:
        w2[ch, ch, 1, 1] = 1.0
        w2[ch, EREF, 1, 1] = -1.0
    sd[e + "neck.3.weight"][tok_designed] = 0.0
    sd[e + "neck.3.bias"][tok_designed] = 0.0

    # ---------------- prompt encoder: eight square-wave-like frequencies, clean PE on the channels the design reads
    p = "prompt_encoder."
    G = sd[p + "pe_layer.positional_encoding_gaussian_matrix"]        # [2, 128]; PE = sin / cos(2 pi (2u - 1) G)
    G[:, [EREF, TREF] + cch + ids + bch] = 0.0                              # sin channel == 0 there
    for k, P in enumerate((128.0, 256.0, 512.0, 1024.0)):
        f = 512.0 / P if P < 1024.0 else -512.0 / P                    # PE phase is -2 pi f: flips the sign for P = 1024
        G[:, P0 + k] = torch.tensor([f, 0.0])
        G[:, P0 + 4 + k] = torch.tensor([0.0, f])
    for i in range(4):
        sd[p + f"point_embeddings.{i}.weight"][0, all_designed] = 0.0
    sd[p + "not_a_point_embed.weight"][0, all_designed] = 0.0
    sd[p + "no_mask_embed.weight"][0, all_designed] = 0.0
    A = c["id_amp"]
    sd[p + "point_embeddings.1.weight"][0, ID0 + 5] = A
    sd[p + "point_embeddings.0.weight"][0, ID0 + 5] = A
    sd[p + "not_a_point_embed.weight"][0, PAD_K] = A
    sd[p + "not_a_point_embed.weight"][0, PAD_V] = A
    m_ = "mask_decoder."
    sd[m_ + "iou_token.weight"][0, all_designed] = 0.0
    sd[m_ + "mask_tokens.weight"][:, all_designed] = 0.0
    sd[m_ + "iou_token.weight"][0, ID0] = A
    for i in range(4):
        sd[m_ + "mask_tokens.weight"][i, ID0 + 1 + i] = A

    # ---------------- token side: nothing random writes the designed token channels
    t = m_ + "transformer."
    for li in range(2):
        lp = f"{t}layers.{li}."
        for nm in ("self_attn.out_proj", "cross_attn_token_to_image.out_proj", "mlp.lin2"):
            sd[lp + nm + ".weight"][tok_designed] = 0.0
            sd[lp + nm + ".bias"][tok_designed] = 0.0
        for nm in ("norm1", "norm2", "norm3"):
            sd[lp + nm + ".weight"][tok_designed] = 1.0
            sd[lp + nm + ".bias"][tok_designed] = 0.0
    sd[t + "final_attn_token_to_image.out_proj.weight"][tok_designed] = 0.0
    sd[t + "final_attn_token_to_image.out_proj.bias"][tok_designed] = 0.0
    # layer 0 self attention (it REPLACES the tokens): head 0 = identity on the id and PE-bit channels
    sa = t + "layers.0.self_attn."
    aq = math.sqrt(12.0 * math.sqrt(32.0)) / A
    for nm in ("q_proj", "k_proj", "v_proj"):
        sd[sa + nm + ".weight"][:32] = 0.0
        sd[sa + nm + ".bias"][:32] = 0.0
    sd[sa + "out_proj.weight"][:, :32] = 0.0
    for i in range(8):
        sd[sa + "q_proj.weight"][i, ID0 + i] = aq
        sd[sa + "k_proj.weight"][i, ID0 + i] = aq
        sd[sa + "v_proj.weight"][i, ID0 + i] = 1.0
        sd[sa + "out_proj.weight"][ID0 + i, i] = 1.0
        sd[sa + "v_proj.weight"][8 + i, P0 + i] = 1.0
        sd[sa + "out_proj.weight"][P0 + i, 8 + i] = 1.0
    # layer 0 token MLP: saturate the PE bit channels to +-bit_amp (0 for the tokens without a position)
    l0 = t + "layers.0."
    sd[l0 + "norm2.weight"][bch] = 0.0
    gsat, bit_amp = 40.0, 4.0
    sd[l0 + "mlp.lin1.weight"][:16] = 0.0
    sd[l0 + "mlp.lin2.weight"][:, :16] = 0.0
    for k in range(8):
        sd[l0 + "mlp.lin1.weight"][2 * k, P0 + k], sd[l0 + "mlp.lin1.weight"][2 * k, TREF] = gsat, -gsat
        sd[l0 + "mlp.lin1.weight"][2 * k + 1, P0 + k], sd[l0 + "mlp.lin1.weight"][2 * k + 1, TREF] = gsat, -gsat
        sd[l0 + "mlp.lin1.bias"][2 * k], sd[l0 + "mlp.lin1.bias"][2 * k + 1] = 1.0, -1.0
        sd[l0 + "mlp.lin2.weight"][B0 + k, 2 * k], sd[l0 + "mlp.lin2.weight"][B0 + k, 2 * k + 1] = bit_amp, -bit_amp
        sd[l0 + "mlp.lin2.bias"][B0 + k] = -bit_amp
    sd[l0 + "norm3.weight"][PAD_K] = 0.0

    # ---------------- layer 0 image->token attention: heads 0 / 1 = block match in x / y
    ia = l0 + "cross_attn_image_to_token."
    for nm in ("q_proj", "k_proj", "v_proj"):
        sd[ia + nm + ".weight"][:32] = 0.0
        sd[ia + nm + ".bias"][:32] = 0.0
    sd[ia + "out_proj.weight"][:, :32] = 0.0
    sd[ia + "out_proj.weight"] *= 0.5
    # one matching bit scores a^2 * kappa_x * bit_token / 4 with kappa_x = the embedding's code amplitude after the neck's two
    # LayerNorms (0.4 .. 0.9, measured on the oracle) and bit_token = the token bit amplitude after norm3 (3.0 +- 4 %)
    a = math.sqrt(4.0 * c["unit"] / (c["kappa_min"] * c["bit_token"]))
    for h in range(2):
        for k in range(4):
            sd[ia + "q_proj.weight"][16 * h + k, C0 + 4 * h + k] = a
            sd[ia + "q_proj.weight"][16 * h + k, EREF] = -a
            sd[ia + "k_proj.weight"][16 * h + k, B0 + 4 * h + k] = a
            sd[ia + "k_proj.weight"][16 * h + k, TREF] = -a
        sd[ia + "q_proj.weight"][16 * h + 4, CONE], sd[ia + "q_proj.weight"][16 * h + 4, EREF] = a, -a
        sd[ia + "k_proj.weight"][16 * h + 4, PAD_K] = 3.0 * a * c["bit_token"] / A      # s_pad = 3 x (one matching bit): own block 4, else <= 2
        sd[ia + "v_proj.weight"][16 * h, PAD_V], sd[ia + "v_proj.weight"][16 * h, TREF] = -1.0, 1.0
    for li in range(2):
        lp = f"{t}layers.{li}."
        sd[lp + "cross_attn_image_to_token.out_proj.weight"][img_designed] = 0.0
        sd[lp + "cross_attn_image_to_token.out_proj.bias"][img_designed] = 0.0
        sd[lp + "norm4.weight"][img_designed] = 1.0
        sd[lp + "norm4.bias"][img_designed] = 0.0
    for h in range(2):
        sd[ia + "out_proj.weight"][EM0 + h, 16 * h] = 1.0
    sd[t + "layers.1.cross_attn_image_to_token.out_proj.weight"] *= 0.5

    # ---------------- up-scaling: +x / -x pairs
    u0 = m_ + "output_upscaling.0."
    W1 = sd[u0 + "weight"]                                            # [256, 64, 2, 2]
    g1, g1m = 6.0, 3.0
    W1[:, :28] = 0.0
    W1[img_designed, :] = 0.0
    W1[:, 46:64] = -W1[:, 28:46]
    for ky in range(2):
        for kx in range(2):
            for tt in range(3):
                for j in range(4):
                    s = (2 * ky + j // 2) * 4 + (2 * kx + j % 2)
                    cp, cm = 8 * tt + j, 8 * tt + 4 + j
                    W1[EI0 + 16 * tt + s, cp, ky, kx], W1[EREF, cp, ky, kx] = g1, -g1
                    W1[EI0 + 16 * tt + s, cm, ky, kx], W1[EREF, cm, ky, kx] = -g1, g1
            for h in range(2):
                W1[EM0 + h, 24 + h, ky, kx], W1[EREF, 24 + h, ky, kx] = g1m, -g1m
                W1[EM0 + h, 26 + h, ky, kx], W1[EREF, 26 + h, ky, kx] = -g1m, g1m
    b1 = sd[u0 + "bias"]
    b1[:28] = 0.0
    b1[46:64] = -b1[28:46]
    ln = m_ + "output_upscaling.1."
    sd[ln + "weight"][:28] = 1.0
    sd[ln + "bias"][:28] = 0.0
    sd[ln + "weight"][46:64] = sd[ln + "weight"][28:46]
    u3 = m_ + "output_upscaling.3."
    W2 = sd[u3 + "weight"]                                            # [64, 32, 2, 2]
    W2[:, :10] = 0.0
    W2[:28, :] = 0.0
    for ky in range(2):
        for kx in range(2):
            j = 2 * ky + kx
            for tt in range(3):
                W2[8 * tt + j, 2 * tt, ky, kx], W2[8 * tt + 4 + j, 2 * tt, ky, kx] = 1.0, -1.0
                W2[8 * tt + j, 2 * tt + 1, ky, kx], W2[8 * tt + 4 + j, 2 * tt + 1, ky, kx] = -1.0, 1.0
            for h in range(2):
                W2[24 + h, 6 + h, ky, kx], W2[26 + h, 6 + h, ky, kx] = 1.0, -1.0
                W2[24 + h, 8 + h, ky, kx], W2[26 + h, 8 + h, ky, kx] = -1.0, 1.0
    sd[u3 + "bias"][:10] = 0.0

    # ---------------- hyper-networks and IoU head: designed last-layer bias + small random weights
    Hi, Hm = 14.0, 14.0
    for i in range(4):
        hp = f"{m_}output_hypernetworks_mlps.{i}."
        sd[hp + "layers.2.weight"] = n(32, 256, std=0.3 / 16)
        b2 = n(32, std=0.6)
        b2[:10] = 0.0
        tt = max(i - 1, 0)
        b2[2 * tt], b2[2 * tt + 1] = Hi, -Hi
        for h in range(2):
            b2[6 + h], b2[8 + h] = Hm, -Hm
        sd[hp + "layers.2.bias"] = b2
    ip = m_ + "iou_prediction_head."
    # predicted IoUs: spread of ~0.03 between prompts (few near-ties in the NMS order); the systematic part of the random
    # last layer is removed by the calibration of tools/calibrate_cells.py when one exists for this (model type, seed)
    sd[ip + "layers.2.weight"] = n(4, 256, std=0.2 / 16)
    bias = torch.tensor([0.93, 0.95, 0.94, 0.93])
    if iou_offset is not None:
        bias = bias - torch.tensor(iou_offset, dtype=torch.float32)
    sd[ip + "layers.2.bias"] = bias
