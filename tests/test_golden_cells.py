"""The committed fp32 reference fixture of the benchmarked configuration is self-consistent (CPU only)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cells_vit_b_tile1000.npz")


def test_golden_fixture_is_consistent():
    from oracle import amg_ref as A
    g = np.load(GOLDEN)
    assert g["iou_preds"].shape == (3072,) and g["boxes"].shape == (3072, 4)
    kept, off, cnt = g["kept"], g["rle_offsets"], g["rle_counts"]
    assert len(kept) + 1 == len(off) and off[-1] == len(cnt) and len(kept) > 100
    assert (g["iou_preds"][kept] > 0.88).all() and (g["stability"][kept] >= 0.95).all()
    lab = g["labels"]
    assert lab.shape == (1024, 1024) and 100 < lab.max() < 1000
    for j in (0, len(kept) // 2, len(kept) - 1):
        c = cnt[off[j]:off[j + 1]]
        assert c.sum() == 1024 * 1024
        m = A.rle_to_mask({"size": [1024, 1024], "counts": c})
        ys, xs = np.where(m)
        assert [xs.min(), ys.min(), xs.max(), ys.max()] == g["boxes"][kept[j]].tolist()   # batched_mask_to_box convention


def test_cells_checkpoint_prompt_lattice():
    """The designed part of the synthetic 'cells' checkpoint: prompt bits == lattice block bits of the image tokens."""
    import torch
    from micro_sam_amd.synthetic import CELLS, _block_bits, synthetic_state_dict
    sd = synthetic_state_dict("vit_b", 0, variant="cells")
    pos = sd["image_encoder.pos_embed"][0]
    centre = torch.arange(64, dtype=torch.float32) * 16 + 8
    assert torch.equal(pos[5, :, 19:23], _block_bits(centre)) and torch.equal(pos[:, 7, 23:27], _block_bits(centre))
    G = sd["prompt_encoder.pe_layer.positional_encoding_gaussian_matrix"]
    for px in (16.0, 48.0, 528.0, 1008.0):
        u = (px + 0.5) / 1024.0
        pe = torch.sin(2 * torch.pi * (2 * u - 1) * G[0, CELLS["P0"]:CELLS["P0"] + 4])
        assert torch.equal(torch.sign(pe), _block_bits(torch.tensor(px + 0.5)))
