"""Pin the oracle's model arithmetic against transformers.models.sam (same network, independent code)."""
import numpy as np
import pytest
import torch

from oracle import amg_ref as A
from oracle import sam_ref as S
from micro_sam_amd.synthetic import synthetic_tile

transformers = pytest.importorskip("transformers")


@pytest.fixture(scope="module")
def hf_and_feats(vit_b_sd):
    from hf_map import load_into_hf
    hf = load_into_hf(vit_b_sd)
    img = A.to_image(synthetic_tile(0))
    x = S.preprocess(torch.as_tensor(img).permute(2, 0, 1)[None])
    with torch.no_grad():
        f_o = S.image_encoder(vit_b_sd, x)
        f_h = hf.vision_encoder(x).last_hidden_state
    return hf, f_o, f_h


def test_encoder_matches_hf(hf_and_feats):
    _, f_o, f_h = hf_and_feats
    assert f_o.shape == (1, 256, 64, 64)
    assert (f_o - f_h).abs().max().item() < 2e-4


def test_decoder_points_matches_hf(vit_b_sd, hf_and_feats):
    hf, f_o, _ = hf_and_feats
    pts = torch.tensor([[[100.5, 200.25]], [[512.0, 512.0]], [[900.0, 30.0]]])
    lbl = torch.ones(3, 1, dtype=torch.int)
    with torch.no_grad():
        _, iou, low = S.predict_torch(vit_b_sd, f_o, (1024, 1024), (1024, 1024), pts, lbl, return_logits=True)
        out = hf(image_embeddings=f_o, input_points=pts[None], input_labels=lbl[None].long(), multimask_output=True)
    scale = low.abs().max().item()
    assert (out.pred_masks[0] - low).abs().max().item() < 1e-5 * scale + 1e-3
    assert (out.iou_scores[0] - iou).abs().max().item() < 1e-5


def test_decoder_box_matches_hf(vit_b_sd, hf_and_feats):
    hf, f_o, _ = hf_and_feats
    bx = torch.tensor([[100., 100., 400., 300.]])
    with torch.no_grad():
        _, iou, low = S.predict_torch(vit_b_sd, f_o, (1024, 1024), (1024, 1024), None, None, boxes=bx,
                                      multimask_output=False, return_logits=True)
        out = hf(image_embeddings=f_o, input_boxes=bx[None], multimask_output=False)
    scale = low.abs().max().item()
    assert (out.pred_masks[0] - low).abs().max().item() < 1e-5 * scale + 1e-3
    assert (out.iou_scores[0] - iou).abs().max().item() < 1e-5


def test_bf16_mode_close_to_fp32(vit_b_sd, hf_and_feats):
    _, f_o, _ = hf_and_feats
    img = A.to_image(synthetic_tile(0))
    x = S.preprocess(torch.as_tensor(img).permute(2, 0, 1)[None])
    with torch.no_grad():
        f_b = S.image_encoder(vit_b_sd, x, precision="bf16")
    # embeddings are LayerNorm2d outputs (unit scale): bf16 operand rounding stays at the 1e-2 level
    assert (f_b - f_o).abs().mean().item() < 2e-2
    assert (f_b - f_o).abs().max().item() < 0.25
