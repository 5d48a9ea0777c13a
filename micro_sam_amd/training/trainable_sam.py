"""``TrainableSAM`` (reference ``micro_sam/training/trainable_sam.py:12-114``): the wrapper the trainer drives.

Scope of this build: every part of SAM can be trained.  Forward and backward of the parts whose
parameters ``require_grad`` run on the HIP kernels through ``training.functional`` (MFMA GEMM in both directions, LayerNorm /
attention forward + backward kernels; elementwise glue, the 32-channel hyper-network product, the bilinear
``postprocess_masks`` and the losses are torch autograd ops on the device): ``mask_decoder_forward`` below,
``encoders.image_encoder_forward`` and ``encoders.prompt_encoder_forward``.  Parts that are frozen (the reference's
``freeze=[...]`` of ``micro_sam/training/util.py:get_trainable_sam_model``) are evaluated by the inference kernels without a
tape.  The mask decoder path is measured against the fp32 oracle; the encoder paths were written after
round 2's GPU minutes were spent and have their first GPU run pending (their composition is checked on the CPU against the
oracle's autograd, tests/test_training_encoders_host.py).
"""
from __future__ import annotations

from typing import Any, Dict, List, Tuple

import torch
import torch.nn.functional as F
from torch import nn

from ..modeling import GRID, IMG_SIZE, PROMPT_DIM, Sam
from ..transforms import ResizeLongestSide
from . import functional as HF
from .encoders import image_encoder_forward, prompt_encoder_forward


def _dec_attention(mod, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """Upstream ``Attention.forward`` (two-way transformer) on the differentiable HIP primitives."""
    q = HF.linear(q, mod.q_proj.weight, mod.q_proj.bias)
    k = HF.linear(k, mod.k_proj.weight, mod.k_proj.bias)
    v = HF.linear(v, mod.v_proj.weight, mod.v_proj.bias)
    b, nq, c = q.shape
    h = mod.num_heads

    def sep(t):
        return t.reshape(b, t.shape[1], h, c // h).transpose(1, 2)
    out = HF.attention(sep(q), sep(k), sep(v))
    out = out.transpose(1, 2).reshape(b, nq, c)
    return HF.linear(out, mod.out_proj.weight, mod.out_proj.bias)


def _ln(mod, x: torch.Tensor) -> torch.Tensor:
    return HF.layer_norm(x, mod.weight, mod.bias, mod.eps)


def mask_decoder_forward(md, image_embeddings: torch.Tensor, image_pe: torch.Tensor, sparse: torch.Tensor,
                         dense: torch.Tensor, multimask_output: bool) -> Tuple[torch.Tensor, torch.Tensor]:
    """Differentiable ``MaskDecoder.forward`` (upstream ``predict_masks``; same arithmetic as ``oracle/sam_ref.mask_decoder``)
    -> (low_res_masks [B, C, 256, 256], iou_predictions [B, C])."""
    B = sparse.shape[0]
    tokens = torch.cat([md.iou_token.weight, md.mask_tokens.weight], dim=0).unsqueeze(0).expand(B, -1, -1)
    tokens = torch.cat((tokens, sparse), dim=1)
    src = (image_embeddings.expand(B, -1, -1, -1) + dense).flatten(2).permute(0, 2, 1)        # [B, 4096, 256]
    pos = image_pe.expand(B, -1, -1, -1).flatten(2).permute(0, 2, 1)
    tr = md.transformer
    queries, keys, query_pe = tokens, src, tokens
    for i, layer in enumerate(tr.layers):
        if layer.skip_first_layer_pe:
            queries = _dec_attention(layer.self_attn, queries, queries, queries)
        else:
            q = queries + query_pe
            queries = queries + _dec_attention(layer.self_attn, q, q, queries)
        queries = _ln(layer.norm1, queries)
        q, k = queries + query_pe, keys + pos
        queries = _ln(layer.norm2, queries + _dec_attention(layer.cross_attn_token_to_image, q, k, keys))
        m = HF.linear(F.relu(HF.linear(queries, layer.mlp.lin1.weight, layer.mlp.lin1.bias)), layer.mlp.lin2.weight,
                      layer.mlp.lin2.bias)
        queries = _ln(layer.norm3, queries + m)
        q, k = queries + query_pe, keys + pos
        keys = _ln(layer.norm4, keys + _dec_attention(layer.cross_attn_image_to_token, k, q, queries))
    q, k = queries + query_pe, keys + pos
    queries = _ln(tr.norm_final_attn, queries + _dec_attention(tr.final_attn_token_to_image, q, k, keys))
    iou_token_out, mask_tokens_out = queries[:, 0, :], queries[:, 1:5, :]
    # output_upscaling: ConvTranspose2d(256, 64, 2, 2) - LayerNorm2d - GELU - ConvTranspose2d(64, 32, 2, 2) - GELU, as GEMMs on
    # token-major rows (a 2x2 stride-2 transposed convolution = one linear map per input pixel to its 2x2 output block)
    up = md.output_upscaling
    w1 = up[0].weight.permute(2, 3, 1, 0).reshape(4 * 64, PROMPT_DIM)                          # [(ky, kx, co), ci]
    y = HF.linear(keys, w1, up[0].bias.repeat(4))                                              # [B, 4096, 256]
    y = y.reshape(B, GRID, GRID, 2, 2, 64).permute(0, 1, 3, 2, 4, 5).reshape(B, 2 * GRID, 2 * GRID, 64)
    y = F.gelu(HF.layer_norm(y, up[1].weight, up[1].bias, up[1].eps))
    w2 = up[3].weight.permute(2, 3, 1, 0).reshape(4 * 32, 64)
    y = HF.linear(y, w2, up[3].bias.repeat(4))                                                 # [B, 128, 128, 128]
    y = y.reshape(B, 2 * GRID, 2 * GRID, 2, 2, 32).permute(0, 1, 3, 2, 4, 5).reshape(B, 4 * GRID, 4 * GRID, 32)
    y = F.gelu(y)
    hyper = []
    for i, mlp in enumerate(md.output_hypernetworks_mlps):
        t = mask_tokens_out[:, i, :]
        for j, lin in enumerate(mlp.layers):
            t = HF.linear(t, lin.weight, lin.bias)
            if j < len(mlp.layers) - 1:
                t = F.relu(t)
        hyper.append(t)
    hyper = torch.stack(hyper, dim=1)                                                          # [B, 4, 32]
    masks = torch.einsum("bmc,bhwc->bmhw", hyper, y)                                           # plain batched product (rocBLAS)
    t = iou_token_out
    for j, lin in enumerate(md.iou_prediction_head.layers):
        t = HF.linear(t, lin.weight, lin.bias)
        if j < len(md.iou_prediction_head.layers) - 1:
            t = F.relu(t)
    sl = slice(1, None) if multimask_output else slice(0, 1)
    return masks[:, sl, :, :], t[:, sl]


def postprocess_masks(masks: torch.Tensor, input_size, original_size) -> torch.Tensor:
    """Differentiable ``Sam.postprocess_masks`` (bilinear to 1024^2, crop the padding, bilinear to the original size)."""
    masks = F.interpolate(masks, (IMG_SIZE, IMG_SIZE), mode="bilinear", align_corners=False)
    masks = masks[..., : input_size[0], : input_size[1]]
    return F.interpolate(masks, tuple(original_size), mode="bilinear", align_corners=False)


class TrainableSAM(nn.Module):
    """Reference ``TrainableSAM`` (same methods / record formats)."""

    def __init__(self, sam: Sam) -> None:
        super().__init__()
        self.sam = sam
        self.transform = ResizeLongestSide(sam.image_encoder.img_size)

    def preprocess(self, x: torch.Tensor) -> Tuple[torch.Tensor, Tuple[int, int]]:
        x = self.transform.apply_image_torch(x)
        input_size = x.shape[-2:]
        x = (x - self.sam.pixel_mean.unsqueeze(0)) / self.sam.pixel_std.unsqueeze(0)
        h, w = x.shape[-2:]
        x = F.pad(x, (0, self.sam.image_encoder.img_size - w, 0, self.sam.image_encoder.img_size - h))
        return x, input_size

    def _trains(self, module: nn.Module) -> bool:
        return torch.is_grad_enabled() and any(p.requires_grad for p in module.parameters())

    def image_embeddings_oft(self, batched_inputs):
        input_images, input_size = self.preprocess(
            torch.stack([x["image"] for x in batched_inputs], dim=0).to(self.sam.device, non_blocking=True).float())
        for i in range(len(batched_inputs)):
            batched_inputs[i]["input_size"] = input_size
        if self._trains(self.sam.image_encoder):
            if hasattr(self.sam.image_encoder, "forward_taped"):       # vit_t: TinyViT is a tree of torch operators, autograd is its tape
                image_embeddings = self.sam.image_encoder.forward_taped(input_images)
            else:
                image_embeddings = image_encoder_forward(self.sam.image_encoder, input_images)      # with a tape
        else:
            # (the inference kernels' operand copies rebuild themselves when a parameter's version counter moved:
            # modeling.ImageEncoderViT._prepare / Sam._prepare_decoder)
            image_embeddings = self.sam.image_encoder(input_images)       # inference kernels, no tape
        return image_embeddings, batched_inputs

    def forward(self, batched_inputs: List[Dict[str, Any]], image_embeddings: torch.Tensor,
                multimask_output: bool = False) -> List[Dict[str, Any]]:
        dev = self.sam.device
        train_prompt = self._trains(self.sam.prompt_encoder)
        # the decoder needs a tape when it trains itself or when a gradient has to pass through it to the encoders
        train = self._trains(self.sam.mask_decoder) or train_prompt or (torch.is_grad_enabled() and image_embeddings.requires_grad)
        outputs = []
        for image_record, curr_embedding in zip(batched_inputs, image_embeddings):
            points = None
            if "point_coords" in image_record:
                points = (image_record["point_coords"].to(dev, non_blocking=True), image_record["point_labels"].to(dev, non_blocking=True))
            boxes = image_record["boxes"].to(dev, non_blocking=True) if "boxes" in image_record else None
            masks = image_record["mask_inputs"].to(dev, non_blocking=True) if "mask_inputs" in image_record else None
            if train_prompt:
                sparse, dense = prompt_encoder_forward(self.sam.prompt_encoder, points, boxes, masks)
            else:
                sparse, dense = self.sam.prompt_encoder(points=points, boxes=boxes, masks=masks)
            if train:
                low_res_masks, iou_predictions = mask_decoder_forward(
                    self.sam.mask_decoder, curr_embedding.unsqueeze(0), self.sam.prompt_encoder.get_dense_pe(), sparse, dense,
                    multimask_output)
            else:
                low_res_masks, iou_predictions = self.sam.mask_decoder(
                    image_embeddings=curr_embedding.unsqueeze(0), image_pe=self.sam.prompt_encoder.get_dense_pe(),
                    sparse_prompt_embeddings=sparse, dense_prompt_embeddings=dense, multimask_output=multimask_output)
            masks_out = postprocess_masks(low_res_masks, image_record["input_size"], image_record["original_size"])
            outputs.append({"low_res_masks": low_res_masks, "masks": masks_out, "iou_predictions": iou_predictions})
        return outputs
