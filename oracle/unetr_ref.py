"""CPU oracle of the convolutional decoder of micro_sam's automatic instance segmentation.  TEST INFRASTRUCTURE ONLY (only ``tests/``
import it; the product never does).

What it restates: the graph the reference itself fixes - ``DecoderAdapter._forward_impl`` (``micro_sam/instance_segmentation.py:710-733``):

    z9 = deconv1(z12); z6 = deconv2(z9); z3 = deconv3(z6); z0 = deconv4(z3)
    x = decoder(base(z12), encoder_inputs=[z9, z6, z3]); x = deconv_out(x)
    x = out_conv(decoder_head(cat([x, z0], dim=1))); final activation; postprocess_masks (:735, torch_em UNETR.postprocess_masks)

over torch_em's published blocks (torch_em is not vendored in the reference and not installed here: PARITY UNPINNED against the library
itself - the layer names / widths are those of the checkpoint, ``micro_sam_amd/models/unetr.py`` has the details):

    ConvBlock2d.block        = (InstanceNorm2d, Conv2d 3x3 pad 1, ReLU) x 2 - convolutions at indices 1 and 4
    Deconv2DBlock.block      = (up-sampler, SingleConv2DBlock[.block = Conv2d 3x3 pad 1], BatchNorm2d, ReLU)
    up-sampler               = SingleDeconv2DBlock[.block = ConvTranspose2d(k 2, s 2)]  or  Upsampler2d[bilinear x2, .conv = Conv2d 1x1]
    Decoder                  = per level: x = sampler_i(x); x = block_i(cat([x, skip_i], dim=1))

Functional fp32 torch over the decoder's ``state_dict`` (keys without the ``encoder.`` part), eval mode (BatchNorm on its running
statistics).  Independent of ``micro_sam_amd`` - it shares no code with the module tree or the HIP path it checks."""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
IMG_SIZE = 1024


def _conv_block(sd: Dict[str, Tensor], pre: str, x: Tensor) -> Tensor:
    for idx in (1, 4):
        x = F.instance_norm(x, eps=1e-5)
        x = F.relu(F.conv2d(x, sd[f"{pre}.block.{idx}.weight"], sd[f"{pre}.block.{idx}.bias"], padding=1))
    return x


def _upsample(sd: Dict[str, Tensor], pre: str, x: Tensor) -> Tensor:
    if f"{pre}.block.weight" in sd:                                   # SingleDeconv2DBlock
        return F.conv_transpose2d(x, sd[f"{pre}.block.weight"], sd[f"{pre}.block.bias"], stride=2)
    x = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)        # Upsampler2d
    return F.conv2d(x, sd[f"{pre}.conv.weight"], sd[f"{pre}.conv.bias"])


def _deconv_block(sd: Dict[str, Tensor], pre: str, x: Tensor) -> Tensor:
    x = _upsample(sd, f"{pre}.block.0", x)
    x = F.conv2d(x, sd[f"{pre}.block.1.block.weight"], sd[f"{pre}.block.1.block.bias"], padding=1)
    x = F.batch_norm(x, sd[f"{pre}.block.2.running_mean"], sd[f"{pre}.block.2.running_var"], sd[f"{pre}.block.2.weight"],
                     sd[f"{pre}.block.2.bias"], training=False, eps=1e-5)
    return F.relu(x)


def decode(sd: Dict[str, Tensor], z12: Tensor, final_activation: Optional[str] = "Sigmoid") -> Tensor:
    """Image embeddings [B, 256, 64, 64] -> [B, out_channels, 1024, 1024]."""
    z12 = z12.to(torch.float32)
    z9 = _deconv_block(sd, "deconv1", z12)
    z6 = _deconv_block(sd, "deconv2", z9)
    z3 = _deconv_block(sd, "deconv3", z6)
    z0 = _deconv_block(sd, "deconv4", z3)
    x = _conv_block(sd, "base", z12)
    for i, skip in enumerate((z9, z6, z3)):
        x = _upsample(sd, f"decoder.samplers.{i}", x)
        x = _conv_block(sd, f"decoder.blocks.{i}", torch.cat([x, skip], dim=1))
    x = _upsample(sd, "deconv_out", x)
    x = _conv_block(sd, "decoder_head", torch.cat([x, z0], dim=1))
    x = F.conv2d(x, sd["out_conv.weight"], sd["out_conv.bias"])
    if final_activation == "Sigmoid":
        x = torch.sigmoid(x)
    elif final_activation is not None:
        raise ValueError(final_activation)
    return x


def postprocess_masks(masks: Tensor, input_size: Tuple[int, int], original_size: Tuple[int, int]) -> Tensor:
    masks = F.interpolate(masks, (IMG_SIZE, IMG_SIZE), mode="bilinear", align_corners=False)
    masks = masks[..., : input_size[0], : input_size[1]]
    return F.interpolate(masks, tuple(original_size), mode="bilinear", align_corners=False)


@torch.no_grad()
def decoder_forward(sd: Dict[str, Tensor], embeddings: Tensor, input_shape, original_shape, final_activation: Optional[str] = "Sigmoid") -> Tensor:
    """``DecoderAdapter.forward`` (reference :732-735)."""
    return postprocess_masks(decode(sd, embeddings, final_activation), tuple(input_shape), tuple(original_shape))
