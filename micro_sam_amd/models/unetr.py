"""The convolutional decoder of micro_sam's automatic instance segmentation: ``torch_em.model.UNETR`` with a SAM image encoder as backbone,
as the reference builds and wraps it (``micro_sam/instance_segmentation.py:688-828``: ``DecoderAdapter``, ``get_unetr``, ``get_decoder``).

PARITY UNPINNED.  torch_em is neither vendored in the reference nor installed here, so this file restates its published module tree from
what the reference itself fixes -

  * the adapter's forward graph (``DecoderAdapter._forward_impl`` :710-733: z9 = deconv1(z12), z6 = deconv2(z9), z3 = deconv3(z6),
    z0 = deconv4(z3); x = decoder(base(z12), [z9, z6, z3]); x = decoder_head(cat(deconv_out(x), z0)); out_conv; final activation),
  * the attribute names it reads (``base, out_conv, deconv_out, decoder_head, final_activation, postprocess_masks, decoder, deconv1..4``),
  * how ``get_unetr`` tells the two up-sampler flavours apart (:765-772: ``decoder.samplers.*`` keys contain ``.block.`` for transposed
    convolutions - ``SingleDeconv2DBlock.block`` -, ``.conv.`` for bilinear interpolation + 1 x 1 convolution - ``Upsampler2d.conv``) -

and from torch_em's block definitions (``ConvBlock2d``: InstanceNorm, 3 x 3 conv, ReLU, twice, as ``block`` = Sequential with the convolutions
at 1 and 4; ``Deconv2DBlock.block`` = Sequential(up-sampler, ``SingleConv2DBlock`` 3 x 3, BatchNorm2d, ReLU); ``Decoder.blocks / .samplers``).
What could NOT be restated with confidence is the channel width of every layer, so **the widths are read from the checkpoint**: given a
``decoder_state`` the module tree is built to the shapes of its tensors (a state with other key names fails loudly, naming them); without one
(a fresh decoder for training) the self-consistent default below is used.  The module tree here owns the parameters and is the taped
(training) form; INFERENCE on a GPU runs the library's fp32 kernels through ``models/unetr_hip.py`` (round 5; checked against
``oracle/unetr_ref.py``), its input, the image embedding, comes from the HIP encoder."""
from __future__ import annotations

import warnings
from collections import OrderedDict
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

IMG_SIZE = 1024
# default widths (no checkpoint): skip of level i = output width of that level's sampler, so that cat(sampler(x), skip) has the input width
# of the level's block; the chain z9 -> z6 -> z3 -> z0 follows the same halving
DEFAULT_FEATURES = (512, 256, 128, 64)


class ConvBlock2d(nn.Module):
    """torch_em ``ConvBlock2d``: (InstanceNorm2d, Conv2d 3 x 3, ReLU) twice."""

    def __init__(self, in_channels: int, out_channels: int) -> None:
        super().__init__()
        self.block = nn.Sequential(nn.InstanceNorm2d(in_channels), nn.Conv2d(in_channels, out_channels, 3, padding=1), nn.ReLU(inplace=True),
                                   nn.InstanceNorm2d(out_channels), nn.Conv2d(out_channels, out_channels, 3, padding=1), nn.ReLU(inplace=True))

    def forward(self, x):
        return self.block(x)


class Upsampler2d(nn.Module):
    """Bilinear interpolation by ``scale_factor`` followed by a 1 x 1 convolution (torch_em ``Upsampler2d``)."""

    def __init__(self, scale_factor: int, in_channels: int, out_channels: int) -> None:
        super().__init__()
        self.scale_factor = scale_factor
        self.conv = nn.Conv2d(in_channels, out_channels, 1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=self.scale_factor, mode="bilinear", align_corners=False))


class SingleDeconv2DBlock(nn.Module):
    def __init__(self, scale_factor: int, in_channels: int, out_channels: int) -> None:
        super().__init__()
        self.block = nn.ConvTranspose2d(in_channels, out_channels, kernel_size=2, stride=2, padding=0, output_padding=0)

    def forward(self, x):
        return self.block(x)


class SingleConv2DBlock(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, kernel_size: int) -> None:
        super().__init__()
        self.block = nn.Conv2d(in_channels, out_channels, kernel_size, stride=1, padding=(kernel_size - 1) // 2)

    def forward(self, x):
        return self.block(x)


class Deconv2DBlock(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, kernel_size: int = 3, use_conv_transpose: bool = True) -> None:
        super().__init__()
        up = SingleDeconv2DBlock if use_conv_transpose else Upsampler2d
        self.block = nn.Sequential(up(2, in_channels, out_channels), SingleConv2DBlock(out_channels, out_channels, kernel_size),
                                   nn.BatchNorm2d(out_channels), nn.ReLU(True))

    def forward(self, x):
        return self.block(x)


class Decoder(nn.Module):
    """torch_em ``Decoder``: per level up-sample, concatenate the skip input, convolution block."""

    def __init__(self, block_io: List[Tuple[int, int]], sampler_io: List[Tuple[int, int]], use_conv_transpose: bool) -> None:
        super().__init__()
        up = SingleDeconv2DBlock if use_conv_transpose else Upsampler2d
        self.blocks = nn.ModuleList([ConvBlock2d(i, o) for i, o in block_io])
        self.samplers = nn.ModuleList([up(2, i, o) for i, o in sampler_io])

    def forward(self, x, encoder_inputs):
        assert len(encoder_inputs) == len(self.blocks)
        for block, sampler, skip in zip(self.blocks, self.samplers, encoder_inputs):
            x = sampler(x)
            x = block(torch.cat([x, skip], dim=1))
        return x


def _widths_from_state(state: Dict[str, torch.Tensor]) -> Dict[str, object]:
    """Layer widths and the up-sampler flavour of a ``decoder_state`` (keys without the ``encoder.`` part)."""
    def conv_io(key):
        if key not in state:
            raise RuntimeError(f"decoder_state: the parameters for '{key}' could not be found (keys present: {sorted(state)[:8]} ...)")
        w = state[key]
        return int(w.shape[1]), int(w.shape[0])
    transpose = any(".block." in k for k in state if k.startswith("decoder.samplers"))
    upkey = "block" if transpose else "conv"

    def up_io(prefix):
        w = state.get(f"{prefix}.{upkey}.weight")
        if w is None:
            raise RuntimeError(f"decoder_state: the parameters for '{prefix}.{upkey}.weight' could not be found")
        return (int(w.shape[0]), int(w.shape[1])) if transpose else (int(w.shape[1]), int(w.shape[0]))     # ConvTranspose2d weight is [in, out, 2, 2]
    n_levels = len({k.split(".")[2] for k in state if k.startswith("decoder.blocks.")})
    return {
        "use_conv_transpose": transpose,
        "base": conv_io("base.block.1.weight"),
        "blocks": [conv_io(f"decoder.blocks.{i}.block.1.weight") for i in range(n_levels)],
        "samplers": [up_io(f"decoder.samplers.{i}") for i in range(n_levels)],
        "deconv": [up_io(f"deconv{i}.block.0") for i in (1, 2, 3, 4)],
        "deconv_out": up_io("deconv_out"),
        "head": conv_io("decoder_head.block.1.weight"),
        "out": conv_io("out_conv.weight"),
    }


def _default_widths(embed_dim: int, out_channels: int, use_conv_transpose: bool) -> Dict[str, object]:
    f = DEFAULT_FEATURES
    return {
        "use_conv_transpose": use_conv_transpose,
        "base": (embed_dim, f[0]),
        "blocks": [(f[i], f[i + 1]) for i in range(3)],
        "samplers": [(f[i], f[i + 1]) for i in range(3)],
        "deconv": [(embed_dim, f[1]), (f[1], f[2]), (f[2], f[3]), (f[3], f[3])],
        "deconv_out": (f[3], f[3]),
        "head": (2 * f[3], f[3]),
        "out": (f[3], out_channels),
    }


class UNETR(nn.Module):
    """``torch_em.model.UNETR(backbone="sam", encoder=<image encoder>, use_skip_connection=False, resize_input=True, use_sam_stats=True)``:
    the attribute names the reference's ``DecoderAdapter`` reads, ``forward`` for whole images (encoder + decoder)."""

    def __init__(self, encoder: nn.Module, widths: Dict[str, object], final_activation: Optional[str] = "Sigmoid") -> None:
        super().__init__()
        self.encoder = encoder
        self.use_conv_transpose = bool(widths["use_conv_transpose"])
        t = self.use_conv_transpose
        up = SingleDeconv2DBlock if t else Upsampler2d
        self.decoder = Decoder(widths["blocks"], widths["samplers"], t)
        d = widths["deconv"]
        self.deconv1, self.deconv2 = Deconv2DBlock(*d[0], use_conv_transpose=t), Deconv2DBlock(*d[1], use_conv_transpose=t)
        self.deconv3, self.deconv4 = Deconv2DBlock(*d[2], use_conv_transpose=t), Deconv2DBlock(*d[3], use_conv_transpose=t)
        self.base = ConvBlock2d(*widths["base"])
        self.out_conv = nn.Conv2d(widths["out"][0], widths["out"][1], 1)
        self.deconv_out = up(2, *widths["deconv_out"])
        self.decoder_head = ConvBlock2d(*widths["head"])
        self.out_channels = int(widths["out"][1])
        if final_activation is None:
            self.final_activation = None
        elif isinstance(final_activation, str):
            act = getattr(nn, final_activation, None)
            if act is None:
                raise ValueError(f"Invalid activation: {final_activation}")
            self.final_activation = act()
        else:
            self.final_activation = final_activation
        self.register_buffer("pixel_mean", torch.tensor([123.675, 116.28, 103.53]).view(1, -1, 1, 1), persistent=False)
        self.register_buffer("pixel_std", torch.tensor([58.395, 57.12, 57.375]).view(1, -1, 1, 1), persistent=False)

    def postprocess_masks(self, masks: torch.Tensor, input_size, original_size) -> torch.Tensor:
        masks = F.interpolate(masks, (IMG_SIZE, IMG_SIZE), mode="bilinear", align_corners=False)
        masks = masks[..., : input_size[0], : input_size[1]]
        return F.interpolate(masks, tuple(original_size), mode="bilinear", align_corners=False)

    def decode(self, z12: torch.Tensor) -> torch.Tensor:
        """The decoder on image embeddings [B, 256, 64, 64] -> [B, out_channels, 1024, 1024] (the adapter's ``_forward_impl``)."""
        z9 = self.deconv1(z12)
        z6 = self.deconv2(z9)
        z3 = self.deconv3(z6)
        z0 = self.deconv4(z3)
        x = self.base(z12)
        x = self.decoder(x, encoder_inputs=[z9, z6, z3])
        x = self.deconv_out(x)
        x = torch.cat([x, z0], dim=1)
        x = self.decoder_head(x)
        x = self.out_conv(x)
        if self.final_activation is not None:
            x = self.final_activation(x)
        return x

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """Whole images [B, 3, H, W] in 0..255: resize the longest side to 1024, SAM statistics, pad, encode, decode, back to (H, W)."""
        original = tuple(x.shape[-2:])
        scale = IMG_SIZE / max(original)
        size = (int(original[0] * scale + 0.5), int(original[1] * scale + 0.5))
        x = F.interpolate(x.float(), size, mode="bilinear", align_corners=False)
        x = (x - self.pixel_mean.to(x.device)) / self.pixel_std.to(x.device)
        x = F.pad(x, (0, IMG_SIZE - size[1], 0, IMG_SIZE - size[0]))
        z12 = self.encoder(x)
        if isinstance(z12, (tuple, list)):
            z12 = z12[0]
        return self.postprocess_masks(self.decode(z12.float()), size, original)


class DecoderAdapter(nn.Module):
    """Reference ``DecoderAdapter`` (:688-735): the UNETR decoder as one module applied to precomputed embeddings."""

    def __init__(self, unetr: nn.Module) -> None:
        super().__init__()
        self.base = unetr.base
        self.out_conv = unetr.out_conv
        self.deconv_out = unetr.deconv_out
        self.decoder_head = unetr.decoder_head
        self.final_activation = unetr.final_activation
        self.postprocess_masks = unetr.postprocess_masks
        self.decoder = unetr.decoder
        self.deconv1, self.deconv2, self.deconv3, self.deconv4 = unetr.deconv1, unetr.deconv2, unetr.deconv3, unetr.deconv4

    def _forward_impl(self, input_):
        z12 = input_
        z9 = self.deconv1(z12)
        z6 = self.deconv2(z9)
        z3 = self.deconv3(z6)
        z0 = self.deconv4(z3)
        x = self.base(z12)
        x = self.decoder(x, encoder_inputs=[z9, z6, z3])
        x = self.deconv_out(x)
        x = torch.cat([x, z0], dim=1)
        x = self.decoder_head(x)
        x = self.out_conv(x)
        if self.final_activation is not None:
            x = self.final_activation(x)
        return x

    def _hip(self):
        if getattr(self, "_hip_decoder", None) is None:
            from .unetr_hip import HipUnetrDecoder
            object.__setattr__(self, "_hip_decoder", HipUnetrDecoder(self))
        return self._hip_decoder

    @torch.no_grad()
    def forward(self, input_, input_shape, original_shape):
        """Inference on a GPU runs the library's fp32 kernels (``models/unetr_hip.py``: implicit-GEMM convolutions, InstanceNorm, the
        fused ``postprocess_masks``); ``_forward_impl`` above - the same graph as torch operators - is what trains and what a CPU-only
        caller (the reference's own CPU path) gets.  MSAM_UNETR_TORCH=1 forces the operator form (A/B)."""
        import os
        dev = self.out_conv.weight.device
        if dev.type == "cuda" and os.environ.get("MSAM_UNETR_TORCH", "0") != "1":
            return self._hip().forward(input_.to(device=dev, dtype=torch.float32), input_shape, original_shape)
        x = self._forward_impl(input_.to(device=dev, dtype=torch.float32))
        return self.postprocess_masks(x, input_shape, original_shape)


def get_unetr(image_encoder: nn.Module, decoder_state: Optional["OrderedDict[str, torch.Tensor]"] = None, device=None,
              out_channels: int = 3, flexible_load_checkpoint: bool = False, final_activation: Optional[str] = "Sigmoid") -> nn.Module:
    """Reference ``get_unetr`` (:738-809): a UNETR on the SAM image encoder; with a ``decoder_state`` the decoder parameters are loaded from
    it - strictly (a missing parameter raises) or, with ``flexible_load_checkpoint``, re-initialising what is missing or of another shape."""
    if device is None:
        from .. import util
        device = util.get_device(None)
    device = torch.device(device)               # (the decoder is torch operators: it runs wherever its embeddings are)
    embed_dim = 256
    if decoder_state is None:
        widths = _default_widths(embed_dim, out_channels, use_conv_transpose=False)       # reference: interpolation for up-sampling by default
    else:
        decoder_state = OrderedDict((k, v) for k, v in decoder_state.items() if not k.startswith("encoder"))
        try:
            widths = _widths_from_state(decoder_state)
            if "out" in widths and widths["out"][1] != out_channels:
                # the reference builds the UNETR with the REQUESTED number of output channels: a checkpoint with another head is a
                # size mismatch - an error when loading strictly, a re-initialised out_conv (with the warning below) when
                # flexible_load_checkpoint re-heads a pretrained decoder
                if not flexible_load_checkpoint:
                    raise RuntimeError(f"The parameters for 'out_conv.weight' could not be found or has a size mismatch "
                                       f"({widths['out'][1]} output channels in the checkpoint, {out_channels} requested).")
                widths["out"] = (widths["out"][0], out_channels)
        except RuntimeError:
            if not flexible_load_checkpoint:
                raise
            transpose = any(".block." in k for k in decoder_state if k.startswith("decoder.samplers"))
            widths = _default_widths(embed_dim, out_channels, use_conv_transpose=transpose)
    unetr = UNETR(image_encoder, widths, final_activation=final_activation)
    if decoder_state is not None:
        own = unetr.state_dict()
        for k, v in own.items():
            if k.startswith("encoder") or k in ("pixel_mean", "pixel_std"):
                continue
            if flexible_load_checkpoint:
                if k in decoder_state:
                    if v.shape != decoder_state[k].shape:
                        warnings.warn(f"Shape of '{k}' did not match. Hence, we reinitialize it.")
                    else:
                        own[k] = decoder_state[k]
                else:
                    warnings.warn(f"Could not find '{k}' in the pretrained state dict. Hence, we reinitialize it.")
            else:
                if k not in decoder_state or v.shape != decoder_state[k].shape:
                    raise RuntimeError(f"The parameters for '{k}' could not be found or has a size mismatch.")
                own[k] = decoder_state[k]
        unetr.load_state_dict(own)
    # (the image encoder keeps its own device handling - it is the HIP encoder of the predictor; only the decoder parts move)
    for name, child in unetr.named_children():
        if name != "encoder":
            child.to(device)
    unetr.eval()
    return unetr


def get_decoder(image_encoder: nn.Module, decoder_state: "OrderedDict[str, torch.Tensor]", device=None) -> DecoderAdapter:
    """Reference ``get_decoder`` (:812-828)."""
    return DecoderAdapter(get_unetr(image_encoder, decoder_state, device))
