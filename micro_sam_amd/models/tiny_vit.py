"""MobileSAM's image encoder (TinyViT-5M) behind ``get_sam_model("vit_t")`` - BASELINE configs[0], the reference's CPU-runnable
plumbing case (``micro_sam/util.py:35-43,435-439``: ``mobile_sam.sam_model_registry["vit_t"]``).

Module tree and parameter names of ``mobile_sam/modeling/tiny_vit_sam.py`` (MobileSAM / micro_sam ``vit_t*`` checkpoints load with
``load_state_dict``; the classification head ``norm_head`` / ``head`` of the checkpoints is kept as parameters and not evaluated).  Hyper-
parameters: ``mobile_sam/build_sam.py`` build_sam_vit_t.

SCOPE.  This encoder is NOT on the hand-written HIP path: its convolutions (strided 3 x 3 stem, depthwise 3 x 3, BatchNorm) and its
32-channel-head window attention with offset-bias tables have no kernel in ``csrc/`` yet, so they run as torch operators on the module's
device (MIOpen / hipBLASLt on the GPU) in fp32 - plumbing for the vit_t API surface and config 1's checks; prompt encoder, mask
decoder and everything after the embedding are the HIP path of every other model type.  Nothing here falls back silently: the decoder
still needs the GPU library.
"""
from __future__ import annotations

import itertools
from typing import Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

EMBED_DIMS = (64, 128, 160, 320)
DEPTHS = (2, 2, 6, 2)
NUM_HEADS = (2, 4, 5, 10)
WINDOW_SIZES = (7, 7, 14, 7)
IMG_SIZE, GRID, PROMPT_DIM = 1024, 64, 256
PIXEL_MEAN = (123.675, 116.28, 103.53)
PIXEL_STD = (58.395, 57.12, 57.375)


class LayerNorm2d(nn.Module):
    def __init__(self, num_channels: int, eps: float = 1e-6) -> None:
        super().__init__()
        self.weight = nn.Parameter(torch.ones(num_channels))
        self.bias = nn.Parameter(torch.zeros(num_channels))
        self.eps = eps

    def forward(self, x):
        u = x.mean(1, keepdim=True)
        s = (x - u).pow(2).mean(1, keepdim=True)
        return self.weight[:, None, None] * ((x - u) / torch.sqrt(s + self.eps)) + self.bias[:, None, None]


class Conv2d_BN(nn.Sequential):
    def __init__(self, a, b, ks=1, stride=1, pad=0, groups=1) -> None:
        super().__init__()
        self.add_module("c", nn.Conv2d(a, b, ks, stride, pad, 1, groups, bias=False))
        self.add_module("bn", nn.BatchNorm2d(b))


class PatchEmbed(nn.Module):
    def __init__(self, in_chans: int, embed_dim: int) -> None:
        super().__init__()
        self.seq = nn.Sequential(Conv2d_BN(in_chans, embed_dim // 2, 3, 2, 1), nn.GELU(), Conv2d_BN(embed_dim // 2, embed_dim, 3, 2, 1))

    def forward(self, x):
        return self.seq(x)


class MBConv(nn.Module):
    def __init__(self, chans: int, expand_ratio: float) -> None:
        super().__init__()
        hidden = int(chans * expand_ratio)
        self.conv1, self.act1 = Conv2d_BN(chans, hidden, 1), nn.GELU()
        self.conv2, self.act2 = Conv2d_BN(hidden, hidden, 3, 1, 1, groups=hidden), nn.GELU()
        self.conv3, self.act3 = Conv2d_BN(hidden, chans, 1), nn.GELU()

    def forward(self, x):
        return self.act3(x + self.conv3(self.act2(self.conv2(self.act1(self.conv1(x))))))


class PatchMerging(nn.Module):
    def __init__(self, input_resolution: Tuple[int, int], dim: int, out_dim: int) -> None:
        super().__init__()
        self.input_resolution = input_resolution
        self.act = nn.GELU()
        self.conv1 = Conv2d_BN(dim, out_dim, 1)
        self.conv2 = Conv2d_BN(out_dim, out_dim, 3, 1 if out_dim in (320, 448, 576) else 2, 1, groups=out_dim)
        self.conv3 = Conv2d_BN(out_dim, out_dim, 1)

    def forward(self, x):
        if x.ndim == 3:
            h, w = self.input_resolution
            x = x.view(len(x), h, w, -1).permute(0, 3, 1, 2)
        x = self.conv3(self.act(self.conv2(self.act(self.conv1(x)))))
        return x.flatten(2).transpose(1, 2)


class ConvLayer(nn.Module):
    def __init__(self, dim, input_resolution, depth, out_dim, expand_ratio) -> None:
        super().__init__()
        self.blocks = nn.ModuleList([MBConv(dim, expand_ratio) for _ in range(depth)])
        self.downsample = PatchMerging(input_resolution, dim, out_dim)

    def forward(self, x):
        for blk in self.blocks:
            x = blk(x)
        return self.downsample(x)


class Mlp(nn.Module):
    def __init__(self, dim: int, hidden: int) -> None:
        super().__init__()
        self.norm = nn.LayerNorm(dim)
        self.fc1, self.fc2, self.act = nn.Linear(dim, hidden), nn.Linear(hidden, dim), nn.GELU()

    def forward(self, x):
        return self.fc2(self.act(self.fc1(self.norm(x))))


class Attention(nn.Module):
    def __init__(self, dim: int, key_dim: int, num_heads: int, resolution: Tuple[int, int]) -> None:
        super().__init__()
        self.num_heads, self.key_dim, self.scale = num_heads, key_dim, key_dim ** -0.5
        self.d = key_dim                                    # attn_ratio 1
        self.dh = key_dim * num_heads
        self.norm = nn.LayerNorm(dim)
        self.qkv = nn.Linear(dim, self.dh + 2 * key_dim * num_heads)
        self.proj = nn.Linear(self.dh, dim)
        points = list(itertools.product(range(resolution[0]), range(resolution[1])))
        offsets, idxs = {}, []
        for p1 in points:
            for p2 in points:
                off = (abs(p1[0] - p2[0]), abs(p1[1] - p2[1]))
                if off not in offsets:
                    offsets[off] = len(offsets)
                idxs.append(offsets[off])
        self.attention_biases = nn.Parameter(torch.zeros(num_heads, len(offsets)))
        self.register_buffer("attention_bias_idxs", torch.LongTensor(idxs).view(len(points), len(points)), persistent=False)

    def forward(self, x):
        B, N, _ = x.shape
        qkv = self.qkv(self.norm(x)).view(B, N, self.num_heads, -1)
        q, k, v = qkv.split([self.key_dim, self.key_dim, self.d], dim=3)
        q, k, v = q.permute(0, 2, 1, 3), k.permute(0, 2, 1, 3), v.permute(0, 2, 1, 3)
        attn = (q @ k.transpose(-2, -1)) * self.scale + self.attention_biases[:, self.attention_bias_idxs]
        x = (attn.softmax(dim=-1) @ v).transpose(1, 2).reshape(B, N, self.dh)
        return self.proj(x)


class TinyViTBlock(nn.Module):
    def __init__(self, dim, input_resolution, num_heads, window_size, mlp_ratio, local_conv_size) -> None:
        super().__init__()
        self.input_resolution, self.window_size = input_resolution, window_size
        self.attn = Attention(dim, dim // num_heads, num_heads, (window_size, window_size))
        self.mlp = Mlp(dim, int(dim * mlp_ratio))
        self.local_conv = Conv2d_BN(dim, dim, local_conv_size, 1, local_conv_size // 2, groups=dim)

    def forward(self, x):
        H, W = self.input_resolution
        B, L, C = x.shape
        ws = self.window_size
        res_x = x
        if H == ws and W == ws:
            x = self.attn(x)
        else:
            x = x.view(B, H, W, C)
            pad_b, pad_r = (ws - H % ws) % ws, (ws - W % ws) % ws
            if pad_b or pad_r:
                x = F.pad(x, (0, 0, 0, pad_r, 0, pad_b))
            pH, pW = H + pad_b, W + pad_r
            nH, nW = pH // ws, pW // ws
            x = x.view(B, nH, ws, nW, ws, C).transpose(2, 3).reshape(B * nH * nW, ws * ws, C)
            x = self.attn(x)
            x = x.view(B, nH, nW, ws, ws, C).transpose(2, 3).reshape(B, pH, pW, C)
            if pad_b or pad_r:
                x = x[:, :H, :W].contiguous()
            x = x.view(B, L, C)
        x = res_x + x
        x = self.local_conv(x.transpose(1, 2).reshape(B, C, H, W)).view(B, C, L).transpose(1, 2)
        return x + self.mlp(x)


class BasicLayer(nn.Module):
    def __init__(self, dim, input_resolution, depth, num_heads, window_size, mlp_ratio, local_conv_size, out_dim, downsample) -> None:
        super().__init__()
        self.blocks = nn.ModuleList([TinyViTBlock(dim, input_resolution, num_heads, window_size, mlp_ratio, local_conv_size)
                                     for _ in range(depth)])
        self.downsample = PatchMerging(input_resolution, dim, out_dim) if downsample else None

    def forward(self, x):
        for blk in self.blocks:
            x = blk(x)
        return x if self.downsample is None else self.downsample(x)


class TinyViT(nn.Module):
    """``mobile_sam.modeling.TinyViT(img_size=1024, embed_dims=[64, 128, 160, 320], depths=[2, 2, 6, 2], num_heads=[2, 4, 5, 10],
    window_sizes=[7, 7, 14, 7], mlp_ratio=4, mbconv_expand_ratio=4, local_conv_size=3)``."""

    def __init__(self, num_classes: int = 1000) -> None:
        super().__init__()
        self.img_size = IMG_SIZE
        self.patch_embed = PatchEmbed(3, EMBED_DIMS[0])
        res = IMG_SIZE // 4
        self.layers = nn.ModuleList()
        for i in range(4):
            r = res // (2 ** (i - 1 if i == 3 else i))
            out_dim = EMBED_DIMS[min(i + 1, 3)]
            if i == 0:
                self.layers.append(ConvLayer(EMBED_DIMS[0], (r, r), DEPTHS[0], out_dim, 4.0))
            else:
                self.layers.append(BasicLayer(EMBED_DIMS[i], (r, r), DEPTHS[i], NUM_HEADS[i], WINDOW_SIZES[i], 4.0, 3, out_dim, i < 3))
        self.norm_head = nn.LayerNorm(EMBED_DIMS[-1])
        self.head = nn.Linear(EMBED_DIMS[-1], num_classes)
        self.neck = nn.Sequential(nn.Conv2d(EMBED_DIMS[-1], PROMPT_DIM, kernel_size=1, bias=False), LayerNorm2d(PROMPT_DIM),
                                  nn.Conv2d(PROMPT_DIM, PROMPT_DIM, kernel_size=3, padding=1, bias=False), LayerNorm2d(PROMPT_DIM))
        self.precision = "fp32"

    # -- the encoder interface the rest of the package uses (modeling.ImageEncoderViT)
    def invalidate(self) -> None:
        pass

    def set_precision(self, precision: str) -> None:
        if precision not in ("bf16", "fp32"):
            raise ValueError("vit_t: the TinyViT encoder runs torch fp32 operators (no 16-bit / fp8 kernels); 'bf16' is accepted as the "
                             "package default and ignored")

    def set_split_io(self, on: bool) -> None:
        pass

    def _run(self, x: torch.Tensor) -> torch.Tensor:
        assert x.dim() == 4 and x.shape[1:] == (3, IMG_SIZE, IMG_SIZE), x.shape
        x = x.to(device=self.neck[0].weight.device, dtype=torch.float32)
        x = self.patch_embed(x)
        for layer in self.layers:
            x = layer(x)
        B, _, C = x.shape
        return self.neck(x.view(B, GRID, GRID, C).permute(0, 3, 1, 2))

    @torch.no_grad()
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """Inference: no tape, and the BatchNorms ALWAYS normalise with their running statistics - also when a trainer has put the
        model into train() mode with this encoder frozen (a no_grad forward in train mode would normalise with batch statistics and
        overwrite the checkpoint's running_mean / running_var on every step)."""
        bns = [m for m in self.modules() if isinstance(m, nn.BatchNorm2d) and m.training]
        for m in bns:
            m.training = False
        try:
            return self._run(x)
        finally:
            for m in bns:
                m.training = True

    def forward_taped(self, x: torch.Tensor) -> torch.Tensor:
        """The same operators under autograd (fine-tuning the encoder, reference ``TrainableSAM.image_embeddings_oft`` on mobile_sam's
        ``TinyViT``): BatchNorm behaves as ``self.training`` says - batch statistics + running-statistics update under ``train()``,
        exactly what torch does in the reference."""
        return self._run(x)

    @torch.no_grad()
    def forward_u8(self, images: torch.Tensor) -> torch.Tensor:
        """uint8 HWC batch [B, h, w, 3] (after ``ResizeLongestSide.apply_image``): ``Sam.preprocess`` (normalise, zero-pad to 1024^2)
        then ``forward``."""
        assert images.dtype == torch.uint8 and images.dim() == 4 and images.shape[-1] == 3, images.shape
        dev = self.neck[0].weight.device
        x = images.to(dev).permute(0, 3, 1, 2).float()
        mean = torch.tensor(PIXEL_MEAN, device=dev).view(1, 3, 1, 1)
        std = torch.tensor(PIXEL_STD, device=dev).view(1, 3, 1, 1)
        x = (x - mean) / std
        h, w = x.shape[-2:]
        return self.forward(F.pad(x, (0, IMG_SIZE - w, 0, IMG_SIZE - h)))
