"""CPU oracle for the MobileSAM image encoder (TinyViT-5M, ``model_type="vit_t"``).  TEST INFRASTRUCTURE ONLY.

micro_sam builds vit_t through ``mobile_sam.sam_model_registry`` (``micro_sam/util.py:35-43,435-439``; dependency
``git+https://github.com/ChaoningZhang/MobileSAM.git`` HEAD, ``environment.yaml:33-34`` - un-vendored and absent here, as is every other
TinyViT implementation: PARITY UNPINNED beyond the published architecture).  This file restates ``mobile_sam/modeling/tiny_vit_sam.py`` as
published, functionally over a state dict with MobileSAM's parameter names (``image_encoder.`` prefix):

* ``patch_embed.seq``: Conv2d_BN(3, 32, 3, 2, 1) - GELU - Conv2d_BN(32, 64, 3, 2, 1)  (Conv2d_BN = conv ``c`` without bias + BatchNorm
  ``bn``, evaluated with its running statistics, eps 1e-5): 1024 -> 256;
* ``layers.0``: ConvLayer - 2 x MBConv(64, expand 4: 1x1 -> GELU -> depthwise 3x3 -> GELU -> 1x1, + shortcut, GELU) and
  ``downsample`` = PatchMerging(64 -> 128: 1x1 -> GELU -> depthwise 3x3 stride 2 -> GELU -> 1x1): 256 -> 128, tokens [B, HW, C];
* ``layers.1..3``: BasicLayer - TinyViTBlocks (dims 128 / 160 / 320, depths 2 / 6 / 2, heads 4 / 5 / 10, windows 7 / 14 / 7, head dim
  32): window attention with a learned bias per |dy|, |dx| offset (``attention_biases`` [heads, offsets], windows zero-padded, no
  masking), residual, depthwise 3x3 ``local_conv`` (Conv2d_BN), MLP (LayerNorm - fc1 - GELU - fc2) with residual; ``downsample``
  PatchMerging 128 -> 160 with stride 2 (128 -> 64) and 160 -> 320 with stride 1 (out_dim 320 keeps the 64 x 64 grid);
* ``neck``: conv1x1(320 -> 256) - LayerNorm2d - conv3x3 - LayerNorm2d, as in the ViT encoders -> [B, 256, 64, 64].

Hyper-parameters: ``mobile_sam/build_sam.py`` build_sam_vit_t (embed_dims [64, 128, 160, 320], depths [2, 2, 6, 2], num_heads [2, 4, 5, 10],
window_sizes [7, 7, 14, 7], mlp_ratio 4, mbconv_expand_ratio 4, local_conv_size 3).  The classification head of TinyViT (``norm_head``,
``head``: 1000 classes) is in MobileSAM checkpoints but not on this path.  Pinned only by the published figures: 5.78 M encoder parameters
without the head (MobileSAM paper, table 3), output [1, 256, 64, 64] (tests/test_vit_t_host.py).
"""
from __future__ import annotations

import itertools
from typing import Dict

import torch
import torch.nn.functional as F

Tensor = torch.Tensor

EMBED_DIMS = (64, 128, 160, 320)
DEPTHS = (2, 2, 6, 2)
NUM_HEADS = (2, 4, 5, 10)
WINDOWS = (7, 7, 14, 7)
RESOLUTIONS = (256, 128, 64, 64)         # token grid at the input of layer i (img 1024 / 4, then the PatchMergings)
BN_EPS = 1e-5


def conv_bn(sd: Dict[str, Tensor], pre: str, x: Tensor, stride: int = 1, pad: int = 0, groups: int = 1) -> Tensor:
    x = F.conv2d(x, sd[pre + "c.weight"], None, stride, pad, 1, groups)
    return F.batch_norm(x, sd[pre + "bn.running_mean"], sd[pre + "bn.running_var"], sd[pre + "bn.weight"], sd[pre + "bn.bias"], False, 0.0,
                        BN_EPS)


def mbconv(sd, pre: str, x: Tensor) -> Tensor:
    y = F.gelu(conv_bn(sd, pre + "conv1.", x))
    y = F.gelu(conv_bn(sd, pre + "conv2.", y, 1, 1, y.shape[1]))
    y = conv_bn(sd, pre + "conv3.", y)
    return F.gelu(x + y)


def patch_merging(sd, pre: str, x: Tensor, res: int, out_dim: int) -> Tensor:
    if x.dim() == 3:
        x = x.view(x.shape[0], res, res, -1).permute(0, 3, 1, 2)
    x = F.gelu(conv_bn(sd, pre + "conv1.", x))
    stride = 1 if out_dim in (320, 448, 576) else 2
    x = F.gelu(conv_bn(sd, pre + "conv2.", x, stride, 1, out_dim))
    x = conv_bn(sd, pre + "conv3.", x)
    return x.flatten(2).transpose(1, 2)


def attention_bias_idxs(ws: int) -> Tensor:
    points = list(itertools.product(range(ws), range(ws)))
    offsets, idxs = {}, []
    for p1 in points:
        for p2 in points:
            off = (abs(p1[0] - p2[0]), abs(p1[1] - p2[1]))
            if off not in offsets:
                offsets[off] = len(offsets)
            idxs.append(offsets[off])
    return torch.tensor(idxs, dtype=torch.long).view(len(points), len(points))


def window_attention(sd, pre: str, x: Tensor, heads: int, ws: int) -> Tensor:
    """x [B', N = ws*ws, C]: LayerNorm - qkv - softmax(q k^T / sqrt(d) + bias[|dy|, |dx|]) v - proj (attn_ratio 1: d = key_dim)."""
    Bn, N, C = x.shape
    kd = C // heads
    y = F.layer_norm(x, (C,), sd[pre + "norm.weight"], sd[pre + "norm.bias"], 1e-5)
    qkv = F.linear(y, sd[pre + "qkv.weight"], sd[pre + "qkv.bias"]).view(Bn, N, heads, 3 * kd)
    q, k, v = (t.permute(0, 2, 1, 3) for t in qkv.split([kd, kd, kd], dim=3))
    bias = sd[pre + "attention_biases"][:, attention_bias_idxs(ws).to(x.device)]
    attn = ((q @ k.transpose(-2, -1)) * kd ** -0.5 + bias).softmax(dim=-1)
    out = (attn @ v).transpose(1, 2).reshape(Bn, N, C)
    return F.linear(out, sd[pre + "proj.weight"], sd[pre + "proj.bias"])


def tinyvit_block(sd, pre: str, x: Tensor, res: int, heads: int, ws: int) -> Tensor:
    B, L, C = x.shape
    shortcut = x
    if res == ws:
        x = window_attention(sd, pre + "attn.", x, heads, ws)
    else:
        x = x.view(B, res, res, C)
        pad = (ws - res % ws) % ws
        if pad:
            x = F.pad(x, (0, 0, 0, pad, 0, pad))
        p = res + pad
        n = p // ws
        x = x.view(B, n, ws, n, ws, C).transpose(2, 3).reshape(B * n * n, ws * ws, C)
        x = window_attention(sd, pre + "attn.", x, heads, ws)
        x = x.view(B, n, n, ws, ws, C).transpose(2, 3).reshape(B, p, p, C)
        if pad:
            x = x[:, :res, :res].contiguous()
        x = x.view(B, L, C)
    x = shortcut + x
    x = x.transpose(1, 2).reshape(B, C, res, res)
    x = conv_bn(sd, pre + "local_conv.", x, 1, 1, C)
    x = x.view(B, C, L).transpose(1, 2)
    y = F.layer_norm(x, (C,), sd[pre + "mlp.norm.weight"], sd[pre + "mlp.norm.bias"], 1e-5)
    y = F.linear(F.gelu(F.linear(y, sd[pre + "mlp.fc1.weight"], sd[pre + "mlp.fc1.bias"])), sd[pre + "mlp.fc2.weight"], sd[pre + "mlp.fc2.bias"])
    return x + y


def layer_norm_2d(x: Tensor, w: Tensor, b: Tensor, eps: float = 1e-6) -> Tensor:
    u = x.mean(1, keepdim=True)
    s = (x - u).pow(2).mean(1, keepdim=True)
    return w[:, None, None] * ((x - u) / torch.sqrt(s + eps)) + b[:, None, None]


@torch.no_grad()
def image_encoder(sd: Dict[str, Tensor], x: Tensor) -> Tensor:
    """[B, 3, 1024, 1024] normalised (``Sam.preprocess``) -> [B, 256, 64, 64]; keys prefixed ``image_encoder.``."""
    sd = {k[len("image_encoder."):]: v for k, v in sd.items() if k.startswith("image_encoder.")}
    x = conv_bn(sd, "patch_embed.seq.0.", x, 2, 1)
    x = conv_bn(sd, "patch_embed.seq.2.", F.gelu(x), 2, 1)
    for b in range(DEPTHS[0]):
        x = mbconv(sd, f"layers.0.blocks.{b}.", x)
    x = patch_merging(sd, "layers.0.downsample.", x, RESOLUTIONS[0], EMBED_DIMS[1])
    for i in (1, 2, 3):
        for b in range(DEPTHS[i]):
            x = tinyvit_block(sd, f"layers.{i}.blocks.{b}.", x, RESOLUTIONS[i], NUM_HEADS[i], WINDOWS[i])
        if i < 3:
            x = patch_merging(sd, f"layers.{i}.downsample.", x, RESOLUTIONS[i], EMBED_DIMS[i + 1])
    B, _, C = x.shape
    x = x.view(B, 64, 64, C).permute(0, 3, 1, 2)
    x = F.conv2d(x, sd["neck.0.weight"])
    x = layer_norm_2d(x, sd["neck.1.weight"], sd["neck.1.bias"])
    x = F.conv2d(x, sd["neck.2.weight"], padding=1)
    return layer_norm_2d(x, sd["neck.3.weight"], sd["neck.3.bias"])
