"""Differentiable forward passes of the image encoder and the prompt encoder for un-frozen fine-tuning (reference
``micro_sam/training/trainable_sam.py:71-81`` ``image_embeddings_oft`` under autograd and ``:96-99`` the prompt encoder; the
reference trains the whole SAM by default, ``micro_sam/training/util.py get_trainable_sam_model(freeze=None)``).

Same construction as ``trainable_sam.mask_decoder_forward``: torch owns the tape, shapes and the elementwise glue (residual
adds, GELU, window partition, the im2col gather of the neck); every matrix product runs ``functional.linear`` (MFMA GEMM in both
directions), every LayerNorm ``functional.layer_norm``, the attention with its decomposed relative position bias
``functional.relpos_attention`` (``msam_relpos_attention_forward / _backward``).  The two bias tensors of that attention,
``q . R_h`` and ``q . R_w`` (64 x 64 or 14 x 14 tables per head: < 0.1 % of the block's flops), are formed by a torch
``einsum`` so that autograd carries their gradients back to the queries and to the ``rel_pos_h / rel_pos_w`` parameters.

The arithmetic follows ``oracle/sam_ref.image_encoder`` / ``prompt_encoder`` in their fp32 form with bf16 GEMM operands (what
``functional.linear`` does; the reference fine-tunes under AMP bf16).

Round 3: the attention takes ``functional.relpos_attention_auto`` - by default the two products as library batched GEMMs on bf16
operands with fp32 scores / softmax (``functional.RELPOS_ATTENTION_IMPL``), the hand-written fp32 kernels as the checked reference.
The composition is checked on the CPU against the oracle's autograd with the primitives replaced by torch stand-ins
(tests/test_training_encoders_host.py) and on the GPU against the oracle (tests/test_gpu_training_encoders.py).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.nn.functional as F

from ..modeling import GRID, IMG_SIZE, PATCH, PROMPT_DIM
from . import functional as HF


def _rel_pos_table(rel_pos: torch.Tensor, size: int) -> torch.Tensor:
    """``get_rel_pos(size, size, rel_pos)`` -> [size, size, head_dim] (q index, k index); differentiable in ``rel_pos``."""
    want = 2 * size - 1
    if rel_pos.shape[0] != want:
        rel_pos = F.interpolate(rel_pos.reshape(1, rel_pos.shape[0], -1).permute(0, 2, 1), size=want, mode="linear")
        rel_pos = rel_pos.reshape(-1, want).permute(1, 0)
    idx = torch.arange(size, device=rel_pos.device)
    return rel_pos[(idx[:, None] - idx[None, :]) + (size - 1)]


def _qkv_projection(attn, x: torch.Tensor) -> torch.Tensor:
    """``attn.qkv(x)``; with LoRA surgery (``models.peft_sam.AttentionLoRA``, reference models/peft_sam.py:81-95) the frozen
    projection plus the low-rank branches ``alpha * B(A(x))`` on the q / k / v column blocks - the branches stay separate
    products here so that A and B receive gradients (inference merges them into the weight)."""
    mod = attn.qkv
    if not hasattr(mod, "qkv_proj"):
        return HF.linear(x, mod.weight, mod.bias)
    out = HF.linear(x, mod.qkv_proj.weight, mod.qkv_proj.bias)
    branches = []
    for m in ("q", "k", "v"):
        if hasattr(mod, f"w_a_linear_{m}"):
            low = HF.linear(x, getattr(mod, f"w_a_linear_{m}").weight, None)
            branches.append(mod.alpha * HF.linear(low, getattr(mod, f"w_b_linear_{m}").weight, None))
        else:
            branches.append(torch.zeros(x.shape[:-1] + (mod.dim,), dtype=out.dtype, device=out.device))
    return out + torch.cat(branches, dim=-1)


def _mlp(mlp, y: torch.Tensor) -> torch.Tensor:
    """``MLPBlock.forward``; with ``MLPLoRA`` (reference models/peft_sam.py:127-131) both layers carry a low-rank branch."""
    if not hasattr(mlp, "mlp_layer"):
        return HF.linear(F.gelu(HF.linear(y, mlp.lin1.weight, mlp.lin1.bias)), mlp.lin2.weight, mlp.lin2.bias)
    base = mlp.mlp_layer
    h = HF.linear(y, base.lin1.weight, base.lin1.bias) + HF.linear(HF.linear(y, mlp.w_a_linear_1.weight, None), mlp.w_b_linear_1.weight, None)
    h = F.gelu(h)
    return HF.linear(h, base.lin2.weight, base.lin2.bias) + HF.linear(HF.linear(h, mlp.w_a_linear_2.weight, None), mlp.w_b_linear_2.weight, None)


def _attention(attn, x: torch.Tensor) -> torch.Tensor:
    """Upstream ``Attention.forward`` on a [B', S, S, C] grid (a batch of windows or of whole images)."""
    Bp, S, _, C = x.shape
    heads = attn.num_heads
    hd = C // heads
    N = S * S
    qkv = _qkv_projection(attn, x.reshape(Bp, N, C))
    q, k, v = qkv.reshape(Bp, N, 3, heads, hd).permute(2, 0, 3, 1, 4).reshape(3, Bp * heads, N, hd).unbind(0)
    r_q = q.reshape(Bp * heads, S, S, hd)                                              # unscaled queries (upstream behaviour)
    bias_h = torch.einsum("bhwc,hkc->bhwk", r_q, _rel_pos_table(attn.rel_pos_h, S)).reshape(Bp * heads, N, S)
    bias_w = torch.einsum("bhwc,wkc->bhwk", r_q, _rel_pos_table(attn.rel_pos_w, S)).reshape(Bp * heads, N, S)
    o = HF.relpos_attention_auto(q, k, v, bias_h, bias_w, attn.scale)
    o = o.reshape(Bp, heads, S, S, hd).permute(0, 2, 3, 1, 4).reshape(Bp, S, S, C)
    return HF.linear(o, attn.proj.weight, attn.proj.bias)


def _window_partition(x: torch.Tensor, ws: int):
    B, H, W, C = x.shape
    pad_h, pad_w = (ws - H % ws) % ws, (ws - W % ws) % ws
    if pad_h or pad_w:
        x = F.pad(x, (0, 0, 0, pad_w, 0, pad_h))
    Hp, Wp = H + pad_h, W + pad_w
    x = x.reshape(B, Hp // ws, ws, Wp // ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws, ws, C)
    return x, (Hp, Wp)


def _window_unpartition(windows: torch.Tensor, ws: int, pad_hw, hw) -> torch.Tensor:
    (Hp, Wp), (H, W) = pad_hw, hw
    B = windows.shape[0] // (Hp * Wp // ws // ws)
    x = windows.reshape(B, Hp // ws, Wp // ws, ws, ws, -1).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, -1)
    return x[:, :H, :W, :]


def image_encoder_forward(enc, x: torch.Tensor) -> torch.Tensor:
    """``ImageEncoderViT.forward`` with a tape: [B, 3, 1024, 1024] normalised + padded images -> [B, 256, 64, 64]."""
    B = x.shape[0]
    D = enc.embed_dim
    # patch embedding: a 16 x 16 / 16 convolution = one linear map per patch, columns ordered (c, ky, kx) like the weight
    patches = x.reshape(B, 3, GRID, PATCH, GRID, PATCH).permute(0, 2, 4, 1, 3, 5).reshape(B, GRID, GRID, 3 * PATCH * PATCH)
    w = enc.patch_embed.proj.weight
    x = HF.linear(patches, w.reshape(w.shape[0], -1), enc.patch_embed.proj.bias) + enc.pos_embed
    for blk in enc.blocks:
        shortcut = x
        y = HF.layer_norm(x, blk.norm1.weight, blk.norm1.bias, blk.norm1.eps)
        if blk.window_size > 0:
            H, W = y.shape[1:3]
            y, pad_hw = _window_partition(y, blk.window_size)
            y = _window_unpartition(_attention(blk.attn, y), blk.window_size, pad_hw, (H, W))
        else:
            y = _attention(blk.attn, y)
        x = shortcut + y
        y = HF.layer_norm(x, blk.norm2.weight, blk.norm2.bias, blk.norm2.eps)
        x = x + _mlp(blk.mlp, y)
    conv1, ln1, conv3, ln2 = enc.neck[0], enc.neck[1], enc.neck[2], enc.neck[3]
    y = HF.linear(x, conv1.weight.reshape(PROMPT_DIM, D), None)                         # 1 x 1 convolution
    y = HF.layer_norm(y, ln1.weight, ln1.bias, ln1.eps)                                 # LayerNorm2d = LayerNorm over the channels
    cols = F.unfold(y.permute(0, 3, 1, 2), kernel_size=3, padding=1).transpose(1, 2)    # [B, 4096, 256 * 9], columns (c, ky, kx)
    y = HF.linear(cols, conv3.weight.reshape(PROMPT_DIM, -1), None).reshape(B, GRID, GRID, PROMPT_DIM)
    y = HF.layer_norm(y, ln2.weight, ln2.bias, ln2.eps)
    return y.permute(0, 3, 1, 2)


def _pe_encoding(gauss: torch.Tensor, coords: torch.Tensor) -> torch.Tensor:
    """``PositionEmbeddingRandom._pe_encoding`` for coordinates in [0, 1] (the Gaussian matrix is a buffer, not trained)."""
    c = (2 * coords - 1) @ gauss
    c = 2 * torch.pi * c
    return torch.cat([torch.sin(c), torch.cos(c)], dim=-1)


def prompt_encoder_forward(pe, points: Optional[Tuple[torch.Tensor, torch.Tensor]], boxes: Optional[torch.Tensor],
                           masks: Optional[torch.Tensor]) -> Tuple[torch.Tensor, torch.Tensor]:
    """``PromptEncoder.forward`` with a tape -> (sparse [B, N, 256], dense [B, 256, 64, 64]).  The learned quantities here
    are six 256-vectors and the three tiny convolutions of ``mask_downscaling``: embedding sums, LayerNorm2d over 4 / 16
    channels and GELU are torch glue, the convolutions run ``functional.linear`` on their 2 x 2 patches."""
    gauss = pe.pe_layer.positional_encoding_gaussian_matrix
    dev = gauss.device
    bs = points[0].shape[0] if points is not None else (boxes.shape[0] if boxes is not None else (masks.shape[0] if masks is not None else 1))
    parts = []
    if points is not None:
        coords, labels = points[0].to(dev, torch.float32) + 0.5, points[1].to(dev)
        if boxes is None:
            coords = torch.cat([coords, torch.zeros((bs, 1, 2), device=dev)], dim=1)
            labels = torch.cat([labels, -torch.ones((bs, 1), device=dev, dtype=labels.dtype)], dim=1)
        e = _pe_encoding(gauss, coords / IMG_SIZE)
        pad, neg, pos = (labels == -1)[..., None], (labels == 0)[..., None], (labels == 1)[..., None]
        e = torch.where(pad, torch.zeros_like(e), e)
        e = e + pad * pe.not_a_point_embed.weight + neg * pe.point_embeddings[0].weight + pos * pe.point_embeddings[1].weight
        parts.append(e)
    if boxes is not None:
        c = (boxes.to(dev, torch.float32) + 0.5).reshape(-1, 2, 2) / IMG_SIZE
        e = _pe_encoding(gauss, c)
        corner = torch.stack([pe.point_embeddings[2].weight[0], pe.point_embeddings[3].weight[0]])
        parts.append(e + corner[None])
    sparse = torch.cat(parts, dim=1) if parts else torch.empty((bs, 0, PROMPT_DIM), device=dev)
    if masks is None:
        dense = pe.no_mask_embed.weight.reshape(1, -1, 1, 1).expand(bs, -1, GRID, GRID)
        return sparse, dense
    md = pe.mask_downscaling

    def conv2x2(t, conv):                       # [B, H, W, Cin] -> [B, H/2, W/2, Cout]: Conv2d(kernel 2, stride 2) on its patches
        b, h, w_, cin = t.shape
        p = t.reshape(b, h // 2, 2, w_ // 2, 2, cin).permute(0, 1, 3, 5, 2, 4).reshape(b, h // 2, w_ // 2, cin * 4)   # (c, ky, kx)
        return HF.linear(p, conv.weight.reshape(conv.weight.shape[0], -1), conv.bias)
    m = masks.to(dev, torch.float32).permute(0, 2, 3, 1)
    m = F.gelu(F.layer_norm(conv2x2(m, md[0]), (md[1].weight.shape[0],), md[1].weight, md[1].bias, md[1].eps))
    m = F.gelu(F.layer_norm(conv2x2(m, md[3]), (md[4].weight.shape[0],), md[4].weight, md[4].bias, md[4].eps))
    dense = HF.linear(m, md[6].weight.reshape(PROMPT_DIM, -1), md[6].bias).permute(0, 3, 1, 2)
    return sparse, dense
