"""SAM model objects behind micro_sam's ``SamPredictor`` boundary, computed by libmsam_hip.so.

The module tree and parameter names are those of upstream ``segment_anything`` (the names micro_sam depends
on: ``micro_sam/models/build_sam.py:26``, ``micro_sam/util.py:457,578-598``, ``micro_sam/models/peft_sam.py:41-44``,
``micro_sam/instance_segmentation.py:774-783``), so real checkpoints ``load_state_dict``
unchanged and ``blocks[i].attn.qkv`` etc. stay ordinary ``nn.Module``s.  The ``forward`` passes do not run torch
ops: they hand device pointers of cached bf16/fp32 weight copies to the C ABI (``include/msam_hip.h``).

Hyper-parameters: ``micro_sam/models/build_sam.py:40-142``.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib

VIT_CONFIGS = {
    # micro_sam/models/build_sam.py:40-84
    "vit_b": dict(embed_dim=768, depth=12, num_heads=12, global_attn_indexes=(2, 5, 8, 11)),
    "vit_l": dict(embed_dim=1024, depth=24, num_heads=16, global_attn_indexes=(5, 11, 17, 23)),
    "vit_h": dict(embed_dim=1280, depth=32, num_heads=16, global_attn_indexes=(7, 15, 23, 31)),
}
IMG_SIZE, PATCH, GRID, WINDOW, PROMPT_DIM = 1024, 16, 64, 14, 256


def _bf16(t: torch.Tensor) -> torch.Tensor:
    return t.detach().to(torch.bfloat16).contiguous()


def _d16(t: torch.Tensor) -> torch.Tensor:
    """Decoder weights in the decoder's 16-bit type of this library build (fp16 by default, see csrc/common.h)."""
    return t.detach().to(_lib.decoder_dtype()).contiguous()


class _ParamWatch:
    """Identity of the parameter set behind cached 16-bit operand copies: per parameter the object, its in-place version counter
    (optimizer step, ``copy_``, ``load_state_dict``), its storage address and its device - so a replaced ``Parameter``
    (``module.weight = nn.Parameter(...)`` updates the module's ``_parameters`` dict, which is what is read here), a
    ``param.data = ...`` assignment (``module.to()`` / ``half()`` keep the old counter) and a move to another device rebuild the
    copies as well.  The (dict, name) slots are collected once per module tree - walking ``parameters()`` on every
    call cost 0.4 ms per decode (measured, round 4) - and again after ``invalidate()``, which is also what code that ADDS modules or
    writes through ``p.data`` in place (``p.data.mul_()``, an EMA) has to call."""

    def __init__(self, *modules) -> None:
        self._modules = modules
        self._slots = None

    def reset(self) -> None:
        self._slots = None

    def key(self) -> int:
        if self._slots is None:
            self._slots = [(m._parameters, name) for root in self._modules for m in root.modules() for name in m._parameters]
        acc = []
        for d, name in self._slots:
            q = d.get(name)
            if q is not None:
                acc.append((id(q), q._version, q.data_ptr(), q.device.index))
        return hash(tuple(acc))


def _split16(w: torch.Tensor) -> torch.Tensor:
    """Weight [N, K] as hi + lo pairs of the decoder's 16-bit type, rows [Whi | Whi | Wlo] (against activation rows [hi | lo | hi])."""
    w = w.detach().to(torch.float32)
    hi = w.to(_lib.decoder_dtype())
    lo = (w - hi.to(torch.float32)).to(_lib.decoder_dtype())
    return torch.cat([hi, hi, lo], dim=1).contiguous()


def _f32(t: torch.Tensor) -> torch.Tensor:
    return t.detach().to(torch.float32).contiguous()


class LayerNorm2d(nn.Module):
    def __init__(self, num_channels: int, eps: float = 1e-6) -> None:
        super().__init__()
        self.weight = nn.Parameter(torch.ones(num_channels))
        self.bias = nn.Parameter(torch.zeros(num_channels))
        self.eps = eps


class MLPBlock(nn.Module):
    def __init__(self, dim: int, mlp_dim: int, act=nn.GELU) -> None:
        super().__init__()
        self.lin1 = nn.Linear(dim, mlp_dim)
        self.lin2 = nn.Linear(mlp_dim, dim)
        self.act = act()


class _ViTAttention(nn.Module):
    def __init__(self, dim: int, num_heads: int, input_size: int) -> None:
        super().__init__()
        self.num_heads = num_heads
        head_dim = dim // num_heads
        self.scale = head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.proj = nn.Linear(dim, dim)
        self.use_rel_pos = True
        self.rel_pos_h = nn.Parameter(torch.zeros(2 * input_size - 1, head_dim))
        self.rel_pos_w = nn.Parameter(torch.zeros(2 * input_size - 1, head_dim))


class _ViTBlock(nn.Module):
    def __init__(self, dim: int, num_heads: int, window_size: int) -> None:
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = _ViTAttention(dim, num_heads, GRID if window_size == 0 else window_size)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = MLPBlock(dim, dim * 4)
        self.window_size = window_size


class _PatchEmbed(nn.Module):
    def __init__(self, embed_dim: int) -> None:
        super().__init__()
        self.proj = nn.Conv2d(3, embed_dim, kernel_size=PATCH, stride=PATCH)


def _resize_rel_pos(rel_pos: torch.Tensor, size: int) -> torch.Tensor:
    """get_rel_pos' table interpolation for checkpoints whose table length differs from 2S-1."""
    want = 2 * size - 1
    if rel_pos.shape[0] == want:
        return rel_pos
    r = F.interpolate(rel_pos.float().reshape(1, rel_pos.shape[0], -1).permute(0, 2, 1), size=want, mode="linear")
    return r.reshape(-1, want).permute(1, 0)


class ImageEncoderViT(nn.Module):
    """``predictor.model.image_encoder``: [B,3,1024,1024] fp32 (normalised, padded) -> [B,256,64,64] fp32."""

    def __init__(self, embed_dim: int, depth: int, num_heads: int, global_attn_indexes: Tuple[int, ...]) -> None:
        super().__init__()
        self.img_size = IMG_SIZE
        self.embed_dim, self.depth, self.num_heads = embed_dim, depth, num_heads
        self.global_attn_indexes = tuple(global_attn_indexes)
        self.patch_embed = _PatchEmbed(embed_dim)
        self.pos_embed = nn.Parameter(torch.zeros(1, GRID, GRID, embed_dim))
        self.blocks = nn.ModuleList(
            [_ViTBlock(embed_dim, num_heads, 0 if i in self.global_attn_indexes else WINDOW) for i in range(depth)])
        self.neck = nn.Sequential(
            nn.Conv2d(embed_dim, PROMPT_DIM, kernel_size=1, bias=False), LayerNorm2d(PROMPT_DIM),
            nn.Conv2d(PROMPT_DIM, PROMPT_DIM, kernel_size=3, padding=1, bias=False), LayerNorm2d(PROMPT_DIM))
        self.use_glds = 0
        self.precision = "bf16"          # "fp8": qkv / proj / lin1 / lin2 on e4m3 operands (set_precision; BASELINE config 5)
        # patch embedding + neck (1.2 % of the flops) on hi + lo operand pairs: these two sites carry most of what the 8-bit operand
        # rounding costs in mask parity (profiles/r03_enc_ablation.txt); set_split_io(False) = plain 16-bit
        self.split_io = True
        self._prep = None
        self._watch = _ParamWatch(self)
        self._workspace = None
        self.register_load_state_dict_post_hook(lambda m, k: m.invalidate())

    def invalidate(self) -> None:
        self._prep = None
        self._watch.reset()

    def set_precision(self, precision: str) -> None:
        """"bf16" (default: every matrix-product operand bf16); "fp16": every operand and stored activation IEEE fp16 instead -
        the same kernels on the fp16 MFMAs of the same rate, 11 instead of 8 significand bits (activations of the encoder are
        LayerNorm outputs, q / k / v, softmax probabilities and GELU outputs: inside fp16's range; the residual stream stays
        fp32); "fp8": the four large projections of every block take OCP e4m3 operands on the MX-scaled MFMA (per-token
        activation scales, per-output-channel weight scales; fp32 accumulation, scales applied in the GEMM epilogue);
        attention, patch embedding and neck stay bf16; "fp32": the strict mode (micro_sam_amd/strict.py) - the reference's formulation
        on fp32 kernels (f32-input MFMA products, erf GELU, expf softmax), ~1/16 of the MFMA rate, results to fp32 rounding;
        "split16": that formulation with every product on fp16 operand pairs (3 MFMAs of the 16-bit pipe per product, fp32-level accuracy)."""
        if precision not in ("bf16", "fp16", "fp8", "fp32", "split16"):
            raise ValueError(f"Invalid encoder precision {precision!r}: expect 'bf16', 'fp16', 'fp8', 'fp32' or 'split16'")
        if precision != self.precision:
            self.precision = precision
            self.invalidate()

    def set_split_io(self, on: bool) -> None:
        """hi + lo operand pairs at the patch embedding and the neck (default on); off = every operand plainly 16-bit (ablation)."""
        if bool(on) != bool(self.split_io):
            self.split_io = bool(on)
            self.invalidate()

    def _prepare(self):
        if self._prep is not None:
            if self._watch.key() == self._prep_versions:
                return self._prep
            self._prep = None       # a parameter was updated in place (optimizer step, copy_): rebuild the operand copies
        dev = self.pos_embed.device
        _lib.require_gpu(dev)
        D = self.embed_dim
        keep: List[torch.Tensor] = []

        def k(t):
            keep.append(t)
            return t.data_ptr()

        p = _lib.EncoderParams()
        p.embed_dim, p.depth, p.heads = D, self.depth, self.num_heads
        f16 = self.precision == "fp16"
        p.dtype16 = _lib.F16 if f16 else _lib.BF16

        def _bf16(t):               # the encoder's 16-bit operand type (shadows the module-level bf16 helper in this method)
            return t.detach().to(torch.float16 if f16 else torch.bfloat16).contiguous()
        # stored head_dim: the attention kernels contract over multiples of 32 channels; vit_h's 80-channel heads are
        # zero-padded to 96 (exact: padded q / k channels add 0 to every score, padded v channels give 0 outputs that meet
        # zero columns of the padded proj weight)
        H, hd = self.num_heads, D // self.num_heads
        if hd not in (64, 80):
            raise NotImplementedError(f"micro_sam_amd: head_dim {hd} is not supported (64: vit_b / vit_l, 80: vit_h)")
        hs = 64 if hd == 64 else 96
        p.head_dim_stored = hs

        def pad_rows(t):            # [3*H*hd, ...] -> [3*H*hs, ...]
            if hs == hd:
                return t
            t = t.detach().reshape(3, H, hd, -1)
            return F.pad(t, (0, 0, 0, hs - hd)).reshape(3 * H * hs, *([t.shape[-1]] if t.shape[-1] > 1 else []))

        def pad_cols(t, groups):    # [..., groups*hd] -> [..., groups*hs]
            if hs == hd:
                return t
            t = t.detach().reshape(t.shape[0], groups, hd)
            return F.pad(t, (0, hs - hd)).reshape(t.shape[0], groups * hs)
        split = bool(self.split_io)
        p.split_io = 1 if split else 0

        def _io(t):                 # weight of a split site: [Whi | Whi | Wlo] (msam_encoder_t.split_io), else the plain 16-bit copy
            if not split:
                return _bf16(t)
            t = t.detach().float()
            hi = _bf16(t)
            lo = _bf16(t - hi.float())
            return torch.cat([hi, hi, lo], dim=1).contiguous()
        p.patch_w = k(_io(self.patch_embed.proj.weight.reshape(D, 3 * PATCH * PATCH)))
        p.patch_b = k(_f32(self.patch_embed.proj.bias))
        p.pos_embed = k(_f32(self.pos_embed.reshape(GRID * GRID, D)))
        for i, blk in enumerate(self.blocks):
            size = GRID if blk.window_size == 0 else blk.window_size
            p.is_global[i] = 1 if blk.window_size == 0 else 0
            p.ln1_w[i], p.ln1_b[i] = k(_f32(blk.norm1.weight)), k(_f32(blk.norm1.bias))
            p.qkv_w[i] = k(_bf16(pad_rows(blk.attn.qkv.weight)))
            p.qkv_b[i] = k(_f32(pad_rows(blk.attn.qkv.bias.reshape(-1, 1)).reshape(-1)))
            p.rel_h[i] = k(_bf16(pad_cols(_resize_rel_pos(blk.attn.rel_pos_h, size), 1)))
            p.rel_w[i] = k(_bf16(pad_cols(_resize_rel_pos(blk.attn.rel_pos_w, size), 1)))
            p.proj_w[i], p.proj_b[i] = k(_bf16(pad_cols(blk.attn.proj.weight, H))), k(_f32(blk.attn.proj.bias))
            p.ln2_w[i], p.ln2_b[i] = k(_f32(blk.norm2.weight)), k(_f32(blk.norm2.bias))
            p.lin1_w[i], p.lin1_b[i] = k(_bf16(blk.mlp.lin1.weight)), k(_f32(blk.mlp.lin1.bias))
            p.lin2_w[i], p.lin2_b[i] = k(_bf16(blk.mlp.lin2.weight)), k(_f32(blk.mlp.lin2.bias))
            if self.precision == "fp8":
                from .ops import quant_weight_fp8
                for name, wt in (("qkv", pad_rows(blk.attn.qkv.weight)), ("proj", pad_cols(blk.attn.proj.weight, H)),
                                 ("lin1", blk.mlp.lin1.weight), ("lin2", blk.mlp.lin2.weight)):
                    w8, cs = quant_weight_fp8(wt.detach().float().cpu())          # e4m3 cast on the host (once per model)
                    getattr(p, name + "_w8")[i] = k(w8.to(dev))
                    getattr(p, name + "_cs")[i] = k(cs.to(dev))
        p.neck0_w = k(_io(self.neck[0].weight.reshape(PROMPT_DIM, D)))
        p.neck1_w, p.neck1_b = k(_f32(self.neck[1].weight)), k(_f32(self.neck[1].bias))
        p.neck2_w = k(_io(self.neck[2].weight.permute(0, 2, 3, 1).reshape(PROMPT_DIM, 9 * PROMPT_DIM)))
        p.neck3_w, p.neck3_b = k(_f32(self.neck[3].weight)), k(_f32(self.neck[3].bias))
        p.use_glds = int(self.use_glds)
        p.fp8 = 1 if self.precision == "fp8" else 0
        self._prep = (p, keep)
        self._prep_versions = self._watch.key()
        return self._prep

    def _get_workspace(self, params, B: int) -> torch.Tensor:
        need = _lib.load().msam_encoder_workspace_bytes(C.byref(params), B)
        if self._workspace is None or self._workspace.numel() < need or self._workspace.device != self.pos_embed.device:
            self._workspace = torch.empty(need, dtype=torch.uint8, device=self.pos_embed.device)
        return self._workspace

    def _strict(self):
        if getattr(self, "_strict_enc", None) is None:
            from .strict import StrictEncoder
            self._strict_enc = StrictEncoder(self)
        return self._strict_enc

    @torch.no_grad()
    def forward(self, x: torch.Tensor, tap_block: Optional[int] = None):
        assert x.dim() == 4 and x.shape[1:] == (3, IMG_SIZE, IMG_SIZE), x.shape
        if self.precision in ("fp32", "split16"):
            if tap_block is not None:
                raise NotImplementedError("micro_sam_amd: tap_block is a test hook of the 16-bit encoder")
            return self._strict().forward(x=x)
        params, _ = self._prepare()
        x = x.to(device=self.pos_embed.device, dtype=torch.float32).contiguous()
        B = x.shape[0]
        ws = self._get_workspace(params, B)
        out = torch.empty((B, PROMPT_DIM, GRID, GRID), dtype=torch.float32, device=x.device)
        tap = None
        if tap_block is not None:
            tap = torch.empty((B * GRID * GRID, self.embed_dim), dtype=torch.float32, device=x.device)
        _lib.check(_lib.load().msam_encoder_forward(
            C.byref(params), x.data_ptr(), None, IMG_SIZE, IMG_SIZE, B, out.data_ptr(), ws.data_ptr(), ws.numel(),
            _lib.ptr(tap), -1 if tap_block is None else int(tap_block), _lib.stream_ptr()), "msam_encoder_forward")
        return (out, tap) if tap_block is not None else out

    @torch.no_grad()
    def forward_u8(self, images: torch.Tensor) -> torch.Tensor:
        """uint8 HWC batch [B,h,w,3] (after ``ResizeLongestSide.apply_image``): ``Sam.preprocess`` fused on device."""
        assert images.dtype == torch.uint8 and images.dim() == 4 and images.shape[-1] == 3, images.shape
        if self.precision in ("fp32", "split16"):
            return self._strict().forward(images_u8=images)
        params, _ = self._prepare()
        images = images.to(self.pos_embed.device).contiguous()
        B, h, w = images.shape[:3]
        ws = self._get_workspace(params, B)
        out = torch.empty((B, PROMPT_DIM, GRID, GRID), dtype=torch.float32, device=images.device)
        _lib.check(_lib.load().msam_encoder_forward(
            C.byref(params), None, images.data_ptr(), h, w, B, out.data_ptr(), ws.data_ptr(), ws.numel(), None, -1,
            _lib.stream_ptr()), "msam_encoder_forward")
        return out


# ------------------------------------------------------------------------------------------------ prompt encoder

class PositionEmbeddingRandom(nn.Module):
    def __init__(self, num_pos_feats: int = 128) -> None:
        super().__init__()
        self.register_buffer("positional_encoding_gaussian_matrix", torch.randn((2, num_pos_feats)))


class PromptEncoder(nn.Module):
    """Upstream parameter names; evaluated inside the fused decoder call (``Sam.decode``) or, as a stand-alone module call
    (``sam.prompt_encoder(points, boxes, masks)``, micro_sam/training/trainable_sam.py:96-99), by ``msam_prompt_encode``."""

    def __init__(self) -> None:
        super().__init__()
        self.embed_dim = PROMPT_DIM
        self.input_image_size = (IMG_SIZE, IMG_SIZE)
        self.image_embedding_size = (GRID, GRID)
        self.pe_layer = PositionEmbeddingRandom(PROMPT_DIM // 2)
        self.point_embeddings = nn.ModuleList([nn.Embedding(1, PROMPT_DIM) for _ in range(4)])
        self.not_a_point_embed = nn.Embedding(1, PROMPT_DIM)
        self.mask_input_size = (4 * GRID, 4 * GRID)
        self.mask_downscaling = nn.Sequential(
            nn.Conv2d(1, 4, kernel_size=2, stride=2), LayerNorm2d(4), nn.GELU(),
            nn.Conv2d(4, 16, kernel_size=2, stride=2), LayerNorm2d(16), nn.GELU(),
            nn.Conv2d(16, PROMPT_DIM, kernel_size=1))
        self.no_mask_embed = nn.Embedding(1, PROMPT_DIM)
        self._dense_pe_fn = None
        self._sam_ref = None

    def get_dense_pe(self) -> torch.Tensor:
        """[1,256,64,64] dense positional encoding (computed by the decoder's constant pass)."""
        if self._dense_pe_fn is None:
            raise RuntimeError("get_dense_pe: the prompt encoder is not attached to a Sam model")
        return self._dense_pe_fn()

    @torch.no_grad()
    def forward(self, points, boxes, masks):
        """``(sparse [B, N, 256], dense [B, 256, 64, 64])`` as upstream ``PromptEncoder.forward``: points = (coords [B,Np,2] in
        the 1024 input frame, labels [B,Np]) or None, boxes [B,4] or None, masks [B,1,256,256] or None."""
        if self._sam_ref is None or self._sam_ref() is None:
            raise RuntimeError("prompt_encoder: not attached to a Sam model")
        return self._sam_ref()._prompt_encode(points, boxes, masks)


# ------------------------------------------------------------------------------------------------ mask decoder

class _DecAttention(nn.Module):
    def __init__(self, embedding_dim: int, num_heads: int, downsample_rate: int = 1) -> None:
        super().__init__()
        self.embedding_dim, self.num_heads = embedding_dim, num_heads
        self.internal_dim = embedding_dim // downsample_rate
        self.q_proj = nn.Linear(embedding_dim, self.internal_dim)
        self.k_proj = nn.Linear(embedding_dim, self.internal_dim)
        self.v_proj = nn.Linear(embedding_dim, self.internal_dim)
        self.out_proj = nn.Linear(self.internal_dim, embedding_dim)


class _TwoWayBlock(nn.Module):
    def __init__(self, skip_first_layer_pe: bool) -> None:
        super().__init__()
        self.self_attn = _DecAttention(PROMPT_DIM, 8)
        self.norm1 = nn.LayerNorm(PROMPT_DIM)
        self.cross_attn_token_to_image = _DecAttention(PROMPT_DIM, 8, downsample_rate=2)
        self.norm2 = nn.LayerNorm(PROMPT_DIM)
        self.mlp = MLPBlock(PROMPT_DIM, 2048, nn.ReLU)
        self.norm3 = nn.LayerNorm(PROMPT_DIM)
        self.norm4 = nn.LayerNorm(PROMPT_DIM)
        self.cross_attn_image_to_token = _DecAttention(PROMPT_DIM, 8, downsample_rate=2)
        self.skip_first_layer_pe = skip_first_layer_pe


class TwoWayTransformer(nn.Module):
    def __init__(self) -> None:
        super().__init__()
        self.layers = nn.ModuleList([_TwoWayBlock(i == 0) for i in range(2)])
        self.final_attn_token_to_image = _DecAttention(PROMPT_DIM, 8, downsample_rate=2)
        self.norm_final_attn = nn.LayerNorm(PROMPT_DIM)


class _MLP(nn.Module):
    def __init__(self, input_dim: int, hidden_dim: int, output_dim: int, num_layers: int) -> None:
        super().__init__()
        h = [hidden_dim] * (num_layers - 1)
        self.layers = nn.ModuleList(nn.Linear(n, k) for n, k in zip([input_dim] + h, h + [output_dim]))


class MaskDecoder(nn.Module):
    def __init__(self, num_multimask_outputs: int = 3) -> None:
        super().__init__()
        if num_multimask_outputs != 3:
            raise NotImplementedError("micro_sam_amd: only num_multimask_outputs == 3 is supported")
        self.transformer_dim = PROMPT_DIM
        self.transformer = TwoWayTransformer()
        self.num_multimask_outputs = num_multimask_outputs
        self.iou_token = nn.Embedding(1, PROMPT_DIM)
        self.num_mask_tokens = num_multimask_outputs + 1
        self.mask_tokens = nn.Embedding(self.num_mask_tokens, PROMPT_DIM)
        self.output_upscaling = nn.Sequential(
            nn.ConvTranspose2d(PROMPT_DIM, PROMPT_DIM // 4, kernel_size=2, stride=2), LayerNorm2d(PROMPT_DIM // 4),
            nn.GELU(), nn.ConvTranspose2d(PROMPT_DIM // 4, PROMPT_DIM // 8, kernel_size=2, stride=2), nn.GELU())
        self.output_hypernetworks_mlps = nn.ModuleList([_MLP(PROMPT_DIM, PROMPT_DIM, PROMPT_DIM // 8, 3)
                                                        for _ in range(self.num_mask_tokens)])
        self.iou_prediction_head = _MLP(PROMPT_DIM, 256, self.num_mask_tokens, 3)
        self._sam_ref = None

    @torch.no_grad()
    def forward(self, image_embeddings: torch.Tensor, image_pe: torch.Tensor, sparse_prompt_embeddings: torch.Tensor,
                dense_prompt_embeddings: torch.Tensor, multimask_output: bool):
        """``(low_res_masks [B, C, 256, 256], iou_predictions [B, C])`` as upstream ``MaskDecoder.forward``
        (micro_sam/training/trainable_sam.py:100-106) for ONE image embedding [1, 256, 64, 64]; ``image_pe`` must be the
        model's own ``prompt_encoder.get_dense_pe()`` (its projections are precomputed tables of the HIP decoder)."""
        if self._sam_ref is None or self._sam_ref() is None:
            raise RuntimeError("mask_decoder: not attached to a Sam model")
        return self._sam_ref()._decode_embeddings(image_embeddings, image_pe, sparse_prompt_embeddings,
                                                  dense_prompt_embeddings, multimask_output)


# ------------------------------------------------------------------------------------------------ Sam

class Sam(nn.Module):
    mask_threshold: float = 0.0
    image_format: str = "RGB"

    def __init__(self, image_encoder: ImageEncoderViT, prompt_encoder: PromptEncoder, mask_decoder: MaskDecoder,
                 pixel_mean=(123.675, 116.28, 103.53), pixel_std=(58.395, 57.12, 57.375)) -> None:
        super().__init__()
        self.image_encoder = image_encoder
        self.prompt_encoder = prompt_encoder
        self.mask_decoder = mask_decoder
        self.register_buffer("pixel_mean", torch.Tensor(pixel_mean).view(-1, 1, 1), False)
        self.register_buffer("pixel_std", torch.Tensor(pixel_std).view(-1, 1, 1), False)
        self.use_glds = 0
        self._dec = None            # (params, keep-alive tensors, consts buffer)
        # the two products of the token MLP on hi + lo operand pairs (round 4: the ReLU hidden is the decoder's most rounding-sensitive
        # tensor; 0.2 % of the decoder's flops); set_split_token_mlp(False) = the plain 16-bit operands of rounds 1 - 3
        self.split_token_mlp = True
        self.precision = "default"          # set_precision("strict"): the fp32 formulation (micro_sam_amd/strict.py)
        # type of the low-res logits between the decoder's up-scaling kernel and the fused post-processing on the AMG path
        # (predict_masks_device): torch.float32 (default), or torch.float16 = half the 1.6 GB round trip per tile for ~2 % more tiles/s -
        # NOT the default: the bilinear up-sampling interpolates between neighbours of the full logit magnitude, so fp16's 2^-11
        # relative rounding moves zero crossings as much as the whole decoder's arithmetic does (measured: 2 more of 152 instances
        # below IoU 0.999 on the bench tile, profiles/r04_experiments.md section 4).  MSAM_AMG_LOW_RES=fp16 selects it process-wide.
        import os
        self.amg_low_res_dtype = torch.float16 if os.environ.get("MSAM_AMG_LOW_RES", "fp32") == "fp16" else torch.float32
        self._watch = _ParamWatch(self.prompt_encoder, self.mask_decoder)
        self._img_state = None      # (key, buffer)
        self._dec_ws = None
        self.prompt_encoder._dense_pe_fn = self._dense_pe
        import weakref
        self.prompt_encoder._sam_ref = weakref.ref(self)
        self.mask_decoder._sam_ref = weakref.ref(self)
        self.register_load_state_dict_post_hook(lambda m, k: m.invalidate())

    @property
    def device(self) -> torch.device:
        return self.pixel_mean.device

    def invalidate(self) -> None:
        self._dec = None
        self._img_state = None
        self._watch.reset()
        self.image_encoder.invalidate()

    def set_precision(self, mode: str) -> None:
        """"default": the 16-bit throughput path (bf16 image encoder, fp16 folded / chained mask decoder; DESIGN.md sections 3, 4).
        "strict": the reference's formulation on fp32 kernels for the image encoder, the prompt encoder and the mask decoder
        (micro_sam_amd/strict.py): results equal to the reference CPU path up to fp32 rounding, at ~1/30 of the default throughput.
        The integer post-processing is the same (bit-exact) code in both modes."""
        if mode not in ("default", "strict", "split16"):
            raise ValueError(f"Invalid precision mode {mode!r}: expect 'default', 'split16' or 'strict'")
        if mode == getattr(self, "precision", "default"):
            return
        enc_now = getattr(self.image_encoder, "precision", None)
        if getattr(self, "precision", "default") == "default" and enc_now in ("bf16", "fp16", "fp8"):
            self._default_encoder_precision = enc_now                    # (a caller's fp16 / fp8 choice survives a round trip)
        self.precision = mode
        if hasattr(self.image_encoder, "set_precision"):                 # (vit_t: TinyViT runs fp32 torch operators in every mode)
            self.image_encoder.set_precision({"strict": "fp32", "split16": "split16"}.get(mode, getattr(self, "_default_encoder_precision", "bf16")))
        if getattr(self, "_mask_params", None) is not None:
            self._mask_params.exact_gelu = 0 if mode == "default" else 1 # (a field of a struct shared with the lane views: set where the mode is set)
        self._img_state = None

    @property
    def reference_formulation(self) -> bool:
        """True in the strict and the split16 mode: the reference's un-folded formulation (micro_sam_amd/strict.py) is the one that runs."""
        return self.precision in ("strict", "split16")

    def _strict_decoder(self):
        if getattr(self, "_strict_dec", None) is None:
            from .strict import StrictDecoder
            self._strict_dec = StrictDecoder(self)
        return self._strict_dec

    def _decode_strict(self, features, sparse, dense, multimask_output: bool):
        _, _, consts = self._prepare_decoder()
        pos = consts[: GRID * GRID * PROMPT_DIM * 4].view(torch.float32).reshape(GRID * GRID, PROMPT_DIM)
        return self._strict_decoder().decode(features, sparse, dense, pos, multimask_output)

    def set_split_token_mlp(self, on: bool) -> None:
        if bool(on) != bool(self.split_token_mlp):
            self.split_token_mlp = bool(on)
            self._dec = None
            self._img_state = None

    def lane_view(self) -> "Sam":
        """A second handle on THIS model for a concurrent decode lane (another HIP stream working on another image): parameters, modules
        and the prepared decoder constants are shared, the per-call scratch - the prepared image state and the decoder workspace - is
        the view's own, so two lanes never write the same buffers.  A weight update is noticed by every handle on its own
        (``_prepare_decoder`` compares the parameter key).  Used by the pipelined slice / tile loops
        (multi_dimensional_segmentation.segment_slices)."""
        import copy
        view = copy.copy(self)              # shallow: the _parameters / _modules / _buffers dicts are the same objects
        view._img_state = None
        view._dec_ws = None
        return view

    def _apply(self, fn, *args, **kwargs):   # .to(device) / .cuda() move parameters: drop cached device copies
        self.invalidate()
        return super()._apply(fn, *args, **kwargs)

    # -- reference: Sam.preprocess / Sam.postprocess_masks (micro_sam/util.py:670, trainable_sam.py:108)
    def preprocess(self, x: torch.Tensor) -> torch.Tensor:
        x = (x - self.pixel_mean) / self.pixel_std
        h, w = x.shape[-2:]
        return F.pad(x, (0, IMG_SIZE - w, 0, IMG_SIZE - h))

    def postprocess_masks(self, masks: torch.Tensor, input_size, original_size) -> torch.Tensor:
        """Returns fp32 logits at the original size, computed by the fused HIP resampling kernel."""
        from .ops import postprocess_masks as _pp
        b, c = masks.shape[:2]
        res = _pp(masks.reshape(b * c, 256, 256), tuple(input_size), tuple(original_size), self.mask_threshold, 1.0,
                  want_logits=True)
        return res["logits"].reshape(b, c, *original_size)

    # -- decoder plumbing
    def _prepare_decoder(self):
        if self._dec is not None:
            if self._watch.key() == self._dec_versions:
                return self._dec
            self._dec = None        # a parameter was updated in place (optimizer step, copy_): rebuild the 16-bit copies / tables
            self._img_state = None
        dev = self.device
        _lib.require_gpu(dev)
        pe, md = self.prompt_encoder, self.mask_decoder
        keep: List[torch.Tensor] = []

        def k(t):
            t = t.to(dev)
            keep.append(t)
            return t.data_ptr()

        def attn(dst, mod):
            dst.q_w, dst.q_b = k(_d16(mod.q_proj.weight)), k(_f32(mod.q_proj.bias))
            dst.k_w, dst.k_b = k(_d16(mod.k_proj.weight)), k(_f32(mod.k_proj.bias))
            dst.v_w, dst.v_b = k(_d16(mod.v_proj.weight)), k(_f32(mod.v_proj.bias))
            dst.o_w, dst.o_b = k(_d16(mod.out_proj.weight)), k(_f32(mod.out_proj.bias))

        p = _lib.DecoderParams()
        p.pe_gauss = k(_f32(pe.pe_layer.positional_encoding_gaussian_matrix))
        p.point_embed = k(_f32(torch.cat([e.weight for e in pe.point_embeddings], dim=0)))
        p.not_a_point = k(_f32(pe.not_a_point_embed.weight.reshape(-1)))
        p.no_mask = k(_f32(pe.no_mask_embed.weight.reshape(-1)))
        p.out_tokens = k(_f32(torch.cat([md.iou_token.weight, md.mask_tokens.weight], dim=0)))
        for i, blk in enumerate(md.transformer.layers):
            L = p.layer[i]
            attn(L.self_attn, blk.self_attn)
            attn(L.t2i, blk.cross_attn_token_to_image)
            attn(L.i2t, blk.cross_attn_image_to_token)
            for j, nm in enumerate((blk.norm1, blk.norm2, blk.norm3, blk.norm4), start=1):
                setattr(L, f"n{j}_w", k(_f32(nm.weight)))
                setattr(L, f"n{j}_b", k(_f32(nm.bias)))
            L.mlp1_w, L.mlp1_b = k(_d16(blk.mlp.lin1.weight)), k(_f32(blk.mlp.lin1.bias))
            L.mlp2_w, L.mlp2_b = k(_d16(blk.mlp.lin2.weight)), k(_f32(blk.mlp.lin2.bias))
            if self.split_token_mlp:
                L.mlp1_ws, L.mlp2_ws = k(_split16(blk.mlp.lin1.weight)), k(_split16(blk.mlp.lin2.weight))
        attn(p.final_attn, md.transformer.final_attn_token_to_image)
        p.nf_w, p.nf_b = k(_f32(md.transformer.norm_final_attn.weight)), k(_f32(md.transformer.norm_final_attn.bias))
        up = md.output_upscaling
        # ConvTranspose2d weight [ci, co, ky, kx] -> GEMM weight rows n = (ky*2+kx)*co_n + co, cols ci
        # (round 6) handed over CENTRED over the 64 output channels: LayerNorm2d follows the layer directly, so (W - mean_co W) x + (b - mean b) IS the
        # centred activation - the up-scaling kernel then has no mean to compute (csrc/upfused.hip CEN).  Centring happens on the fp32 weights, before
        # their rounding to the decoder's 16-bit type.  MSAM_UP_CENTRED=0: plain weights (the kernel computes the mean itself; the round 1 - 5 form).
        w1 = up[0].weight.detach().float().permute(2, 3, 1, 0).reshape(4, 64, PROMPT_DIM)          # [sub-pixel, co, ci]
        b1 = up[0].bias.detach().float()
        centred = os.environ.get("MSAM_UP_CENTRED", "1") != "0"
        if centred:
            w1 = w1 - w1.mean(dim=1, keepdim=True)
            b1 = b1 - b1.mean()
        p.up1_w = k(_d16(w1.reshape(4 * 64, PROMPT_DIM)))
        p.up1_b = k(_f32(b1.repeat(4)))
        p.up1_centred = 1 if centred else 0
        p.up_ln_w, p.up_ln_b = k(_f32(up[1].weight)), k(_f32(up[1].bias))
        p.up2_w = k(_d16(up[3].weight.permute(2, 3, 1, 0).reshape(4 * 32, 64)))
        p.up2_b = k(_f32(up[3].bias))

        def pad_rows(wt, bs, rows=128):
            wp = torch.zeros((rows, wt.shape[1]), dtype=wt.dtype, device=wt.device)
            wp[: wt.shape[0]] = wt
            bp = torch.zeros((rows,), dtype=bs.dtype, device=bs.device)
            bp[: bs.shape[0]] = bs
            return wp, bp

        for i, mlp in enumerate(md.output_hypernetworks_mlps):
            for j, lin in enumerate(mlp.layers):
                wt, bs = lin.weight.detach(), lin.bias.detach()
                if j == 2:
                    wt, bs = pad_rows(wt, bs)
                p.hyp_w[i][j], p.hyp_b[i][j] = k(_d16(wt)), k(_f32(bs))
        for j, lin in enumerate(md.iou_prediction_head.layers):
            wt, bs = lin.weight.detach(), lin.bias.detach()
            if j == 2:
                wt, bs = pad_rows(wt, bs)
            p.iou_w[j], p.iou_b[j] = k(_d16(wt)), k(_f32(bs))
        p.use_glds = int(self.use_glds)
        ds = pe.mask_downscaling                                  # mask prompts: Conv, LN2d, GELU, Conv, LN2d, GELU, Conv
        mp = _lib.MaskPromptParams()
        mp.c1_w, mp.c1_b = k(_f32(ds[0].weight.reshape(-1))), k(_f32(ds[0].bias))
        mp.ln1_w, mp.ln1_b = k(_f32(ds[1].weight)), k(_f32(ds[1].bias))
        mp.c2_w, mp.c2_b = k(_f32(ds[3].weight.reshape(-1))), k(_f32(ds[3].bias))
        mp.ln2_w, mp.ln2_b = k(_f32(ds[4].weight)), k(_f32(ds[4].bias))
        mp.c3_w, mp.c3_b = k(_f32(ds[6].weight.reshape(PROMPT_DIM, 16))), k(_f32(ds[6].bias))
        mp.exact_gelu = 1 if self.reference_formulation else 0
        self._mask_params = mp
        lib = _lib.load()
        consts = torch.empty(lib.msam_decoder_const_bytes(), dtype=torch.uint8, device=dev)
        _lib.check(lib.msam_decoder_prepare_const(C.byref(p), consts.data_ptr(), _lib.stream_ptr()),
                   "msam_decoder_prepare_const")
        self._dec = (p, keep, consts)
        self._dec_versions = self._watch.key()
        return self._dec

    def _dense_pe(self) -> torch.Tensor:
        _, _, consts = self._prepare_decoder()
        pos = consts[: GRID * GRID * PROMPT_DIM * 4].view(torch.float32).reshape(GRID * GRID, PROMPT_DIM)
        return pos.t().reshape(1, PROMPT_DIM, GRID, GRID).contiguous()

    def _image_state(self, features: torch.Tensor, prepared=None) -> torch.Tensor:
        # the cache entry keeps the caller's tensor alive and is matched by identity + version: a freed embedding whose
        # address is handed to the next one can no longer alias it (SamPredictor.reset_image / set_image also drop it)
        key = (features.data_ptr(), features._version, tuple(features.shape))
        # first: drops the image state too when a decoder parameter changed (`prepared`: the caller has just done it)
        p, _, consts = prepared if prepared is not None else self._prepare_decoder()
        if self._img_state is not None and self._img_state[0] == key and self._img_state[3] is features:
            return self._img_state[1]
        lib = _lib.load()
        feats = features.to(device=self.device, dtype=torch.float32).reshape(PROMPT_DIM, GRID * GRID).contiguous()
        state = torch.empty(lib.msam_decoder_image_bytes(), dtype=torch.uint8, device=self.device)
        _lib.check(lib.msam_decoder_prepare_image(C.byref(p), consts.data_ptr(), feats.data_ptr(), state.data_ptr(),
                                                  None, 0, _lib.stream_ptr()), "msam_decoder_prepare_image")
        self._img_state = (key, state, feats, features)
        return state

    def _workspace(self, P: int) -> torch.Tensor:
        need = _lib.load().msam_decoder_workspace_bytes(P)
        if self._dec_ws is None or self._dec_ws.numel() < need or self._dec_ws.device != self.device:
            self._dec_ws = None
            self._dec_ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._dec_ws

    @torch.no_grad()
    def _prompt_encode(self, points, boxes, masks):
        """``PromptEncoder.forward`` through ``msam_prompt_encode``."""
        p, _, _ = self._prepare_decoder()
        dev = self.device
        pts = lbl = bx = msk = None
        if points is not None:
            pts = points[0].to(device=dev, dtype=torch.float32).contiguous()
            lbl = points[1].to(device=dev, dtype=torch.int32).contiguous()
            B, Np = pts.shape[0], pts.shape[1]
        else:
            Np = 0
            B = boxes.shape[0] if boxes is not None else (masks.shape[0] if masks is not None else 1)
        if boxes is not None:
            bx = boxes.to(device=dev, dtype=torch.float32).reshape(-1, 4).contiguous()
        if masks is not None:
            msk = masks.to(device=dev, dtype=torch.float32).reshape(-1, 4 * GRID, 4 * GRID).contiguous()
            if msk.shape[0] != B:
                raise ValueError(f"masks holds {msk.shape[0]} masks for {B} prompts")
        Ns = Np + (2 if boxes is not None else (1 if Np > 0 else 0))
        sparse = torch.empty((B, Ns, PROMPT_DIM), dtype=torch.float32, device=dev)
        dense = torch.empty((B, PROMPT_DIM, GRID, GRID), dtype=torch.float32, device=dev) if msk is not None else None
        if Ns > 0 or msk is not None:
            self._mask_params.exact_gelu = 1 if self.reference_formulation else 0
            _lib.check(_lib.load().msam_prompt_encode(
                C.byref(p), C.byref(self._mask_params), _lib.ptr(pts), _lib.ptr(lbl), Np, _lib.ptr(bx), _lib.ptr(msk), B,
                _lib.ptr(sparse) if Ns > 0 else None, _lib.ptr(dense), _lib.stream_ptr()), "msam_prompt_encode")
        if dense is None:
            dense = self.prompt_encoder.no_mask_embed.weight.detach().to(dev).reshape(1, -1, 1, 1).expand(B, -1, GRID, GRID)
        return sparse, dense

    @torch.no_grad()
    def _decode_embeddings(self, image_embeddings, image_pe, sparse, dense, multimask_output: bool):
        """``MaskDecoder.forward`` through ``msam_decoder_forward_embeddings``."""
        if image_embeddings.numel() != PROMPT_DIM * GRID * GRID:
            raise NotImplementedError("micro_sam_amd: mask_decoder takes ONE image embedding [1,256,64,64] per call "
                                      f"(got {tuple(image_embeddings.shape)})")
        p, _, consts = self._prepare_decoder()
        dev = self.device
        if self.reference_formulation:
            nm = self.prompt_encoder.no_mask_embed.weight.detach().to(dev).reshape(1, -1, 1, 1)
            dn = dense.to(device=dev, dtype=torch.float32)
            no_mask = (dn.stride(-1) == 0 and dn.stride(-2) == 0 and torch.equal(dn[:, :, :1, :1], nm.expand(dn.shape[0], -1, 1, 1))) \
                or bool(torch.equal(dn, nm.expand_as(dn)))
            return self._decode_strict(image_embeddings, sparse.to(device=dev, dtype=torch.float32), None if no_mask else dn,
                                       multimask_output)
        own_pe = self._dense_pe()
        if image_pe is not None and (tuple(image_pe.shape) != tuple(own_pe.shape) or
                                     not torch.allclose(image_pe.to(device=dev, dtype=torch.float32), own_pe, atol=1e-5)):
            raise NotImplementedError("micro_sam_amd: mask_decoder needs image_pe == prompt_encoder.get_dense_pe() "
                                      "(the positional projections are precomputed tables)")
        state = self._image_state(image_embeddings)
        sp = sparse.to(device=dev, dtype=torch.float32).contiguous()
        P, Ns = sp.shape[0], sp.shape[1]
        if Ns > 11:
            raise ValueError(f"at most 11 sparse prompt tokens per prompt, got {Ns}")
        # the broadcast no_mask_embed (upstream: weight.reshape(1,-1,1,1).expand(...)) is already part of the prepared image
        # state; any other dense embedding gives every prompt its own source stream
        nm = self.prompt_encoder.no_mask_embed.weight.detach().to(dev).reshape(1, -1, 1, 1)
        dn = dense.to(device=dev, dtype=torch.float32)
        is_no_mask = (dn.stride(-1) == 0 and dn.stride(-2) == 0 and torch.equal(dn[:, :, :1, :1], nm.expand(dn.shape[0], -1, 1, 1))) \
            or bool(torch.equal(dn, nm.expand_as(dn)))
        dptr = eptr = None
        if not is_no_mask:
            if dn.shape[0] != P:
                raise ValueError(f"dense_prompt_embeddings holds {dn.shape[0]} entries for {P} prompts")
            dn = dn.contiguous()
            emb = image_embeddings.to(device=dev, dtype=torch.float32).reshape(PROMPT_DIM, GRID * GRID).contiguous()
            dptr, eptr = dn.data_ptr(), emb.data_ptr()
        nc = 3 if multimask_output else 1
        low = torch.empty((P, nc, 256, 256), dtype=torch.float32, device=dev)
        iou = torch.empty((P, nc), dtype=torch.float32, device=dev)
        ws = self._workspace(P)
        p.low_res_dtype = _lib.F32
        _lib.check(_lib.load().msam_decoder_forward_embeddings(
            C.byref(p), consts.data_ptr(), state.data_ptr(), sp.data_ptr() if Ns > 0 else None, Ns, dptr, eptr, P,
            1 if multimask_output else 0, low.data_ptr(), iou.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_ptr()),
            "msam_decoder_forward_embeddings")
        return low, iou

    @torch.no_grad()
    def decode(self, features: torch.Tensor, point_coords: Optional[torch.Tensor], point_labels: Optional[torch.Tensor],
               boxes: Optional[torch.Tensor] = None, mask_input: Optional[torch.Tensor] = None,
               multimask_output: bool = True, low_res_dtype: torch.dtype = torch.float32) -> Tuple[torch.Tensor, torch.Tensor]:
        """prompt_encoder + mask_decoder of ``SamPredictor.predict_torch`` for one image embedding [1,256,64,64].

        point_coords [P,Np,2] / boxes [P,4] are in the 1024 input frame.  Returns (low_res [P,C,256,256], iou [P,C]).
        ``low_res_dtype=torch.float16``: the low-res logits leave the up-scaling kernel as fp16 (the AMG path: 3072 masks per tile
        are handed to the fused post-processing and never returned - half the HBM round trip; ``predict_torch`` keeps fp32)."""
        if low_res_dtype not in (torch.float32, torch.float16):
            raise ValueError("low_res_dtype is torch.float32 or torch.float16")
        if features.numel() != PROMPT_DIM * GRID * GRID:
            raise ValueError(f"expected one image embedding [1,256,64,64], got {tuple(features.shape)}")
        if self.reference_formulation:
            if point_coords is None and boxes is None and mask_input is None:
                raise ValueError("micro_sam_amd: a prompt needs points, a box and / or a mask input")
            pts = None if point_coords is None else (point_coords, point_labels)
            msk = None if mask_input is None else mask_input.reshape(-1, 1, 4 * GRID, 4 * GRID)
            sparse, dense = self._prompt_encode(pts, None if boxes is None else boxes.reshape(-1, 4), msk)
            low, iou = self._decode_strict(features, sparse, dense if mask_input is not None else None, multimask_output)
            return (low if low_res_dtype == torch.float32 else low.to(low_res_dtype)), iou
        prepared = self._prepare_decoder()
        p, _, consts = prepared
        state = self._image_state(features, prepared)
        lib = _lib.load()
        dev = self.device
        if point_coords is not None:
            pts = point_coords.to(device=dev, dtype=torch.float32).contiguous()
            lbl = point_labels.to(device=dev, dtype=torch.int32).contiguous()
            P, Np = pts.shape[0], pts.shape[1]
        else:
            if boxes is None and mask_input is None:
                raise ValueError("micro_sam_amd: a prompt needs points, a box and / or a mask input")
            # (a mask prompt on its own - reference segment_from_mask(use_box=False, use_points=False) - decodes with the five output
            #  tokens only; the reference's own tests call that prompt unreliable, test/test_prompt_based_segmentation.py:8-13)
            pts = lbl = None
            P, Np = (boxes.shape[0] if boxes is not None else mask_input.reshape(-1, 4 * GRID, 4 * GRID).shape[0]), 0
        bx = None if boxes is None else boxes.to(device=dev, dtype=torch.float32).reshape(-1, 4).contiguous()
        nc = 3 if multimask_output else 1
        low = torch.empty((P, nc, 256, 256), dtype=low_res_dtype, device=dev)
        iou = torch.empty((P, nc), dtype=torch.float32, device=dev)
        p.low_res_dtype = _lib.F16 if low_res_dtype == torch.float16 else _lib.F32      # (read by this call only: set per call)
        need = lib.msam_decoder_workspace_bytes(P)
        if self._dec_ws is None or self._dec_ws.numel() < need or self._dec_ws.device != dev:
            self._dec_ws = None
            self._dec_ws = torch.empty(need, dtype=torch.uint8, device=dev)
        msk = None
        if mask_input is not None:                       # low-res logits of a previous prediction, one per prompt
            msk = mask_input.to(device=dev, dtype=torch.float32).reshape(-1, 4 * GRID, 4 * GRID).contiguous()
            if msk.shape[0] != P:
                raise ValueError(f"mask_input holds {msk.shape[0]} masks for {P} prompts")
        _lib.check(lib.msam_decoder_forward_masks(
            C.byref(p), C.byref(self._mask_params), consts.data_ptr(), state.data_ptr(), _lib.ptr(pts), _lib.ptr(lbl), Np,
            _lib.ptr(bx), _lib.ptr(msk), P, 1 if multimask_output else 0, low.data_ptr(), iou.data_ptr(),
            self._dec_ws.data_ptr(), self._dec_ws.numel(), _lib.stream_ptr()), "msam_decoder_forward")
        return low, iou


def build_sam(model_type: str = "vit_b", num_multimask_outputs: int = 3) -> Sam:
    """micro_sam/models/build_sam.py:87-142 (image_size fixed at 1024)."""
    key = model_type[:5]
    if key == "vit_t":
        # MobileSAM (micro_sam/util.py:435-439): TinyViT image encoder - torch operators, see models/tiny_vit.py - with the same
        # prompt encoder and mask decoder (the HIP path)
        from .models.tiny_vit import TinyViT
        sam = Sam(TinyViT(), PromptEncoder(), MaskDecoder(num_multimask_outputs))
        sam.eval()
        return sam
    if key not in VIT_CONFIGS:
        raise ValueError(f"Invalid model_type: {model_type}. Expect one of {tuple(VIT_CONFIGS) + ('vit_t',)}")
    cfg = VIT_CONFIGS[key]
    sam = Sam(ImageEncoderViT(cfg["embed_dim"], cfg["depth"], cfg["num_heads"], cfg["global_attn_indexes"]),
              PromptEncoder(), MaskDecoder(num_multimask_outputs))
    sam.eval()
    return sam


sam_model_registry = {
    "vit_b": lambda **kw: build_sam("vit_b", **kw),
    "vit_l": lambda **kw: build_sam("vit_l", **kw),
    "vit_h": lambda **kw: build_sam("vit_h", **kw),
    "vit_t": lambda **kw: build_sam("vit_t", **kw),
}
