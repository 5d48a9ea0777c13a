"""CPU oracle for the SAM arithmetic on micro_sam's hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module.  The product (``micro_sam_amd``) never does.

What it restates
----------------
micro_sam has no model arithmetic of its own: ``micro_sam/util.py:435-441`` builds the model through
``segment_anything.sam_model_registry`` (un-vendored third party, ``setup.cfg:58`` — *unpinned*, i.e.
PyPI ``segment-anything`` 1.0) with the hyper-parameters of ``micro_sam/models/build_sam.py:87-142``.
This file restates the published v1.0 algorithm of that package, functionally over an upstream-named
``state_dict`` (names: SURVEY.md Appendix C; evidence inside the reference at
``micro_sam/models/build_sam.py:26`` and ``micro_sam/util.py:578-598``):

* ``image_encoder``      ViT with 14x14 windowed + global attention and decomposed relative position
                         bias, LayerNorm eps 1e-6, exact GELU; neck conv1x1 -> LN2d -> conv3x3 -> LN2d.
* ``prompt_encoder``     random-Fourier positional encoding of points / boxes, mask down-scaling.
* ``mask_decoder``       two-way transformer (depth 2, 8 heads, down-sample 2, LN eps 1e-5), transposed-conv
                         up-scaling, hyper-network mask product, IoU head.
* ``preprocess`` / ``postprocess_masks`` (``Sam``) and ``ResizeLongestSide``.

Parity pinning: the reference ships no golden tensors for this arithmetic.  The
restatement is pinned against ``transformers.models.sam`` (same network, present in the build
container) by ``tests/test_oracle_vs_hf.py`` (weights copied through the Appendix-C key map; the HF
two-way-block LayerNorm eps is forced to upstream's 1e-5) and, for the integer post-processing, against
the reference's own known-answer tests (``tests/test_oracle_amg.py``).

Precision modes
---------------
``precision="fp32"``  strict fp32 (the reference CPU path).
``precision="bf16"``  emulates the rounding points of the HIP pipeline (config 2 of BASELINE.json,
                      "vit_b bf16"): every matrix-product operand is rounded to 16 bits - bfloat16 in the image
                      encoder, ``DECODER_DTYPE`` (fp16 in the default library build) in the mask decoder -,
                      products are accumulated in fp32, everything else (LayerNorm, softmax, GELU, bias,
                      residuals) stays fp32.  Tensors the HIP path *stores* in 16 bits are rounded at the same
                      place.  ``Prec(..., exact_sites / only_sites)`` and ``Prec.site_dtype`` are ablation hooks.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor

# Hyper-parameters, micro_sam/models/build_sam.py:40-84
VIT_CONFIGS = {
    "vit_b": dict(embed_dim=768, depth=12, num_heads=12, global_attn_indexes=(2, 5, 8, 11)),
    "vit_l": dict(embed_dim=1024, depth=24, num_heads=16, global_attn_indexes=(5, 11, 17, 23)),
    "vit_h": dict(embed_dim=1280, depth=32, num_heads=16, global_attn_indexes=(7, 15, 23, 31)),
}
IMG_SIZE = 1024          # build_sam.py:94
PATCH = 16               # build_sam.py:97
GRID = IMG_SIZE // PATCH  # 64
WINDOW = 14              # build_sam.py:111
PROMPT_DIM = 256         # build_sam.py:96
PIXEL_MEAN = (123.675, 116.28, 103.53)   # build_sam.py:132
PIXEL_STD = (58.395, 57.12, 57.375)      # build_sam.py:133


# 16-bit type of the HIP path's mask decoder (weights, image-token stream, folded vectors, probabilities, token-side
# tensors): IEEE fp16 in the default build of libmsam_hip.so (csrc/common.h MSAM_DEC_F16), bf16 in the ablation build.
# tests/conftest.py sets this from ``micro_sam_amd._lib.decoder_dtype()`` on a GPU box.
DECODER_DTYPE = torch.float16
# 16-bit operand type of the image encoder in the "bf16" policy (the product: bf16; torch.float16 = what-if ablation, tools/iou_ablation.py encfp16)
ENCODER_DTYPE = torch.bfloat16
# The product's image encoder takes the operands of the patch embedding and of the two neck convolutions as hi + lo pairs of the
# 16-bit type (msam_encoder_t.split_io, micro_sam_amd.modeling.ImageEncoderViT.split_io; default on): the "bf16" policy rounds
# these sites to hi + lo instead of to 16 bits.  False = the plain 16-bit policy of rounds 1 / 2 (set_split_io(False)).
ENCODER_SPLIT_IO = True


class Prec:
    """Rounding policy (see module docstring)."""

    def __init__(self, precision: str = "fp32", exact_sites=(), only_sites=None):
        """``exact_sites`` / ``only_sites`` (ablation hooks, bf16 mode): decoder rounding sites kept in fp32 / the only sites
        that round.  Sites: "stream" (the image-token stream as stored), "tok" (token-side projections and their stored
        q / k / v / probabilities), "t2i0" (layer 0's shared K / V), "fold" (folded vectors K' / V' / Q'), "table"
        (pe W^T + b tables), "probs" (attention probabilities of the folded kernels), "foldv" (per-head value projection
        weights of the folded token->image attention), "up" (up-scaling operands), "head" (hyper-network / IoU MLPs)."""
        assert precision in ("fp32", "bf16", "fp8"), precision
        self.exact_sites = set(exact_sites)
        self.only_sites = None if only_sites is None else set(only_sites)
        self.dec_dtype = DECODER_DTYPE
        # decoder site -> storage / operand dtype (default: DECODER_DTYPE).  The two products of the token MLP take their operands as
        # hi + lo pairs of the decoder's 16-bit type (round 4: csrc/decoder.hip mlp_split - the 2048-wide ReLU hidden of layer 0 carried
        # 60 % of the decoder's logit error on generic weights, profiles/r04_experiments.md section 3)
        self.site_dtype = {"tok.mlp": "split16" if DECODER_DTYPE == torch.float16 else "split"}
        # "fp8" (BASELINE config 5): the bf16 policy everywhere, except that the four large projections of every encoder
        # block (qkv, proj, lin1, lin2) take OCP e4m3 operands - activations with one scale per token, weights with one
        # scale per output channel (linear_q)
        self.bf16 = precision in ("bf16", "fp8")
        self.fp8 = precision == "fp8"
        # encoder rounding sites (ablation hooks, tools/enc_ablation.py): ``enc_only`` = the only encoder sites that round
        # (None: all), ``enc_blocks`` = the only blocks whose sites round (None: all; patch embedding = -1, neck = depth),
        # ``enc_dtype`` = site -> operand dtype or "split" (default ENCODER_DTYPE).  Sites: "patch", "qkv.x", "qkv.w",
        # "qkvstore" (q / k / v as stored), "relpos" (tables), "probs", "proj.x", "proj.w", "lin1.x", "lin1.w", "lin2.x",
        # "lin2.w", "neck".
        self.enc_only = None
        self.enc_blocks = None
        self.enc_dtype = {"patch": "split", "neck": "split"} if ENCODER_SPLIT_IO else {}
        self.block = -1

    def rounds(self, site: Optional[str]) -> bool:
        if not self.bf16:
            return False
        if site is None:
            return self.only_sites is None
        # sub-sites: "tok.x" / "tok.w" (operands of the token-side products), "tok.mlp(.x / .w)" (the two MLP products) < "tok"
        chain = [site] + [site.rsplit(".", n)[0] for n in range(1, site.count(".") + 1)]
        if self.only_sites is not None:
            return any(c in self.only_sites for c in chain)
        return not any(c in self.exact_sites for c in chain)

    def r(self, x: Tensor, site: Optional[str] = None) -> Tensor:
        """Round to bf16 (and back to fp32) in bf16 mode; identity in fp32 mode (or when the site is kept exact)."""
        if not self.rounds(site):
            return x
        if site is None:
            dt = ENCODER_DTYPE
        else:
            dt = self.dec_dtype
            for n in range(site.count("."), -1, -1):          # the most specific entry wins: "tok.lin.w" > "tok.lin" > "tok"
                key = site.rsplit(".", n)[0] if n else site
                if key in self.site_dtype:
                    dt = self.site_dtype[key]
        if dt == "split":             # bf16 hi + lo operand pair (two / three MFMA passes): ~16 mantissa bits
            hi = x.to(torch.bfloat16).to(torch.float32)
            return hi + (x - hi).to(torch.bfloat16).to(torch.float32)
        if dt == "split16":           # fp16 hi + lo operand pair: ~22 significand bits (the mask decoder's token side since round 4)
            hi = x.to(torch.float16).to(torch.float32)
            return hi + (x - hi).to(torch.float16).to(torch.float32)
        return x.to(dt).to(torch.float32)

    def re(self, x: Tensor, site: str) -> Tensor:
        """Encoder rounding site ``site`` of the current block (``self.block``)."""
        if not self.bf16:
            return x
        if self.enc_only is not None and site not in self.enc_only:
            return x
        if self.enc_blocks is not None and self.block not in self.enc_blocks:
            return x
        dt = self.enc_dtype.get(site, ENCODER_DTYPE)
        if dt == "split":             # hi + lo pair of the encoder's 16-bit type
            hi = x.to(ENCODER_DTYPE).to(torch.float32)
            return hi + (x - hi).to(ENCODER_DTYPE).to(torch.float32)
        return x.to(dt).to(torch.float32)

    def linear(self, x: Tensor, w: Tensor, b: Optional[Tensor] = None, site: Optional[str] = None,
               esite: Optional[str] = None) -> Tensor:
        if esite is not None:
            y = F.linear(self.re(x, esite + ".x"), self.re(w, esite + ".w"))
        else:
            # operands of a token-side product: sub-sites "<site>.x" (activations) / "<site>.w" (weights) of "tok" / "tok.mlp"
            tok = site is not None and site.split(".")[0] == "tok"
            y = F.linear(self.r(x, site + ".x" if tok else site), self.r(w, site + ".w" if tok else site))
        return y if b is None else y + b

    def matmul(self, a: Tensor, b: Tensor) -> Tensor:
        return torch.matmul(self.r(a), self.r(b))

    @staticmethod
    def quant_rows_e4m3(x: Tensor) -> Tensor:
        """Per-row (last axis) e4m3 quantise / dequantise: q = round(x * (1 / scale)), scale = amax / 448 (1 for a zero row):
        the arithmetic of msam_layernorm_fp8 / msam_quant_rows_fp8."""
        amax = x.abs().amax(dim=-1, keepdim=True)
        scale = torch.where(amax > 0, amax * (1.0 / 448.0), torch.ones_like(amax))
        q = (x * (1.0 / scale)).to(torch.float8_e4m3fn).to(torch.float32)
        return q * scale

    def linear_q(self, x: Tensor, w: Tensor, b: Optional[Tensor] = None, src: str = "fp32", esite: Optional[str] = None) -> Tensor:
        """One of the four large encoder projections.  fp8 mode: x is quantised per token from fp32 (LayerNorm outputs,
        src="fp32") or from its bf16 copy (attention output, MLP hidden: src="bf16"), w per output channel
        (micro_sam_amd.ops.quant_weight_fp8).  Other modes: ``linear``."""
        if not self.fp8:
            return self.linear(x, w, b, esite=esite)
        xq = self.quant_rows_e4m3(x if src == "fp32" else self.r(x))
        amax = w.abs().amax(dim=1, keepdim=True)
        ws = torch.where(amax > 0, amax / 448.0, torch.ones_like(amax))
        wq = (w / ws).to(torch.float8_e4m3fn).to(torch.float32) * ws
        y = F.linear(xq, wq)
        return y if b is None else y + b


# ----------------------------------------------------------------------------------------------
# Pre / post processing (upstream Sam.preprocess / postprocess_masks, ResizeLongestSide)
# ----------------------------------------------------------------------------------------------

def get_preprocess_shape(oldh: int, oldw: int, long_side: int = IMG_SIZE) -> Tuple[int, int]:
    """ResizeLongestSide.get_preprocess_shape."""
    scale = long_side * 1.0 / max(oldh, oldw)
    newh, neww = oldh * scale, oldw * scale
    return int(newh + 0.5), int(neww + 0.5)


def apply_image(image: np.ndarray) -> np.ndarray:
    """ResizeLongestSide.apply_image: uint8 HWC -> PIL bilinear resize so that the long side is 1024.

    Called by the reference at micro_sam/util.py:663 and through set_image (util.py:916)."""
    from PIL import Image
    h, w = image.shape[:2]
    newh, neww = get_preprocess_shape(h, w)
    if (newh, neww) == (h, w):
        return np.array(image)
    # torchvision's to_pil_image + resize((h, w)) == PIL BILINEAR with size=(w, h)
    return np.array(Image.fromarray(image).resize((neww, newh), Image.BILINEAR))


def apply_coords(coords: np.ndarray, original_size: Tuple[int, int]) -> np.ndarray:
    """ResizeLongestSide.apply_coords (used at instance_segmentation.py:358)."""
    old_h, old_w = original_size
    new_h, new_w = get_preprocess_shape(old_h, old_w)
    coords = np.array(coords, dtype=float, copy=True)
    coords[..., 0] = coords[..., 0] * (new_w / old_w)
    coords[..., 1] = coords[..., 1] * (new_h / old_h)
    return coords


def apply_boxes(boxes: np.ndarray, original_size: Tuple[int, int]) -> np.ndarray:
    return apply_coords(np.asarray(boxes).reshape(-1, 2, 2), original_size).reshape(-1, 4)


def preprocess(x: Tensor) -> Tensor:
    """Sam.preprocess: normalise with pixel mean/std and zero-pad bottom/right to 1024 (util.py:670)."""
    mean = torch.tensor(PIXEL_MEAN, dtype=torch.float32, device=x.device).view(1, 3, 1, 1)
    std = torch.tensor(PIXEL_STD, dtype=torch.float32, device=x.device).view(1, 3, 1, 1)
    x = (x.to(torch.float32) - mean) / std
    h, w = x.shape[-2:]
    return F.pad(x, (0, IMG_SIZE - w, 0, IMG_SIZE - h))


def postprocess_masks(masks: Tensor, input_size: Tuple[int, int], original_size: Tuple[int, int]) -> Tensor:
    """Sam.postprocess_masks: x4 bilinear to 1024^2, crop the padding, bilinear to the original size."""
    masks = F.interpolate(masks, (IMG_SIZE, IMG_SIZE), mode="bilinear", align_corners=False)
    masks = masks[..., : input_size[0], : input_size[1]]
    masks = F.interpolate(masks, tuple(original_size), mode="bilinear", align_corners=False)
    return masks


# ----------------------------------------------------------------------------------------------
# Image encoder (upstream modeling/image_encoder.py)
# ----------------------------------------------------------------------------------------------

def _get_rel_pos(q_size: int, k_size: int, rel_pos: Tensor) -> Tensor:
    max_rel_dist = int(2 * max(q_size, k_size) - 1)
    if rel_pos.shape[0] != max_rel_dist:
        rel_pos_resized = F.interpolate(
            rel_pos.reshape(1, rel_pos.shape[0], -1).permute(0, 2, 1), size=max_rel_dist, mode="linear",
        )
        rel_pos_resized = rel_pos_resized.reshape(-1, max_rel_dist).permute(1, 0)
    else:
        rel_pos_resized = rel_pos
    q_coords = torch.arange(q_size)[:, None] * max(k_size / q_size, 1.0)
    k_coords = torch.arange(k_size)[None, :] * max(q_size / k_size, 1.0)
    relative_coords = (q_coords - k_coords) + (k_size - 1) * max(q_size / k_size, 1.0)
    return rel_pos_resized[relative_coords.long()]


def _attention_relpos(sd: Dict[str, Tensor], pre: str, x: Tensor, num_heads: int, p: Prec) -> Tensor:
    """Attention with decomposed rel-pos on a [B', S, S, D] window/global grid."""
    Bp, H, W, D = x.shape
    hd = D // num_heads
    scale = hd ** -0.5
    qkv = p.linear_q(x, sd[pre + "qkv.weight"], sd[pre + "qkv.bias"], src="fp32", esite="qkv")
    qkv = p.re(qkv, "qkvstore")  # HIP path stores q, k, v in bf16
    qkv = qkv.reshape(Bp, H * W, 3, num_heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv.reshape(3, Bp * num_heads, H * W, hd).unbind(0)
    attn = (q * scale) @ k.transpose(-2, -1)
    Rh = _get_rel_pos(H, H, p.re(sd[pre + "rel_pos_h"], "relpos"))
    Rw = _get_rel_pos(W, W, p.re(sd[pre + "rel_pos_w"], "relpos"))
    r_q = q.reshape(Bp * num_heads, H, W, hd)      # NB: unscaled q (upstream behaviour)
    rel_h = torch.einsum("bhwc,hkc->bhwk", r_q, Rh)
    rel_w = torch.einsum("bhwc,wkc->bhwk", r_q, Rw)
    attn = (attn.view(-1, H, W, H, W) + rel_h[:, :, :, :, None] + rel_w[:, :, :, None, :]).view(-1, H * W, H * W)
    if p.bf16:
        # flash-style: unnormalised probabilities rounded to bf16, fp32 row sum of the unrounded values
        m = attn.max(dim=-1, keepdim=True).values
        e = torch.exp(attn - m)
        o = (p.re(e, "probs") @ v) / e.sum(dim=-1, keepdim=True)
    else:
        o = attn.softmax(dim=-1) @ v
    o = o.view(Bp, num_heads, H, W, hd).permute(0, 2, 3, 1, 4).reshape(Bp, H, W, D)
    return p.linear_q(o, sd[pre + "proj.weight"], sd[pre + "proj.bias"], src="bf16", esite="proj")


def _window_partition(x: Tensor, ws: int):
    B, H, W, C = x.shape
    pad_h = (ws - H % ws) % ws
    pad_w = (ws - W % ws) % ws
    if pad_h or pad_w:
        x = F.pad(x, (0, 0, 0, pad_w, 0, pad_h))
    Hp, Wp = H + pad_h, W + pad_w
    x = x.view(B, Hp // ws, ws, Wp // ws, ws, C)
    windows = x.permute(0, 1, 3, 2, 4, 5).contiguous().view(-1, ws, ws, C)
    return windows, (Hp, Wp)


def _window_unpartition(windows: Tensor, ws: int, pad_hw, hw):
    Hp, Wp = pad_hw
    H, W = hw
    B = windows.shape[0] // (Hp * Wp // ws // ws)
    x = windows.view(B, Hp // ws, Wp // ws, ws, ws, -1)
    x = x.permute(0, 1, 3, 2, 4, 5).contiguous().view(B, Hp, Wp, -1)
    return x[:, :H, :W, :].contiguous()


def layer_norm_2d(x: Tensor, w: Tensor, b: Tensor, eps: float = 1e-6) -> Tensor:
    """LayerNorm2d (upstream modeling/common.py): normalise over the channel dim of NCHW."""
    u = x.mean(1, keepdim=True)
    s = (x - u).pow(2).mean(1, keepdim=True)
    x = (x - u) / torch.sqrt(s + eps)
    return w[:, None, None] * x + b[:, None, None]


def image_encoder(sd: Dict[str, Tensor], x: Tensor, model_type: str = "vit_b", precision: str = "fp32",
                  return_blocks: bool = False, stop_after_block: Optional[int] = None):
    """ImageEncoderViT.forward: [B,3,1024,1024] normalised -> [B,256,64,64].  Keys prefixed 'image_encoder.'.
    ``stop_after_block`` (test hook, needs ``return_blocks``): return (None, residual streams of blocks 0..stop) without
    running the rest of the network (the vit_h test checks the first blocks only: CPU time)."""
    cfg = VIT_CONFIGS[model_type[:5]]
    p = precision if isinstance(precision, Prec) else Prec(precision)
    pre = "image_encoder."
    D, heads = cfg["embed_dim"], cfg["num_heads"]
    w = sd[pre + "patch_embed.proj.weight"]
    p.block = -1
    x = F.conv2d(p.re(x, "patch"), p.re(w, "patch"), None, stride=PATCH) + sd[pre + "patch_embed.proj.bias"].view(1, -1, 1, 1)
    x = x.permute(0, 2, 3, 1)                                  # NHWC [B,64,64,D]
    x = x + sd[pre + "pos_embed"]
    taps = []
    for i in range(cfg["depth"]):
        bp = f"{pre}blocks.{i}."
        p.block = i
        shortcut = x
        y = F.layer_norm(x, (D,), sd[bp + "norm1.weight"], sd[bp + "norm1.bias"], eps=1e-6)
        if i in cfg["global_attn_indexes"]:
            y = _attention_relpos(sd, bp + "attn.", y, heads, p)
        else:
            H, W = y.shape[1:3]
            y, pad_hw = _window_partition(y, WINDOW)
            y = _attention_relpos(sd, bp + "attn.", y, heads, p)
            y = _window_unpartition(y, WINDOW, pad_hw, (H, W))
        x = shortcut + y
        y = F.layer_norm(x, (D,), sd[bp + "norm2.weight"], sd[bp + "norm2.bias"], eps=1e-6)
        y = p.linear_q(y, sd[bp + "mlp.lin1.weight"], sd[bp + "mlp.lin1.bias"], src="fp32", esite="lin1")
        y = F.gelu(y)                                          # exact erf GELU
        y = p.linear_q(y, sd[bp + "mlp.lin2.weight"], sd[bp + "mlp.lin2.bias"], src="bf16", esite="lin2")
        x = x + y
        if return_blocks:
            taps.append(x.clone())
            if stop_after_block is not None and i == stop_after_block:
                return None, taps
    x = x.permute(0, 3, 1, 2)                                  # NCHW
    p.block = cfg["depth"]
    x = F.conv2d(p.re(x, "neck"), p.re(sd[pre + "neck.0.weight"], "neck"))
    x = layer_norm_2d(x, sd[pre + "neck.1.weight"], sd[pre + "neck.1.bias"])
    x = F.conv2d(p.re(x, "neck"), p.re(sd[pre + "neck.2.weight"], "neck"), padding=1)
    x = layer_norm_2d(x, sd[pre + "neck.3.weight"], sd[pre + "neck.3.bias"])
    return (x, taps) if return_blocks else x


# ----------------------------------------------------------------------------------------------
# Prompt encoder (upstream modeling/prompt_encoder.py)
# ----------------------------------------------------------------------------------------------

def _pe_encoding(G: Tensor, coords: Tensor) -> Tensor:
    """PositionEmbeddingRandom._pe_encoding: coords in [0,1]^2 -> 256-d."""
    coords = 2 * coords - 1
    coords = coords @ G
    coords = 2 * np.pi * coords
    return torch.cat([torch.sin(coords), torch.cos(coords)], dim=-1)


def get_dense_pe(sd: Dict[str, Tensor]) -> Tensor:
    """PromptEncoder.get_dense_pe: [1,256,64,64]."""
    G = sd["prompt_encoder.pe_layer.positional_encoding_gaussian_matrix"]
    grid = torch.ones((GRID, GRID), dtype=torch.float32, device=G.device)
    y_embed = (grid.cumsum(dim=0) - 0.5) / GRID
    x_embed = (grid.cumsum(dim=1) - 0.5) / GRID
    pe = _pe_encoding(G, torch.stack([x_embed, y_embed], dim=-1))
    return pe.permute(2, 0, 1).unsqueeze(0)


def prompt_encoder(sd: Dict[str, Tensor], points: Optional[Tuple[Tensor, Tensor]], boxes: Optional[Tensor],
                   masks: Optional[Tensor]) -> Tuple[Tensor, Tensor]:
    """PromptEncoder.forward -> (sparse [B,N,256], dense [B,256,64,64])."""
    pre = "prompt_encoder."
    G = sd[pre + "pe_layer.positional_encoding_gaussian_matrix"]
    dev = G.device
    if points is not None:
        bs = points[0].shape[0]
    elif boxes is not None:
        bs = boxes.shape[0]
    elif masks is not None:
        bs = masks.shape[0]
    else:
        bs = 1
    sparse = torch.empty((bs, 0, PROMPT_DIM), device=dev)
    if points is not None:
        coords, labels = points
        coords = coords.to(torch.float32) + 0.5
        if boxes is None:
            coords = torch.cat([coords, torch.zeros((bs, 1, 2), device=dev)], dim=1)
            labels = torch.cat([labels, -torch.ones((bs, 1), device=dev, dtype=labels.dtype)], dim=1)
        norm = coords.clone()
        norm[:, :, 0] = norm[:, :, 0] / IMG_SIZE
        norm[:, :, 1] = norm[:, :, 1] / IMG_SIZE
        e = _pe_encoding(G, norm)
        e[labels == -1] = 0.0
        e[labels == -1] += sd[pre + "not_a_point_embed.weight"]
        e[labels == 0] += sd[pre + "point_embeddings.0.weight"]
        e[labels == 1] += sd[pre + "point_embeddings.1.weight"]
        sparse = torch.cat([sparse, e], dim=1)
    if boxes is not None:
        b = boxes.to(torch.float32) + 0.5
        c = b.reshape(-1, 2, 2).clone()
        c[:, :, 0] = c[:, :, 0] / IMG_SIZE
        c[:, :, 1] = c[:, :, 1] / IMG_SIZE
        e = _pe_encoding(G, c)
        e[:, 0, :] += sd[pre + "point_embeddings.2.weight"][0]
        e[:, 1, :] += sd[pre + "point_embeddings.3.weight"][0]
        sparse = torch.cat([sparse, e], dim=1)
    if masks is not None:
        m = F.conv2d(masks, sd[pre + "mask_downscaling.0.weight"], sd[pre + "mask_downscaling.0.bias"], stride=2)
        m = F.gelu(layer_norm_2d(m, sd[pre + "mask_downscaling.1.weight"], sd[pre + "mask_downscaling.1.bias"]))
        m = F.conv2d(m, sd[pre + "mask_downscaling.3.weight"], sd[pre + "mask_downscaling.3.bias"], stride=2)
        m = F.gelu(layer_norm_2d(m, sd[pre + "mask_downscaling.4.weight"], sd[pre + "mask_downscaling.4.bias"]))
        dense = F.conv2d(m, sd[pre + "mask_downscaling.6.weight"], sd[pre + "mask_downscaling.6.bias"])
    else:
        dense = sd[pre + "no_mask_embed.weight"].reshape(1, -1, 1, 1).expand(bs, -1, GRID, GRID)
    return sparse, dense


# ----------------------------------------------------------------------------------------------
# Mask decoder (upstream modeling/mask_decoder.py + transformer.py)
# ----------------------------------------------------------------------------------------------

def _dec_attention(sd, pre: str, q: Tensor, k: Tensor, v: Tensor, p: Prec, heads: int = 8,
                   q_pe: Optional[Tensor] = None, k_pe: Optional[Tensor] = None, mfma_pv: bool = False,
                   fold: bool = False, fold_i2t: bool = False, q_site: str = "tok", kv_site: str = "tok") -> Tensor:
    """q_pe / k_pe: positional encodings of the IMAGE-side operand.  fp32 mode adds them before the projection
    (upstream); bf16 mode follows the HIP dataflow (x + pe) W = x W + pe W with separately rounded operands.
    fold (bf16 mode only, token->image attention over the per-prompt stream with <= 8 tokens): the HIP path folds the
    K / V projections into the token side (csrc/decfold.hip) - same mathematics, different bf16 rounding points."""
    if p.bf16 and fold and q.shape[1] <= 8:
        return _dec_attention_folded(sd, pre, q, k, p, heads, k_pe)
    if p.bf16 and fold_i2t and k.shape[1] <= 8:
        return _dec_attention_i2t_folded(sd, pre, q, k, v, p, heads, q_pe)
    def proj(x, pe, name, site):
        w, b = sd[pre + name + ".weight"], sd[pre + name + ".bias"]
        if pe is None:
            return p.linear(x, w, b, site)
        if not p.rounds(site):
            return p.linear(x + pe, w, b, site)
        return p.linear(x, w, b, site) + p.linear(pe, w, None, site)
    q = proj(q, q_pe, "q_proj", q_site)
    k = proj(k, k_pe, "k_proj", kv_site)
    v = p.linear(v, sd[pre + "v_proj.weight"], sd[pre + "v_proj.bias"], kv_site)
    q, k, v = p.r(q, q_site), p.r(k, kv_site), p.r(v, kv_site)   # HIP path stores the projected q/k/v in bf16
    b, nq, c = q.shape
    def sep(t):
        return t.reshape(b, t.shape[1], heads, c // heads).transpose(1, 2)
    q, k, v = sep(q), sep(k), sep(v)
    attn = (q @ k.transpose(-2, -1)) / math.sqrt(c // heads)
    if p.bf16 and mfma_pv:
        # HIP cross attentions: un-normalised probabilities rounded to bf16 for the P.V MFMA, fp32 row sum
        e = torch.exp(attn - attn.max(dim=-1, keepdim=True).values)
        out = (p.r(e, kv_site) @ v) / e.sum(dim=-1, keepdim=True)
    else:
        out = torch.softmax(attn, dim=-1) @ v
    out = out.transpose(1, 2).reshape(b, nq, c)
    return p.linear(out, sd[pre + "out_proj.weight"], sd[pre + "out_proj.bias"], q_site)


def _dec_attention_folded(sd, pre: str, q: Tensor, keys: Tensor, p: Prec, heads: int, k_pe: Tensor) -> Tensor:
    """bf16 emulation of csrc/decfold.hip: S = keys . bf16(Wk_h^T q_h) + bf16(pe Wk^T + bk)_h . q_h, un-normalised
    probabilities rounded to bf16 for the P . keys MFMA, context kept in fp32 for the per-head value projection."""
    wk, bk = sd[pre + "k_proj.weight"], sd[pre + "k_proj.bias"]
    wv, bv = sd[pre + "v_proj.weight"], sd[pre + "v_proj.bias"]
    q = p.r(p.linear(q, sd[pre + "q_proj.weight"], sd[pre + "q_proj.bias"], "tok"), "tok")            # [b, nq, 128]
    b, nq, ci = q.shape
    hd = ci // heads
    qh = q.reshape(b, nq, heads, hd).transpose(1, 2)                                     # [b, h, nq, 16]
    wkh = p.r(wk, "fold").reshape(heads, hd, -1)                                         # [h, 16, 256]
    qf = p.r(torch.einsum("bhtd,hdc->bhtc", qh, wkh), "fold")                            # folded queries, bf16
    tab = p.r(p.linear(k_pe[:1], wk, bk, "table"), "table")[0].reshape(-1, heads, hd)    # [T, h, 16]
    keys = p.r(keys, "stream")
    s = (torch.einsum("bhtc,bjc->bhtj", qf, keys) + torch.einsum("bhtd,jhd->bhtj", qh, tab)) / math.sqrt(hd)
    e = torch.exp(s - s.max(dim=-1, keepdim=True).values)
    ctx = torch.einsum("bhtj,bjc->bhtc", p.r(e, "probs"), keys) / e.sum(dim=-1, keepdim=True)     # [b, h, nq, 256] fp32
    out = torch.einsum("bhtc,hdc->bhtd", ctx, p.r(wv, "foldv").reshape(heads, hd, -1)) + bv.reshape(1, heads, 1, hd)
    out = out.transpose(1, 2).reshape(b, nq, ci)
    return p.linear(out, sd[pre + "out_proj.weight"], sd[pre + "out_proj.bias"], "tok")


def _dec_attention_i2t_folded(sd, pre: str, keys: Tensor, k_in: Tensor, v_in: Tensor, p: Prec, heads: int,
                              q_pe: Tensor) -> Tensor:
    """bf16 emulation of csrc/decfold.hip fold_i2t_kernel: S = keys . bf16(Wq_h^T k_h) + bf16(pe Wq^T + bq)_h . k_h,
    softmax over the prompt tokens, normalised P rounded to bf16, out = P . bf16(Wo_h v_h) + bo (out_proj folded)."""
    wq, bq = sd[pre + "q_proj.weight"], sd[pre + "q_proj.bias"]
    wo, bo = sd[pre + "out_proj.weight"], sd[pre + "out_proj.bias"]
    k = p.r(p.linear(k_in, sd[pre + "k_proj.weight"], sd[pre + "k_proj.bias"], "tok"), "tok")        # [b, nt, 128]
    v = p.r(p.linear(v_in, sd[pre + "v_proj.weight"], sd[pre + "v_proj.bias"], "tok"), "tok")
    b, nt, ci = k.shape
    hd = ci // heads
    kh = k.reshape(b, nt, heads, hd).transpose(1, 2)                                    # [b, h, nt, 16]
    vh = v.reshape(b, nt, heads, hd).transpose(1, 2)
    kf = p.r(torch.einsum("bhtd,hdc->bhtc", kh, p.r(wq, "fold").reshape(heads, hd, -1)), "fold")        # K' [b, h, nt, 256]
    vf = p.r(torch.einsum("chd,bhtd->bhtc", p.r(wo, "fold").reshape(-1, heads, hd), vh), "fold")        # V' [b, h, nt, 256]
    tab = p.r(p.linear(q_pe[:1], wq, bq, "table"), "table")[0].reshape(-1, heads, hd)   # [T, h, 16]
    keys = p.r(keys, "stream")
    s = (torch.einsum("bjc,bhtc->bjht", keys, kf) + torch.einsum("jhd,bhtd->bjht", tab, kh)) / math.sqrt(hd)
    a = p.r(torch.softmax(s, dim=-1), "probs")
    return torch.einsum("bjht,bhtc->bjc", a, vf) + bo


def _ln(sd, pre: str, x: Tensor, eps: float = 1e-5) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), sd[pre + "weight"], sd[pre + "bias"], eps=eps)


def two_way_transformer(sd, image_embedding: Tensor, image_pe: Tensor, point_embedding: Tensor, p: Prec,
                        debug: Optional[dict] = None):
    pre = "mask_decoder.transformer."
    bs, c, h, w = image_embedding.shape
    keys = image_embedding.flatten(2).permute(0, 2, 1)
    key_pe = image_pe.flatten(2).permute(0, 2, 1)
    queries = point_embedding
    query_pe = point_embedding
    keys = p.r(keys, "stream")  # HIP path keeps the image-token stream in bf16
    for i in range(2):
        lp = f"{pre}layers.{i}."
        if i == 0:
            queries = _dec_attention(sd, lp + "self_attn.", queries, queries, queries, p)
        else:
            q = queries + query_pe
            queries = queries + _dec_attention(sd, lp + "self_attn.", q, q, queries, p)
        queries = _ln(sd, lp + "norm1.", queries)
        q = queries + query_pe
        queries = queries + _dec_attention(sd, lp + "cross_attn_token_to_image.", q, keys, keys, p, k_pe=key_pe,
                                           mfma_pv=True, fold=i > 0, kv_site="t2i0")
        queries = _ln(sd, lp + "norm2.", queries)
        m = p.linear(queries, sd[lp + "mlp.lin1.weight"], sd[lp + "mlp.lin1.bias"], "tok.mlp")
        m = p.linear(F.relu(m), sd[lp + "mlp.lin2.weight"], sd[lp + "mlp.lin2.bias"], "tok.mlp")
        queries = _ln(sd, lp + "norm3.", queries + m)
        q = queries + query_pe
        keys = keys + _dec_attention(sd, lp + "cross_attn_image_to_token.", keys, q, queries, p, q_pe=key_pe,
                                     mfma_pv=True, fold_i2t=True)
        keys = p.r(_ln(sd, lp + "norm4.", keys), "stream")
        if debug is not None:
            debug[f"queries{i}"] = queries.clone()
            debug[f"keys{i}"] = keys.clone()
    q = queries + query_pe
    queries = queries + _dec_attention(sd, pre + "final_attn_token_to_image.", q, keys, keys, p, k_pe=key_pe,
                                       mfma_pv=True, fold=True)
    queries = _ln(sd, pre + "norm_final_attn.", queries)
    if debug is not None:
        debug["queries_final"] = queries.clone()
    return queries, keys


def _mlp3(sd, pre: str, x: Tensor, p: Prec) -> Tensor:
    x = F.relu(p.linear(x, sd[pre + "layers.0.weight"], sd[pre + "layers.0.bias"], "head"))
    x = F.relu(p.linear(x, sd[pre + "layers.1.weight"], sd[pre + "layers.1.bias"], "head"))
    return p.linear(x, sd[pre + "layers.2.weight"], sd[pre + "layers.2.bias"], "head")


def mask_decoder(sd, image_embeddings: Tensor, image_pe: Tensor, sparse: Tensor, dense: Tensor,
                 multimask_output: bool, precision: str = "fp32", debug: Optional[dict] = None) -> Tuple[Tensor, Tensor]:
    """MaskDecoder.forward -> (low-res masks [B,C,256,256], iou [B,C])."""
    p = precision if isinstance(precision, Prec) else Prec(precision)
    pre = "mask_decoder."
    output_tokens = torch.cat([sd[pre + "iou_token.weight"], sd[pre + "mask_tokens.weight"]], dim=0)
    output_tokens = output_tokens.unsqueeze(0).expand(sparse.size(0), -1, -1)
    tokens = torch.cat((output_tokens, sparse), dim=1)
    src = torch.repeat_interleave(image_embeddings, tokens.shape[0], dim=0)
    src = src + dense
    pos_src = torch.repeat_interleave(image_pe, tokens.shape[0], dim=0)
    b, c, h, w = src.shape
    if debug is not None:
        debug["tokens"] = tokens.clone()
    hs, src = two_way_transformer(sd, src, pos_src, tokens, p, debug)
    iou_token_out = hs[:, 0, :]
    mask_tokens_out = hs[:, 1:5, :]
    src = src.transpose(1, 2).view(b, c, h, w)
    up = F.conv_transpose2d(p.r(src, "stream"), p.r(sd[pre + "output_upscaling.0.weight"], "up"), None, stride=2)
    up = up + sd[pre + "output_upscaling.0.bias"].view(1, -1, 1, 1)
    up = F.gelu(layer_norm_2d(up, sd[pre + "output_upscaling.1.weight"], sd[pre + "output_upscaling.1.bias"]))
    if debug is not None:
        debug["up1"] = up.clone()
    up = F.conv_transpose2d(p.r(up, "up"), p.r(sd[pre + "output_upscaling.3.weight"], "up"), None, stride=2)
    up = F.gelu(up + sd[pre + "output_upscaling.3.bias"].view(1, -1, 1, 1))
    hyper = torch.stack(
        [_mlp3(sd, f"{pre}output_hypernetworks_mlps.{i}.", mask_tokens_out[:, i, :], p) for i in range(4)], dim=1)
    if debug is not None:
        debug["hyper"] = hyper.clone()
        debug["up"] = up.clone()
    b, c, h, w = up.shape
    masks = (hyper @ up.view(b, c, h * w)).view(b, -1, h, w)   # fp32 product in both modes
    iou = _mlp3(sd, pre + "iou_prediction_head.", iou_token_out, p)
    sl = slice(1, None) if multimask_output else slice(0, 1)
    return masks[:, sl, :, :], iou[:, sl]


def predict_torch(sd, features: Tensor, input_size, original_size, point_coords: Optional[Tensor],
                  point_labels: Optional[Tensor], boxes: Optional[Tensor] = None, mask_input: Optional[Tensor] = None,
                  multimask_output: bool = True, return_logits: bool = False, precision: str = "fp32",
                  debug: Optional[dict] = None, low_res_fp16: bool = False):
    """SamPredictor.predict_torch -> (masks, iou, low_res).  ``low_res_fp16`` (HIP-like modes only): the AMG path of
    the product hands its low-res logits from the up-scaling kernel to the post-processing kernel as fp16 (round 4)."""
    points = (point_coords, point_labels) if point_coords is not None else None
    sparse, dense = prompt_encoder(sd, points, boxes, mask_input)
    low_res, iou = mask_decoder(sd, features, get_dense_pe(sd), sparse, dense, multimask_output, precision, debug)
    if low_res_fp16 and (precision if isinstance(precision, Prec) else Prec(precision)).bf16:
        low_res = low_res.to(torch.float16).to(torch.float32)
    masks = postprocess_masks(low_res, input_size, original_size)
    if not return_logits:
        masks = masks > 0.0
    return masks, iou, low_res


@torch.no_grad()
def predict(sd, features: Tensor, input_size, original_size, point_coords: Optional[np.ndarray] = None,
            point_labels: Optional[np.ndarray] = None, box: Optional[np.ndarray] = None,
            mask_input: Optional[np.ndarray] = None, multimask_output: bool = True, return_logits: bool = False,
            precision: str = "fp32"):
    """SamPredictor.predict (segment_anything/predictor.py, un-vendored; the reference's call sites:
    micro_sam/prompt_based_segmentation.py:287,393,438,493): numpy prompts of ONE object in original image coordinates
    (points [N,2] XY, labels [N], box [4] XYXY, mask_input [1,256,256]) -> (masks [C,H,W], iou [C], low_res [C,256,256])."""
    coords = labels = boxes = masks_in = None
    if point_coords is not None:
        assert point_labels is not None
        coords = torch.as_tensor(apply_coords(np.asarray(point_coords), original_size), dtype=torch.float32)[None]
        labels = torch.as_tensor(np.asarray(point_labels), dtype=torch.int)[None]
    if box is not None:
        boxes = torch.as_tensor(apply_boxes(np.asarray(box), original_size), dtype=torch.float32)[None]
    if mask_input is not None:
        masks_in = torch.as_tensor(np.asarray(mask_input), dtype=torch.float32)[None]
    masks, iou, low = predict_torch(sd, features, input_size, original_size, coords, labels, boxes, masks_in,
                                    multimask_output=multimask_output, return_logits=return_logits, precision=precision)
    return masks[0].numpy(), iou[0].numpy(), low[0].numpy()
