"""The "strict" precision mode: ``Sam.set_precision("strict")`` / ``SamPredictor.set_precision("strict")``.

north_star asks for results that match the reference CPU path (mask IoU >= 0.999 per instance, identical instance ids).  The
default path computes SAM in 16-bit MFMA arithmetic with a folded / chained decoder (DESIGN.md sections 3, 4) - fast, and within
one boundary pixel of the reference for most instances, but not for all of them.  This module is the other end of the
speed / parity curve: the reference's OWN formulation (segment_anything ``ImageEncoderViT`` / ``MaskDecoder`` as restated in
SURVEY.md Appendix A; reference call sites ``micro_sam/util.py:674`` and ``micro_sam/instance_segmentation.py:361-366``), every
tensor fp32, every product on the f32-input MFMA (``msam_strict_gemm``: exact fp32 products, fp32 accumulation), erf GELU, expf
softmax, IEEE divisions (``csrc/strict.hip``).  What differs from torch's CPU result is the order of the additions inside a
product.  Host code below only owns buffers and the sequence of library calls - there is no torch arithmetic on this path
(``torch.cat`` / ``expand`` / views move data).

Cost: fp32 MFMA runs at 1/16 of the bf16 rate and nothing is folded: a 1024-prompt tile takes 58 ms against 5.8 ms on the default
path (DESIGN.md sections 4 and 6 have the measured numbers and the steps that took it there from 116 ms); it is still 840 x the
reference's CPU rate on the box's 32 host threads.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import threading
import weakref
from typing import Optional, Tuple

import torch

from . import _lib

GRID, PROMPT_DIM, WINDOW = 64, 256, 14
T = GRID * GRID
ACT_NONE, ACT_GELU, ACT_RELU = _lib.ACT_NONE, _lib.ACT_GELU, _lib.ACT_RELU
# prompts per pass of the strict decoder: ~30 MiB of fp32 intermediates per prompt (image-token stream, its projections, the two
# up-scaling stages) = 15 GiB per pass of 512 - sized for the 288 GB of one MI355X (three decode lanes: 45 GiB).  Measured on the bench
# tile (1024 prompts): 128 -> 65.5 ms, 256 -> 61.5, 512 -> 59.3, 1024 -> 58.9 ms per tile (the token side's small products fill the chip
# better; profiles/r05_experiments.md section 7).  Element counts stay below 2^31 up to 1023 prompts per pass.
DECODE_CHUNK = 512
DECODE_CHUNK_SPLIT = int(os.environ.get("MSAM_SPLIT16_CHUNK", "1024"))
# the "image attends to the tokens" step as one launch (msam_strict_i2t_block) instead of four (projection, attention, projection +
# residual, LayerNorm): the same arithmetic, the 0.5 GB per-chunk stream crosses HBM twice instead of seven times.  Tokens <= 16.
FUSED_I2T = True
# the token -> image attentions' k and v projections of the image stream as one product over [Wk; Wv] (msam_sgemm_t.a2_cols: the
# stream crosses HBM once).  Measured: 59.4 ms per tile against 58.5 with two launches - these products are bound by the f32 MFMA,
# not by HBM; off.
FUSED_KV = False
# ... but in the split16 mode the same products are bound by HBM (4.1 TB/s on the k / v shape), so there the one-launch form is the default
FUSED_KV_SPLIT = True
# split16 only: "tokens attend to the image" on a per-prompt stream (layer >= 1 and the final attention) as msam_split16_t2i_attention - the k / v
# projections folded into the token side, online softmax, the stream read once (17 GB -> 4.3 GB per 1024-prompt layer); Tk <= 8
FUSED_T2I = True
# split16 only: the image -> token block with BOTH projections folded into the prompt's tokens (msam_split16_i2t_block: a prompt's operands staged once,
# its 4096 rows walked by one workgroup) instead of si2t_kernel<SPLIT> (W_q / W_o streamed through LDS for every 128-row block); Tk <= 8
FOLDED_I2T = True
# split16: weights handed to the kernels as prepared fp16 pairs (cached per weight tensor) instead of being split again by every workgroup.
# MEASURED SLOWER and therefore off: the product 221 -> 233 us (encoder qkv), 323 -> 448 us (second up-scaling shape: two 8-byte loads per
# thread and k-tile instead of one 16-byte load), the fused image -> token block 3.50 -> 3.64 ms - these kernels are not bound by the
# conversion's vector instructions (profiles/r06_experiments.md)
PREPARED_WEIGHTS = False
# split16 only: the up-scaling's LayerNorm2d, GELU, second transposed convolution, GELU and hyper product as one launch
# (msam_strict_upscale2: the 4.3 GB first-stage stream of a tile is read once; as four launches 34 GB cross HBM)
FUSED_UP2 = True


# fp32 intermediates of one prompt in a decode pass: strict - the image-token stream, its k | v projections and the two up-scaling stages (30 MiB);
# split16 - the stream, the first up-scaling stage and the fused kernels' workspaces (13 MiB live at the peak: keys + up1 + a k | v projection of layer 0)
BYTES_PER_PROMPT = {False: 30 << 20, True: 13 << 20}


def decode_chunk(dev, split: bool) -> int:
    """Prompts per decode pass: the configured chunk (DECODE_CHUNK / DECODE_CHUNK_SPLIT), reduced to what fits HALF of the device memory that is free
    or cached by torch right now (several decode lanes and the AMG state share the device) - a 1024-prompt pass needs 13 GiB (split16) / a 512-prompt
    pass 15 GiB (strict), which a 288 GB MI355X holds many times over but a smaller or busier device may not (the chunk was a constant)."""
    want = DECODE_CHUNK_SPLIT if split else DECODE_CHUNK
    dev = torch.device(dev)
    if dev.type != "cuda":
        return want
    free, _ = torch.cuda.mem_get_info(dev)
    free += torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)          # blocks torch holds but does not use
    fit = int(free // 2 // BYTES_PER_PROMPT[bool(split)])
    if fit >= want:
        return want
    if fit < 16:
        raise RuntimeError(f"micro_sam_amd: {free / 2 ** 30:.1f} GiB of free device memory is not enough for a reference-formulation decode pass "
                           f"(16 prompts need {16 * BYTES_PER_PROMPT[bool(split)] / 2 ** 30:.1f} GiB)")
    return 1 << (fit.bit_length() - 1)                   # a power of two below the fit


def _f32(t: torch.Tensor, dev) -> torch.Tensor:
    return t.detach().to(device=dev, dtype=torch.float32).contiguous()


# ---------------------------------------------------------------------------------------------------- the split16 mode
# ``Sam.set_precision("split16")``: the same formulation and the same sequence of library calls as the strict mode, with every
# matrix product on fp16 operand PAIRS (hi = fp16(x), lo = fp16(x - hi); a . w = a_hi w_hi + a_hi w_lo + a_lo w_hi on the 16-bit MFMA,
# fp32 accumulation; include/msam_hip.h msam_sgemm_t.split16).  Everything outside the products (LayerNorm, softmax, GELU, residuals,
# the tensors in HBM) stays fp32.  fp16's range is handled by power-of-two scales that are undone exactly in the epilogue: weights are
# scaled so that max |w| lies in (2^12, 2^13] (cached per weight tensor), activations enter as they are.
_MODE = threading.local()
_WSCALE = {}
_WPAIRS = {}


class split_mode:
    """Context: products of this thread run in the split16 mode (entered by StrictEncoder.forward / StrictDecoder.decode)."""

    def __init__(self, on: bool) -> None:
        self.on = bool(on)

    def __enter__(self):
        self.prev = getattr(_MODE, "split", False)
        _MODE.split = self.on
        return self

    def __exit__(self, *exc):
        _MODE.split = self.prev
        return False


def split_active() -> bool:
    return getattr(_MODE, "split", False)


def _cached(cache: dict, w: torch.Tensor, extra, make):
    """Per-tensor cache keyed by the tensor OBJECT (a weak reference guards against another tensor that later lives at the same address:
    a probe that allocates one random weight after another got the previous weight's pairs back - measured, round 6)."""
    key = (id(w), extra)
    hit = cache.get(key)
    if hit is not None and hit[0]() is w and hit[1] == (w.data_ptr(), w._version):
        return hit[2]
    val = make()
    cache[key] = (weakref.ref(w), (w.data_ptr(), w._version), val)
    return val


def weight_scale(w: torch.Tensor) -> float:
    """Power of two that brings max |w| into (2^12, 2^13] (fp16: 10 bits of lo below 11 bits of hi stay normal numbers down to 2^-22 of
    the largest weight).  One device synchronisation per weight tensor, cached until the weights are rebuilt (``forget_scales``)."""
    def make():
        m = float(w.abs().max())
        sc = 1.0 if not (m > 0.0 and math.isfinite(m)) else 2.0 ** (13 - math.ceil(math.log2(m)))
        return min(max(sc, 2.0 ** -40), 2.0 ** 40)
    return _cached(_WSCALE, w, None, make)


def weight_pairs(w: torch.Tensor, permute: bool) -> torch.Tensor:
    """``w`` [N, K] as fp16 pairs in the split16 kernels' LDS tile layout (msam_split16_prepare_pairs), scaled by ``weight_scale(w)``; cached with
    the scale."""
    def make():
        t = torch.empty((w.shape[0], 2 * w.shape[1]), dtype=torch.float16, device=w.device)
        _lib.check(_lib.load().msam_split16_prepare_pairs(w.data_ptr(), w.shape[0], w.shape[1], weight_scale(w), 1 if permute else 0, t.data_ptr(),
                                                          _lib.stream_ptr()), "msam_split16_prepare_pairs")
        if w.is_cuda:
            torch.cuda.current_stream(w.device).synchronize()      # (once per weight: decode lanes on other streams read it too)
        return t
    return _cached(_WPAIRS, w, bool(permute), make)


def forget_scales() -> None:
    _WSCALE.clear()
    _WPAIRS.clear()


# ---------------------------------------------------------------------------------------------------- library calls

def gemm(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, act: int = ACT_NONE,
         a2: Optional[torch.Tensor] = None, a2_rows: int = 0, res: Optional[torch.Tensor] = None, res_rows: int = 0,
         out: Optional[torch.Tensor] = None, rows: Optional[int] = None, lda: Optional[int] = None, a_offset: int = 0,
         a2_cols: int = 0) -> torch.Tensor:
    """``act((a + a2[row % a2_rows]) @ w.T + bias) + res[row % res_rows]`` (fp32).  ``a``: [M, K] rows (``rows`` / ``lda`` /
    ``a_offset`` address a strided row set inside a larger buffer: the output tokens of the two-way transformer); ``a2_cols``: ``a2``
    is added for the first ``a2_cols`` output columns only (two projections of one input in one launch)."""
    K = w.shape[1]
    M = a.shape[0] if rows is None else rows
    N = w.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=w.device)
    p = _lib.SGemmParams()
    p.A, p.lda = a.data_ptr() + 4 * a_offset, (a.stride(0) if lda is None else lda)
    if a2 is not None:
        p.A2, p.lda2, p.a2_rows, p.a2_cols = a2.data_ptr(), a2.stride(0), a2_rows, a2_cols
    p.W, p.ldw, p.M, p.N, p.K = w.data_ptr(), w.stride(0), M, N, K
    p.bias = None if bias is None else bias.data_ptr()
    p.act = act
    if res is not None:
        p.res, p.ldr, p.res_rows = res.data_ptr(), res.stride(0), res_rows
    p.out, p.ldc = out.data_ptr(), out.stride(0)
    if split_active():
        p.split16, p.a_scale, p.w_scale = 1, 1.0, weight_scale(w)
    _lib.check(_lib.load().msam_strict_gemm(C.byref(p), _lib.stream_ptr()), "msam_strict_gemm")
    return out


def layer_norm(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float, out: Optional[torch.Tensor] = None, gelu: bool = False,
               nchw_hw: int = 0, rows: Optional[int] = None, dim: Optional[int] = None) -> torch.Tensor:
    rows = x.shape[0] if rows is None else rows
    dim = x.shape[1] if dim is None else dim
    if out is None:
        out = torch.empty_like(x)
    _lib.check(_lib.load().msam_strict_layernorm(x.data_ptr(), w.data_ptr(), b.data_ptr(), float(eps), rows, dim, out.data_ptr(),
                                                 1 if gelu else 0, nchw_hw, _lib.stream_ptr()), "msam_strict_layernorm")
    return out


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, B: int, H: int, Nq: int, Nk: int, D: int, denom: float,
              q_shared: bool = False, kv_shared: bool = False) -> torch.Tensor:
    """softmax((q . k) / denom) @ v for q [B (or 1), Nq, H*D], k / v [B (or 1), Nk, H*D] (2-d row views) -> [B*Nq, H*D]."""
    out = torch.empty((B * Nq, H * D), dtype=torch.float32, device=q.device)
    _lib.check(_lib.load().msam_strict_attention(
        q.data_ptr(), q.stride(0), 0 if q_shared else Nq * q.stride(0), k.data_ptr(), k.stride(0), 0 if kv_shared else Nk * k.stride(0),
        v.data_ptr(), v.stride(0), 0 if kv_shared else Nk * v.stride(0), B, H, Nq, Nk, D, float(denom), out.data_ptr(), out.stride(0),
        Nq * out.stride(0), _lib.stream_ptr()), "msam_strict_attention")
    return out


def i2t_block(keys: torch.Tensor, shared: bool, pos: torch.Tensor, wq, tok_k: torch.Tensor, tok_v: torch.Tensor, wo, norm, B: int, Tk: int,
              out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``norm4(keys + cross_attn_image_to_token(q=keys + pos, k, v))`` in one launch (``msam_strict_i2t_block``): keys [B*T, 256] rows
    (``shared``: [T, 256] for every prompt), tok_k / tok_v [B*Tk, 128] the token side's projections, wq / wo (weight, bias)."""
    if out is None:
        out = torch.empty((B * T, PROMPT_DIM), dtype=torch.float32, device=pos.device) if shared else keys
    p = _lib.SI2TParams()
    p.keys, p.key_batch_stride, p.pos = keys.data_ptr(), (0 if shared else T * PROMPT_DIM), pos.data_ptr()
    p.wq, p.bq, p.wo, p.bo = wq[0].data_ptr(), wq[1].data_ptr(), wo[0].data_ptr(), wo[1].data_ptr()
    p.tok_k, p.tok_v, p.ld_tok, p.tok_batch_stride = tok_k.data_ptr(), tok_v.data_ptr(), tok_k.stride(0), Tk * tok_k.stride(0)
    p.ln_weight, p.ln_bias, p.ln_eps, p.denom = norm[0].data_ptr(), norm[1].data_ptr(), float(norm[2]), 4.0
    p.out, p.B, p.Tk = out.data_ptr(), B, Tk
    if split_active() and FOLDED_I2T and Tk <= 8:
        ws = torch.empty(B * 131328, dtype=torch.uint8, device=pos.device)
        _lib.check(_lib.load().msam_split16_i2t_block(C.byref(p), ws.data_ptr(), ws.numel(), _lib.stream_ptr()), "msam_split16_i2t_block")
        return out
    if split_active():
        p.split16, p.wq_scale, p.wo_scale = 1, weight_scale(wq[0]), weight_scale(wo[0])
        if PREPARED_WEIGHTS:
            p.wq_pairs, p.wo_pairs = weight_pairs(wq[0], False).data_ptr(), weight_pairs(wo[0], True).data_ptr()
    _lib.check(_lib.load().msam_strict_i2t_block(C.byref(p), _lib.stream_ptr()), "msam_strict_i2t_block")
    return out


# ---------------------------------------------------------------------------------------------------- image encoder

class StrictEncoder:
    """``ImageEncoderViT.forward`` in fp32 (the oracle's ``image_encoder(..., precision="fp32")`` step by step)."""

    def __init__(self, enc) -> None:
        self.enc = enc
        self._w = None
        self._key = None

    def _weights(self):
        from .modeling import _resize_rel_pos
        enc = self.enc
        key = enc._watch.key()
        if self._w is not None and key == self._key:
            return self._w
        dev = enc.pos_embed.device
        _lib.require_gpu(dev)
        forget_scales()
        D = enc.embed_dim
        w = {"patch_w": _f32(enc.patch_embed.proj.weight.reshape(D, -1), dev), "patch_b": _f32(enc.patch_embed.proj.bias, dev),
             "pos": _f32(enc.pos_embed.reshape(T, D), dev), "blocks": []}
        for blk in enc.blocks:
            size = GRID if blk.window_size == 0 else blk.window_size
            w["blocks"].append(dict(
                window=blk.window_size, scale=float(blk.attn.scale),
                ln1=(_f32(blk.norm1.weight, dev), _f32(blk.norm1.bias, dev), blk.norm1.eps),
                qkv=(_f32(blk.attn.qkv.weight, dev), _f32(blk.attn.qkv.bias, dev)),
                rel_h=_f32(_resize_rel_pos(blk.attn.rel_pos_h, size), dev), rel_w=_f32(_resize_rel_pos(blk.attn.rel_pos_w, size), dev),
                proj=(_f32(blk.attn.proj.weight, dev), _f32(blk.attn.proj.bias, dev)),
                ln2=(_f32(blk.norm2.weight, dev), _f32(blk.norm2.bias, dev), blk.norm2.eps),
                lin1=(_f32(blk.mlp.lin1.weight, dev), _f32(blk.mlp.lin1.bias, dev)),
                lin2=(_f32(blk.mlp.lin2.weight, dev), _f32(blk.mlp.lin2.bias, dev))))
        nk = enc.neck
        w["neck0"] = _f32(nk[0].weight.reshape(PROMPT_DIM, D), dev)
        w["neck1"] = (_f32(nk[1].weight, dev), _f32(nk[1].bias, dev), nk[1].eps)
        w["neck2"] = _f32(nk[2].weight.permute(0, 2, 3, 1).reshape(PROMPT_DIM, 9 * PROMPT_DIM), dev)      # columns (ky, kx, c)
        w["neck3"] = (_f32(nk[3].weight, dev), _f32(nk[3].bias, dev), nk[3].eps)
        self._w, self._key = w, key
        return w

    @torch.no_grad()
    def forward(self, x: Optional[torch.Tensor] = None, images_u8: Optional[torch.Tensor] = None) -> torch.Tensor:
        """x fp32 [B,3,1024,1024] (after ``Sam.preprocess``) or uint8 HWC [B,h,w,3] (``Sam.preprocess`` fused) -> [B,256,64,64]."""
        with split_mode(self.enc.precision == "split16"):
            return self._forward(x, images_u8)

    def _forward(self, x, images_u8):
        enc = self.enc
        w = self._weights()
        dev = enc.pos_embed.device
        lib = _lib.load()
        D, heads = enc.embed_dim, enc.num_heads
        hd = D // heads
        if x is not None:
            x = x.to(device=dev, dtype=torch.float32).contiguous()
            B, h, wd = x.shape[0], 0, 0
        else:
            images_u8 = images_u8.to(dev).contiguous()
            B, h, wd = images_u8.shape[:3]
        out = torch.empty((B, PROMPT_DIM, GRID, GRID), dtype=torch.float32, device=dev)
        # tiles per pass: the MLP hidden of one tile is 48 MiB (vit_b) - 80 MiB (vit_h) in fp32
        step = 8
        for b0 in range(0, B, step):
            nb = min(step, B - b0)
            patches = torch.empty((nb * T, 3 * 16 * 16), dtype=torch.float32, device=dev)
            _lib.check(lib.msam_strict_patchify(None if x is None else x[b0:b0 + nb].data_ptr(),
                                                None if x is not None else images_u8[b0:b0 + nb].data_ptr(), nb, h, wd, patches.data_ptr(),
                                                _lib.stream_ptr()), "msam_strict_patchify")
            xs = gemm(patches, w["patch_w"], w["patch_b"], res=w["pos"], res_rows=T)                  # conv(x) + bias, + pos_embed
            del patches
            for blk in w["blocks"]:
                y = layer_norm(xs, *blk["ln1"])
                qkv = gemm(y, *blk["qkv"])
                att = torch.empty((nb * T, D), dtype=torch.float32, device=dev)
                _lib.check((lib.msam_split16_relpos_attention if split_active() else lib.msam_strict_relpos_attention)(qkv.data_ptr(), blk["qkv"][1].data_ptr(), blk["rel_h"].data_ptr(),
                                                            blk["rel_w"].data_ptr(), nb, heads, hd, GRID, blk["window"], blk["scale"],
                                                            att.data_ptr(), _lib.stream_ptr()), "msam_strict_relpos_attention")
                del qkv
                gemm(att, *blk["proj"], res=xs, out=xs)                                               # shortcut + attention
                y = layer_norm(xs, *blk["ln2"], out=y)
                hid = gemm(y, *blk["lin1"], act=ACT_GELU)
                gemm(hid, *blk["lin2"], res=xs, out=xs)                                               # x + mlp(norm2(x))
                del hid, att, y
            y = gemm(xs, w["neck0"])
            layer_norm(y, *w["neck1"], out=y)
            cols = torch.empty((nb * T, 9 * PROMPT_DIM), dtype=torch.float32, device=dev)
            _lib.check(lib.msam_strict_im2col3x3(y.data_ptr(), nb, PROMPT_DIM, cols.data_ptr(), _lib.stream_ptr()), "msam_strict_im2col3x3")
            y = gemm(cols, w["neck2"], out=y)
            layer_norm(y, *w["neck3"], out=out[b0:b0 + nb], nchw_hw=T)
        return out


# ---------------------------------------------------------------------------------------------------- mask decoder

class StrictDecoder:
    """``PromptEncoder`` + ``MaskDecoder.forward`` in fp32 for P prompts on one image embedding (the oracle's
    ``mask_decoder(..., precision="fp32")`` step by step, the reference's un-folded two-way transformer)."""

    def __init__(self, sam) -> None:
        self.sam = sam
        self._w = None
        self._key = None

    def _weights(self):
        sam = self.sam
        key = sam._watch.key()
        if self._w is not None and key == self._key:
            return self._w
        dev = sam.device
        forget_scales()
        md, tr = sam.mask_decoder, sam.mask_decoder.transformer

        def attn(m):
            d = {n: (_f32(getattr(m, n + "_proj").weight, dev), _f32(getattr(m, n + "_proj").bias, dev)) for n in ("q", "k", "v", "out")}
            # [Wk; Wv]: the k and v projections of one input as ONE product (msam_sgemm_t.a2_cols: the positional encoding goes to k only)
            d["kv"] = (torch.cat([d["k"][0], d["v"][0]], dim=0).contiguous(), torch.cat([d["k"][1], d["v"][1]], dim=0).contiguous())
            return d

        def norm(m):
            return (_f32(m.weight, dev), _f32(m.bias, dev), m.eps)

        def mlp3(m):
            return [(_f32(lin.weight, dev), _f32(lin.bias, dev)) for lin in m.layers]
        w = {"layers": [], "final": attn(tr.final_attn_token_to_image), "norm_final": norm(tr.norm_final_attn)}
        for blk in tr.layers:
            w["layers"].append(dict(self_attn=attn(blk.self_attn), t2i=attn(blk.cross_attn_token_to_image),
                                    i2t=attn(blk.cross_attn_image_to_token), n1=norm(blk.norm1), n2=norm(blk.norm2), n3=norm(blk.norm3),
                                    n4=norm(blk.norm4), lin1=(_f32(blk.mlp.lin1.weight, dev), _f32(blk.mlp.lin1.bias, dev)),
                                    lin2=(_f32(blk.mlp.lin2.weight, dev), _f32(blk.mlp.lin2.bias, dev))))
        up = md.output_upscaling
        # a 2 x 2 stride-2 transposed convolution = one linear map per input pixel to its 2 x 2 output block: rows (ky, kx, co)
        w["up1"] = (_f32(up[0].weight.permute(2, 3, 1, 0).reshape(4 * 64, PROMPT_DIM), dev), _f32(up[0].bias.repeat(4), dev))
        w["up_ln"] = (_f32(up[1].weight, dev), _f32(up[1].bias, dev), up[1].eps)
        w["up2"] = (_f32(up[3].weight.permute(2, 3, 1, 0).reshape(4 * 32, 64), dev), _f32(up[3].bias.repeat(4), dev))
        w["hyper"] = [mlp3(m) for m in md.output_hypernetworks_mlps]
        w["iou"] = mlp3(md.iou_prediction_head)
        w["out_tokens"] = _f32(torch.cat([md.iou_token.weight, md.mask_tokens.weight], dim=0), dev)
        w["no_mask"] = _f32(sam.prompt_encoder.no_mask_embed.weight.reshape(-1), dev)
        self._w, self._key = w, key
        return w

    def _attn_block(self, aw, q_in, k_in, v_in, B, Nq, Nk, q_pe=None, q_pe_rows=0, k_pe=None, k_pe_rows=0, q_shared=False,
                    kv_shared=False):
        """upstream ``Attention.forward``: projections, heads, softmax, (out_proj is left to the caller: it carries the residual)."""
        q = gemm(q_in, *aw["q"], a2=q_pe, a2_rows=q_pe_rows)
        inner = aw["q"][0].shape[0]
        if (split_active() and FUSED_T2I and k_in is v_in and k_pe is not None and not kv_shared and not q_shared and Nk == T and Nq <= 8
                and inner == 128 and k_pe_rows == T):
            # tokens attending to a per-prompt image stream: k / v projections folded into the token side, one pass over the stream
            out = torch.empty((B * Nq, inner), dtype=torch.float32, device=q.device)
            ws = torch.empty(B * 131072, dtype=torch.uint8, device=q.device)
            p = _lib.ST2IParams()
            p.keys, p.key_batch_stride, p.pos, p.q, p.ldq = k_in.data_ptr(), T * PROMPT_DIM, k_pe.data_ptr(), q.data_ptr(), q.stride(0)
            p.wk, p.wv, p.bv, p.denom = aw["k"][0].data_ptr(), aw["v"][0].data_ptr(), aw["v"][1].data_ptr(), math.sqrt(inner // 8)
            p.out, p.ldo, p.B, p.Tk, p.workspace, p.workspace_bytes = out.data_ptr(), out.stride(0), B, Nq, ws.data_ptr(), ws.numel()
            _lib.check(_lib.load().msam_split16_t2i_attention(C.byref(p), _lib.stream_ptr()), "msam_split16_t2i_attention")
            return out
        if (FUSED_KV or (split_active() and FUSED_KV_SPLIT)) and k_in is v_in and k_pe is not None and inner % 128 == 0 and k_in.shape[0] >= 4096:
            # the image side's k | v: one pass over the per-prompt stream instead of two
            kv = gemm(k_in, *aw["kv"], a2=k_pe, a2_rows=k_pe_rows, a2_cols=inner)
            k, v = kv[:, :inner], kv[:, inner:]
        else:
            k = gemm(k_in, *aw["k"], a2=k_pe, a2_rows=k_pe_rows)
            v = gemm(v_in, *aw["v"])
        D = inner // 8
        return attention(q, k, v, B, 8, Nq, Nk, D, math.sqrt(D), q_shared=q_shared, kv_shared=kv_shared)

    @torch.no_grad()
    def decode(self, features: torch.Tensor, sparse: torch.Tensor, dense: Optional[torch.Tensor], pos: torch.Tensor,
               multimask_output: bool) -> Tuple[torch.Tensor, torch.Tensor]:
        """features [1,256,64,64], sparse fp32 [P, Ns, 256], dense fp32 [P,256,64,64] or None (= the broadcast no_mask_embed),
        pos fp32 [4096, 256] (token-major dense positional encoding) -> (low_res [P, C, 256, 256], iou [P, C])."""
        with split_mode(self.sam.precision == "split16"):
            return self._decode(features, sparse, dense, pos, multimask_output)

    def _decode(self, features, sparse, dense, pos, multimask_output):
        w = self._weights()
        dev = self.sam.device
        lib = _lib.load()
        P, Ns = sparse.shape[0], sparse.shape[1]
        Tk = 5 + Ns
        nc, mask0 = (3, 1) if multimask_output else (1, 0)
        low = torch.empty((P, nc, 256, 256), dtype=torch.float32, device=dev)
        iou = torch.empty((P, nc), dtype=torch.float32, device=dev)
        emb = features.to(device=dev, dtype=torch.float32).reshape(PROMPT_DIM, T).contiguous()
        shared = dense is None
        if shared:
            src_all = torch.empty((T, PROMPT_DIM), dtype=torch.float32, device=dev)
            _lib.check(lib.msam_strict_source(emb.data_ptr(), w["no_mask"].data_ptr(), 0, 1, src_all.data_ptr(), _lib.stream_ptr()),
                       "msam_strict_source")
        tokens_all = torch.cat([w["out_tokens"].unsqueeze(0).expand(P, -1, -1), sparse.to(device=dev, dtype=torch.float32)], dim=1).contiguous()
        # (split16: the second up-scaling stage is never materialised, so a whole 1024-prompt grid fits one pass - half the token-side launches)
        chunk = decode_chunk(dev, split_active() and FUSED_UP2)
        for p0 in range(0, P, chunk):
            pc = min(chunk, P - p0)
            qpe = tokens_all[p0:p0 + pc].reshape(pc * Tk, PROMPT_DIM)                       # query_pe = the prompt tokens themselves
            queries = qpe.clone()
            if shared:
                keys, ks = src_all, True
            else:
                dn = dense[p0:p0 + pc].to(device=dev, dtype=torch.float32).reshape(pc, PROMPT_DIM, T).contiguous()
                keys = torch.empty((pc * T, PROMPT_DIM), dtype=torch.float32, device=dev)
                _lib.check(lib.msam_strict_source(emb.data_ptr(), dn.data_ptr(), PROMPT_DIM * T, pc, keys.data_ptr(), _lib.stream_ptr()),
                           "msam_strict_source")
                ks = False
            for i, L in enumerate(w["layers"]):
                sa = L["self_attn"]
                if i == 0:                                                                   # skip_first_layer_pe
                    att = self._attn_block(sa, queries, queries, queries, pc, Tk, Tk)
                    queries = gemm(att, *sa["out"])
                else:
                    att = self._attn_block(sa, queries, queries, queries, pc, Tk, Tk, q_pe=qpe, k_pe=qpe)
                    queries = gemm(att, *sa["out"], res=queries)
                layer_norm(queries, *L["n1"], out=queries)
                ta = L["t2i"]                                                                # tokens attending to the image
                att = self._attn_block(ta, queries, keys, keys, pc, Tk, T, q_pe=qpe, k_pe=pos, k_pe_rows=T, kv_shared=ks)
                queries = gemm(att, *ta["out"], res=queries)
                layer_norm(queries, *L["n2"], out=queries)
                hid = gemm(queries, *L["lin1"], act=ACT_RELU)
                queries = gemm(hid, *L["lin2"], res=queries)
                layer_norm(queries, *L["n3"], out=queries)
                ia = L["i2t"]                                                                # image attending to the tokens
                if FUSED_I2T and Tk <= 16 and ia["q"][0].shape[0] == 128:
                    tok_k = gemm(queries, *ia["k"], a2=qpe)
                    tok_v = gemm(queries, *ia["v"])
                    keys = i2t_block(keys, ks, pos, ia["q"], tok_k, tok_v, ia["out"], L["n4"], pc, Tk)
                    ks = False
                    del tok_k, tok_v, hid
                    continue
                att = self._attn_block(ia, keys, queries, queries, pc, T, Tk, q_pe=pos, q_pe_rows=T, k_pe=qpe, q_shared=ks)
                # keys + attention: per prompt from here on (in place once the stream is per prompt)
                keys = gemm(att, *ia["out"], res=keys, res_rows=T if ks else 0, out=None if ks else keys)
                ks = False
                layer_norm(keys, *L["n4"], out=keys)
                del att, hid
            fa = w["final"]
            att = self._attn_block(fa, queries, keys, keys, pc, Tk, T, q_pe=qpe, k_pe=pos, k_pe_rows=T)
            queries = gemm(att, *fa["out"], res=queries)
            layer_norm(queries, *w["norm_final"], out=queries)
            del att
            # heads: rows (prompt, token i) of `queries` are a strided row set (stride Tk * 256)
            hyper = torch.empty((pc, 4, 32), dtype=torch.float32, device=dev)
            for i in range(4):
                t = gemm(queries, *w["hyper"][i][0], act=ACT_RELU, rows=pc, lda=Tk * PROMPT_DIM, a_offset=(1 + i) * PROMPT_DIM)
                t = gemm(t, *w["hyper"][i][1], act=ACT_RELU)
                gemm(t, *w["hyper"][i][2], out=hyper[:, i, :])
            t = gemm(queries, *w["iou"][0], act=ACT_RELU, rows=pc, lda=Tk * PROMPT_DIM, a_offset=0)
            t = gemm(t, *w["iou"][1], act=ACT_RELU)
            iou4 = gemm(t, *w["iou"][2])
            iou[p0:p0 + pc] = iou4[:, mask0:mask0 + nc]
            # output_upscaling: ConvT 2x2 - LayerNorm2d - GELU - ConvT 2x2 - GELU on token-major rows, then hyper_in @ upscaled
            up1 = gemm(keys, *w["up1"])                                                      # [pc*4096, 4*64] = [(pc*4096*4), 64]
            if split_active() and FUSED_UP2:
                # LayerNorm2d + GELU + the second transposed convolution + GELU + the hyper product: one pass over the first-stage stream
                del keys
                q = _lib.SUp2Params()
                q.u1, q.ln_weight, q.ln_bias, q.ln_eps = up1.data_ptr(), w["up_ln"][0].data_ptr(), w["up_ln"][1].data_ptr(), float(w["up_ln"][2])
                q.w2, q.b2, q.w_scale = w["up2"][0].data_ptr(), w["up2"][1].data_ptr(), weight_scale(w["up2"][0])
                q.hyper, q.hyper_ld, q.mask0, q.nmask, q.low_res, q.P = hyper.data_ptr(), 32, mask0, nc, low[p0:p0 + pc].data_ptr(), pc
                _lib.check(lib.msam_strict_upscale2(C.byref(q), _lib.stream_ptr()), "msam_strict_upscale2")
                del up1
                continue
            layer_norm(up1, *w["up_ln"], out=up1, gelu=True, rows=pc * T * 4, dim=64)
            up2 = gemm(up1.view(pc * T * 4, 64), *w["up2"], act=ACT_GELU)                    # [(pc*4096*4), 4*32]
            del up1, keys
            _lib.check(lib.msam_strict_hyper_masks(up2.data_ptr(), hyper.data_ptr(), 32, mask0, nc, pc, low[p0:p0 + pc].data_ptr(),
                                                   _lib.stream_ptr()), "msam_strict_hyper_masks")
            del up2
        return low, iou
