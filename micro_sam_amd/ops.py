"""Thin Python wrappers over the operator-level C ABI (tensors in, tensors out; torch only allocates)."""
from __future__ import annotations

import ctypes as C
from typing import Any, Dict, List, Optional, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import ACT_GELU, ACT_NONE, ACT_RELU, BF16, F16, F32  # noqa: F401


def _dec16(*tensors) -> torch.dtype:
    """The decoder kernels take their 16-bit operands in the decoder type of this library build (``_lib.decoder_dtype()``:
    fp16 by default); returns it after checking the given operands."""
    dt = _lib.decoder_dtype()
    for t in tensors:
        if t is not None and t.dtype != dt:
            raise TypeError(f"micro_sam_amd: decoder kernels of this build take {dt} operands, got {t.dtype}")
    return dt


def gemm(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, *, act: int = ACT_NONE,
         out_dtype: torch.dtype = torch.float32, resid: Optional[torch.Tensor] = None, resid_rows: int = 0,
         table: Optional[torch.Tensor] = None, table_cols: int = 0, use_glds: int = 0,
         out: Optional[torch.Tensor] = None, ln_mode: int = 0, ln_w: Optional[torch.Tensor] = None,
         ln_b: Optional[torch.Tensor] = None, ln_eps: float = 1e-5, split_k: int = 0) -> torch.Tensor:
    """act(LN(a[M,K] @ w[N,K]^T + bias + table[row % rows, :table_cols] + resid)); a, w bf16.
    ln_mode 1: LayerNorm over the row (N == 256); 2: LayerNorm over 64-column groups + GELU.
    split_k > 1: the contraction in split_k slices accumulated with fp32 atomics (plain fp32 output, K % (64 * split_k) == 0)."""
    _lib.require_gpu()
    assert a.dtype == w.dtype and a.dtype in (torch.bfloat16, torch.float16) and a.is_contiguous() and w.is_contiguous()
    M, K = a.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype, device=a.device)
    p = _lib.GemmParams()
    p.A, p.lda, p.W, p.ldw, p.M, p.N, p.K = a.data_ptr(), K, w.data_ptr(), K, M, N, K
    if a.dtype == torch.float16:
        p.a_dtype = F16
    p.bias = _lib.ptr(bias)
    if table is not None:
        p.table, p.table_rows, p.table_cols, p.table_ld = table.data_ptr(), table.shape[0], table_cols, table.shape[1]
    if resid is not None:
        p.resid = resid.data_ptr()
        p.resid_dtype = {torch.float32: F32, torch.bfloat16: BF16, torch.float16: F16}[resid.dtype]
        p.resid_rows, p.ldr = resid_rows, resid.shape[1]
    p.act = act
    p.out, p.out_dtype, p.ldc = out.data_ptr(), {torch.float32: F32, torch.bfloat16: BF16, torch.float16: F16}[out.dtype], N
    p.use_glds = use_glds
    p.split_k = int(split_k)
    if ln_mode:
        p.ln_mode, p.ln_w, p.ln_b, p.ln_eps = ln_mode, ln_w.data_ptr(), ln_b.data_ptr(), ln_eps
    _lib.check(_lib.load().msam_gemm_bf16(C.byref(p), _lib.stream_ptr()), "msam_gemm_bf16")
    return out


# ---- fp8 (OCP e4m3) encoder path, BASELINE config 5 -------------------------------------------------------------------

def quant_rows_fp8(x: torch.Tensor):
    """bf16 [rows, dim] -> (fp8 e4m3 [rows, dim], fp32 row scales [rows]) with q = round(x * 448 / amax(row))."""
    _lib.require_gpu()
    assert x.dtype == torch.bfloat16 and x.is_contiguous() and x.dim() == 2
    rows, dim = x.shape
    out = torch.empty((rows, dim), dtype=torch.float8_e4m3fn, device=x.device)
    scale = torch.empty((rows,), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().msam_quant_rows_fp8(x.data_ptr(), rows, dim, out.data_ptr(), scale.data_ptr(), _lib.stream_ptr()),
               "msam_quant_rows_fp8")
    return out, scale


def layernorm_fp8(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, eps: float = 1e-6):
    """LayerNorm of fp32 [rows, dim] (dim in 768 / 1024 / 1280) straight to fp8 rows + row scales."""
    _lib.require_gpu()
    assert x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 2
    rows, dim = x.shape
    out = torch.empty((rows, dim), dtype=torch.float8_e4m3fn, device=x.device)
    scale = torch.empty((rows,), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().msam_layernorm_fp8(x.data_ptr(), weight.data_ptr(), bias.data_ptr(), float(eps), rows, dim,
                                              out.data_ptr(), scale.data_ptr(), _lib.stream_ptr()), "msam_layernorm_fp8")
    return out, scale


def quant_weight_fp8(w: torch.Tensor):
    """fp32 / bf16 weight [N, K] -> (fp8 e4m3 [N, K], fp32 per-output-channel scales [N]); torch's e4m3fn is the OCP format
    of gfx950.  Host-side preparation (done once per model)."""
    wf = w.detach().float()
    amax = wf.abs().amax(dim=1)
    scale = torch.where(amax > 0, amax / 448.0, torch.ones_like(amax))
    return (wf / scale[:, None]).to(torch.float8_e4m3fn).contiguous(), scale.contiguous()


def gemm_fp8(a8: torch.Tensor, a_scale: torch.Tensor, w8: torch.Tensor, w_scale: torch.Tensor,
             bias: Optional[torch.Tensor] = None, *, act: int = ACT_NONE, out_dtype: torch.dtype = torch.float32,
             resid: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """act((a8[M,K] @ w8[N,K]^T) * a_scale[:, None] * w_scale[None, :] + bias + resid) on the MX fp8 MFMA (unit block scales)."""
    _lib.require_gpu()
    assert a8.dtype == torch.float8_e4m3fn and w8.dtype == torch.float8_e4m3fn and a8.is_contiguous() and w8.is_contiguous()
    M, K = a8.shape
    N = w8.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype, device=a8.device)
    p = _lib.GemmParams()
    p.A, p.lda, p.W, p.ldw, p.M, p.N, p.K = a8.data_ptr(), K, w8.data_ptr(), K, M, N, K
    p.bias = _lib.ptr(bias)
    if resid is not None:
        assert resid.dtype == torch.float32
        p.resid, p.resid_dtype, p.resid_rows, p.ldr = resid.data_ptr(), F32, 0, resid.shape[1]
    p.act = act
    p.out, p.out_dtype, p.ldc = out.data_ptr(), (F32 if out.dtype == torch.float32 else BF16), N
    p.a_dtype, p.row_scale, p.col_scale = _lib.FP8, a_scale.data_ptr(), w_scale.data_ptr()
    _lib.check(_lib.load().msam_gemm_bf16(C.byref(p), _lib.stream_ptr()), "msam_gemm_bf16(fp8)")
    return out


def gemm_qkv(a: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, B: int, heads: int, use_glds: int = 0):
    """QKV projection with the ViT attention layout epilogue: returns q, k, v as bf16 [B,heads,tokens,hd]."""
    M, K = a.shape
    N = w.shape[0]
    tokens, hd = M // B, N // 3 // heads
    q, k, v = (torch.empty((B, heads, tokens, hd), dtype=torch.bfloat16, device=a.device) for _ in range(3))
    p = _lib.GemmParams()
    p.A, p.lda, p.W, p.ldw, p.M, p.N, p.K = a.data_ptr(), K, w.data_ptr(), K, M, N, K
    p.bias = bias.data_ptr()
    p.out_mode, p.q, p.k, p.v = 1, q.data_ptr(), k.data_ptr(), v.data_ptr()
    p.heads, p.head_dim, p.tokens, p.use_glds = heads, hd, tokens, use_glds
    _lib.check(_lib.load().msam_gemm_bf16(C.byref(p), _lib.stream_ptr()), "msam_gemm_bf16(qkv)")
    return q, k, v


def gemm_kv(a: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, table: torch.Tensor, tokens: int, use_glds: int = 0):
    """Decoder K|V projection (N == 256): k bf16 [M,128], vT bf16 [M/tokens,128,tokens]."""
    M, K = a.shape
    k = torch.empty((M, 128), dtype=torch.bfloat16, device=a.device)
    vT = torch.empty((M // tokens, 128, tokens), dtype=torch.bfloat16, device=a.device)
    p = _lib.GemmParams()
    p.A, p.lda, p.W, p.ldw, p.M, p.N, p.K = a.data_ptr(), K, w.data_ptr(), K, M, 256, K
    p.bias = bias.data_ptr()
    p.table, p.table_rows, p.table_cols, p.table_ld = table.data_ptr(), table.shape[0], 128, table.shape[1]
    p.out_mode, p.k, p.v, p.tokens, p.use_glds = 2, k.data_ptr(), vT.data_ptr(), tokens, use_glds
    _lib.check(_lib.load().msam_gemm_bf16(C.byref(p), _lib.stream_ptr()), "msam_gemm_bf16(kv)")
    return k, vT


def layernorm(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, eps: float, out_dtype=torch.float32,
              gelu: bool = False, nchw_hw: int = 0) -> torch.Tensor:
    rows, dim = x.shape
    if nchw_hw:
        out = torch.empty((rows // nchw_hw, dim, nchw_hw), dtype=torch.float32, device=x.device)
    else:
        out = torch.empty((rows, dim), dtype=out_dtype, device=x.device)
    _lib.check(_lib.load().msam_layernorm(x.data_ptr(), weight.data_ptr(), bias.data_ptr(), eps, rows, dim, out.data_ptr(),
                                          F32 if out.dtype == torch.float32 else BF16, int(gelu), nchw_hw,
                                          _lib.stream_ptr()), "msam_layernorm")
    return out


def cast_transpose(x: torch.Tensor, want16: bool = True, want_t: bool = True, want_sum: bool = False):
    """x [M, K] fp32 / bf16 with contiguous rows -> (bf16 copy [M, K] | None, bf16 transpose [K, M] | None, fp32 column sums [K] | None)
    in one pass (msam_cast_transpose): the operands of a weight gradient dW = dY^T X and the bias gradient."""
    _lib.require_gpu()
    assert x.dim() == 2 and x.stride(1) == 1 and x.dtype in (torch.float32, torch.bfloat16)
    M, K = x.shape
    o16 = torch.empty((M, K), dtype=torch.bfloat16, device=x.device) if want16 else None
    oT = torch.empty((K, M), dtype=torch.bfloat16, device=x.device) if want_t else None
    cs = torch.zeros((K,), dtype=torch.float32, device=x.device) if want_sum else None
    _lib.check(_lib.load().msam_cast_transpose(x.data_ptr(), F32 if x.dtype == torch.float32 else BF16, M, K, x.stride(0),
                                               o16.data_ptr() if want16 else None, oT.data_ptr() if want_t else None,
                                               cs.data_ptr() if want_sum else None, _lib.stream_ptr()), "msam_cast_transpose")
    return o16, oT, cs


def to_image(x: torch.Tensor) -> torch.Tensor:
    """``util._to_image`` on the device (msam_to_image): [H,W] / [H,W,C] uint8 / uint16 (as int16 / uint16 storage) / float32
    device tensor -> uint8 [H,W,3], bit-identical to the host formula."""
    _lib.require_gpu()
    if x.dim() == 2:
        x = x[..., None]
    if x.dim() != 3:
        raise ValueError(f"Invalid input dimensionality {x.dim()}. Expect either a 2D input (=grayscale image) "
                         "or a 3D input (= image with channels).")
    if x.dtype == torch.uint8:
        dt = _lib.U8
    elif x.dtype in (torch.uint16,):
        dt = _lib.U16
    else:
        x, dt = x.to(torch.float32), F32
    x = x.contiguous()
    H, W, Cc = x.shape
    out = torch.empty((H, W, 3), dtype=torch.uint8, device=x.device)
    ws = torch.empty((8,), dtype=torch.int32, device=x.device)
    _lib.check(_lib.load().msam_to_image(x.data_ptr(), dt, H, W, Cc, out.data_ptr(), ws.data_ptr(), _lib.stream_ptr()),
               "msam_to_image")
    return out


_RESAMPLE_TABLES: Dict[Any, Tuple[torch.Tensor, torch.Tensor]] = {}


def resize_bilinear_u8(images: torch.Tensor, newh: int, neww: int) -> torch.Tensor:
    """``ResizeLongestSide.apply_image`` on the device: Pillow's BILINEAR resize of uint8 [B,H,W,C] images (horizontal pass into an
    8-bit intermediate, then vertical; a pass is skipped when that size does not change), bit-identical to
    ``np.array(Image.fromarray(img).resize((neww, newh), Image.BILINEAR))`` (msam_resample_u8; tables cached per size and device)."""
    _lib.require_gpu()
    from .transforms import pil_bilinear_tables
    assert images.dtype == torch.uint8 and images.dim() == 4
    x = images.contiguous()
    B, H, W, Cc = x.shape
    lib = _lib.load()

    def tables(n_in, n_out):
        key = (n_in, n_out, x.device.index)
        t = _RESAMPLE_TABLES.get(key)
        if t is None:
            b, c = pil_bilinear_tables(n_in, n_out)
            t = (torch.as_tensor(b).to(x.device).contiguous(), torch.as_tensor(c).to(x.device).contiguous())
            _RESAMPLE_TABLES[key] = t
        return t
    if neww != W:
        b, c = tables(W, neww)
        out = torch.empty((B, H, neww, Cc), dtype=torch.uint8, device=x.device)
        _lib.check(lib.msam_resample_u8(x.data_ptr(), B, H, W, Cc, 1, neww, b.data_ptr(), c.data_ptr(), int(c.shape[1]), out.data_ptr(),
                                        _lib.stream_ptr()), "msam_resample_u8")
        x, W = out, neww
    if newh != H:
        b, c = tables(H, newh)
        out = torch.empty((B, newh, W, Cc), dtype=torch.uint8, device=x.device)
        _lib.check(lib.msam_resample_u8(x.data_ptr(), B, H, W, Cc, 0, newh, b.data_ptr(), c.data_ptr(), int(c.shape[1]), out.data_ptr(),
                                        _lib.stream_ptr()), "msam_resample_u8")
        x = out
    return x


def patchify(img: torch.Tensor) -> torch.Tensor:
    B = img.shape[0]
    out = torch.empty((B * 4096, 768), dtype=torch.bfloat16, device=img.device)
    _lib.check(_lib.load().msam_patchify(img.data_ptr(), B, out.data_ptr(), _lib.stream_ptr()), "msam_patchify")
    return out


def patchify_u8(img: torch.Tensor) -> torch.Tensor:
    B, h, w = img.shape[:3]
    out = torch.empty((B * 4096, 768), dtype=torch.bfloat16, device=img.device)
    _lib.check(_lib.load().msam_patchify_u8(img.data_ptr(), B, h, w, out.data_ptr(), _lib.stream_ptr()), "msam_patchify_u8")
    return out


def im2col3x3(x: torch.Tensor) -> torch.Tensor:
    B, _, _, Cc = x.shape
    out = torch.empty((B * 4096, 9 * Cc), dtype=torch.bfloat16, device=x.device)
    _lib.check(_lib.load().msam_im2col3x3(x.data_ptr(), B, Cc, out.data_ptr(), _lib.stream_ptr()), "msam_im2col3x3")
    return out


def _enc16(*tensors) -> int:
    """MSAM_BF16 / MSAM_F16 of the encoder-side 16-bit operands (all of one type)."""
    dts = {t.dtype for t in tensors}
    if dts == {torch.bfloat16}:
        return _lib.BF16
    if dts == {torch.float16}:
        return _lib.F16
    raise ValueError(f"expected all-bfloat16 or all-float16 operands, got {sorted(str(d) for d in dts)}")


def window_attention(q, k, v, rel_h, rel_w, qkv_bias, scale: Optional[float] = None) -> torch.Tensor:
    """q, k, v 16 bit (bf16, or fp16: the encoder's fp16 mode) [B,heads,4096,hd] with hd (stored head_dim) 64 or 96; ``scale``
    defaults to hd ** -0.5 (pass the true head_dim's scale for zero-padded heads)."""
    B, heads, _, hd = q.shape
    dt = _enc16(q, k, v, rel_h, rel_w)
    out = torch.empty((B * 4096, heads * hd), dtype=q.dtype, device=q.device)
    _lib.check(_lib.load().msam_window_attention16(q.data_ptr(), k.data_ptr(), v.data_ptr(), rel_h.data_ptr(), rel_w.data_ptr(),
                                                   qkv_bias.data_ptr(), B, heads, hd, float(hd ** -0.5 if scale is None else scale),
                                                   dt, out.data_ptr(), _lib.stream_ptr()),
               "msam_window_attention")
    return out


def global_attention(q, k, v, rel_h, rel_w, scale: Optional[float] = None) -> torch.Tensor:
    B, heads, _, hd = q.shape
    dt = _enc16(q, k, v, rel_h, rel_w)
    out = torch.empty((B * 4096, heads * hd), dtype=q.dtype, device=q.device)
    _lib.check(_lib.load().msam_global_attention16(q.data_ptr(), k.data_ptr(), v.data_ptr(), rel_h.data_ptr(), rel_w.data_ptr(),
                                                   B, heads, hd, float(hd ** -0.5 if scale is None else scale), dt, out.data_ptr(),
                                                   _lib.stream_ptr()), "msam_global_attention")
    return out


def postprocess_masks(low_res: torch.Tensor, input_size: Tuple[int, int], original_size: Tuple[int, int],
                      mask_threshold: float = 0.0, stability_offset: float = 1.0,
                      want_logits: bool = False) -> Dict[str, torch.Tensor]:
    """Fused Sam.postprocess_masks + stability counts + threshold + boxes + bit packing for masks [N,256,256].

    Returns dict(counts int32 [N,3] = (#>thr+off, #>thr-off, #>thr), boxes int32 [N,4] xyxy, bits uint32
    [N, ceil(H/32), W] as int32 storage, logits fp32 [N,H,W] when requested)."""
    _lib.require_gpu()
    # fp16 low-res logits (the AMG path's hand-over from the decoder) are read as they are: widened on load
    low_res = low_res.contiguous() if low_res.dtype == torch.float16 else low_res.to(torch.float32).contiguous()
    low_dt = _lib.F16 if low_res.dtype == torch.float16 else _lib.F32
    N = low_res.shape[0]
    H, W = int(original_size[0]), int(original_size[1])
    dev = low_res.device
    counts = torch.empty((N, 3), dtype=torch.int32, device=dev)
    boxes = torch.empty((N, 4), dtype=torch.int32, device=dev)
    bits = torch.empty((N, (H + 31) // 32, W), dtype=torch.int32, device=dev)
    logits = torch.empty((N, H, W), dtype=torch.float32, device=dev) if want_logits else None
    lib = _lib.load()
    step = 65535
    for s in range(0, N, step):
        n = min(step, N - s)
        _lib.check(lib.msam_postprocess_masks16(
            low_res[s:].data_ptr(), low_dt, n, int(input_size[0]), int(input_size[1]), H, W, float(mask_threshold),
            float(stability_offset), counts[s:].data_ptr(), boxes[s:].data_ptr(), bits[s:].data_ptr(),
            None if logits is None else logits[s:].data_ptr(), _lib.stream_ptr()), "msam_postprocess_masks")
    out = {"counts": counts, "boxes": boxes, "bits": bits}
    if want_logits:
        out["logits"] = logits
    return out


def rle_encode(bits: torch.Tensor, height: int, width: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Column-major uncompressed RLE of bit masks [N, ceil(H/32), W].  Returns (counts int32 [total], offsets int64 [N+1])."""
    N = bits.shape[0]
    lib = _lib.load()
    n_runs = torch.empty((N,), dtype=torch.int32, device=bits.device)
    _lib.check(lib.msam_rle_run_counts(bits.data_ptr(), N, height, width, n_runs.data_ptr(), _lib.stream_ptr()),
               "msam_rle_run_counts")
    offsets = torch.zeros((N + 1,), dtype=torch.int64, device=bits.device)
    torch.cumsum(n_runs, 0, out=offsets[1:])
    total = int(offsets[-1].item())
    counts = torch.empty((total,), dtype=torch.int32, device=bits.device)
    _lib.check(lib.msam_rle_encode(bits.data_ptr(), N, height, width, offsets.data_ptr(), counts.data_ptr(),
                                   _lib.stream_ptr()), "msam_rle_encode")
    return counts, offsets


def rles_to_list(counts: torch.Tensor, offsets: torch.Tensor, height: int, width: int,
                 as_list: bool = True) -> List[Dict[str, Any]]:
    """Device RLE buffers -> the reference's list-of-dicts format ({"size": [h, w], "counts": [...]}).

    ``as_list=False`` keeps every ``counts`` as an int32 numpy view into one host buffer (no per-element Python
    objects); ``rle_to_mask`` / ``area_from_rle`` / pickling work on both."""
    c = counts.cpu().numpy()
    o = offsets.cpu().numpy()
    if as_list:
        return [{"size": [height, width], "counts": c[o[i]:o[i + 1]].tolist()} for i in range(len(o) - 1)]
    return [{"size": [height, width], "counts": c[o[i]:o[i + 1]]} for i in range(len(o) - 1)]


def unpack_bits(bits: torch.Tensor, height: int) -> torch.Tensor:
    """bit masks [N, ceil(H/32), W] -> bool [N,H,W] (test / binary_mask output helper; torch ops only)."""
    N, wpc, W = bits.shape
    sh = torch.arange(32, device=bits.device, dtype=torch.int32).view(1, 1, 32, 1)
    m = ((bits.unsqueeze(2) >> sh) & 1).to(torch.bool).reshape(N, wpc * 32, W)
    return m[:, :height]


def paint_label_image(bits: torch.Tensor, order: torch.Tensor, height: int, width: int) -> torch.Tensor:
    """label[y,x] = r+1 of the last mask bits[order[r]] covering the pixel (0 if none): int32 [H,W]."""
    label = torch.empty((height, width), dtype=torch.int32, device=bits.device)
    order = order.to(device=bits.device, dtype=torch.int32).contiguous()
    _lib.check(_lib.load().msam_paint_label_image(bits.data_ptr() if order.numel() else None,
                                                  order.data_ptr() if order.numel() else None, order.numel(), height, width,
                                                  label.data_ptr(), _lib.stream_ptr()), "msam_paint_label_image")
    return label


def label_components(seg: torch.Tensor) -> torch.Tensor:
    """4-connected components of equal non-zero value of an int32 [H,W] image: per pixel the KEY of its component's root,
    -1 for background, int32 [H*W].  Keys are positions in block-major order (512 x 512 blocks in raster order, raster order inside
    a block: csrc/common.h bm_key) and the root is the component's smallest key - so ascending root keys are the component numbers
    of the reference's ``elf.parallel.label(block_shape=(512, 512))`` (util.py:1834); for H, W <= 512 a key is the linear index."""
    h, w = seg.shape
    seg = seg.contiguous()
    roots = torch.empty((h * w,), dtype=torch.int32, device=seg.device)
    flag = torch.zeros((1,), dtype=torch.int32, device=seg.device)
    iters = C.c_int32(0)
    _lib.check(_lib.load().msam_label_components(seg.data_ptr(), h, w, roots.data_ptr(), flag.data_ptr(), 16, C.byref(iters),
                                                 _lib.stream_ptr()), "msam_label_components")
    return roots


def box_nms(boxes: torch.Tensor, scores: torch.Tensor, iou_threshold: float) -> torch.Tensor:
    """Greedy NMS on the device (torchvision.ops.nms semantics): kept indices in descending score order (int64)."""
    k = int(boxes.shape[0])
    if k == 0:
        return torch.empty((0,), dtype=torch.int64, device=boxes.device)
    order = torch.sort(scores.float(), descending=True, stable=True).indices
    b = boxes.float()[order].contiguous()
    nblk = (k + 63) // 64
    scratch = torch.empty((k * nblk,), dtype=torch.int64, device=boxes.device)
    keep = torch.empty((k,), dtype=torch.int32, device=boxes.device)
    _lib.check(_lib.load().msam_box_nms(b.data_ptr(), k, float(iou_threshold), scratch.data_ptr(), keep.data_ptr(),
                                        _lib.stream_ptr()), "msam_box_nms")
    return order[keep.bool()]


def mask_nms(bits: torch.Tensor, boxes_xyxy: torch.Tensor, areas: torch.Tensor, scores: torch.Tensor, thresh: float,
             height: int, intersection_over_min: bool = False) -> torch.Tensor:
    """Greedy mask NMS on bit masks [K, ceil(H/32), W] (msam_mask_nms; reference ``util._batched_mask_nms``): returns the kept
    mask indices in descending score order (stable order for equal scores)."""
    _lib.require_gpu()
    k = int(bits.shape[0])
    if k == 0:
        return torch.zeros((0,), dtype=torch.int64, device=bits.device)
    dev = bits.device
    order = torch.sort(scores.to(dev).float(), descending=True, stable=True).indices
    order32 = order.to(torch.int32).contiguous()
    nblk = (k + 63) // 64
    scratch = torch.empty((k * nblk,), dtype=torch.int64, device=dev)
    keep_sorted = torch.empty((k,), dtype=torch.int32, device=dev)
    _lib.check(_lib.load().msam_mask_nms(bits.contiguous().data_ptr(), order32.data_ptr(), boxes_xyxy.to(dev).float().contiguous().data_ptr(),
                                         areas.to(dev).to(torch.int32).contiguous().data_ptr(), k, int(height), int(bits.shape[2]),
                                         float(thresh), int(bool(intersection_over_min)), scratch.data_ptr(), keep_sorted.data_ptr(),
                                         _lib.stream_ptr()), "msam_mask_nms")
    return order[keep_sorted.bool()]


def wsgemm(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, *, table: Optional[torch.Tensor] = None,
           table_cols: int = 0, resid: Optional[torch.Tensor] = None, resid_rows: int = 0, ln_mode: int = 0,
           ln_w: Optional[torch.Tensor] = None, ln_b: Optional[torch.Tensor] = None, ln_eps: float = 1e-5,
           kv_split_tokens: int = 0, out: Optional[torch.Tensor] = None, head_major_tokens: int = 0):
    """Weights-stationary decoder GEMM (see include/msam_hip.h).  Returns out [M,N] in the decoder's 16-bit type, or (k, vT)
    with kv_split_tokens."""
    _lib.require_gpu()
    d16 = _dec16(a, w, resid)
    M, K = a.shape
    N = w.shape[0]
    p = _lib.WsGemmParams()
    p.A, p.W, p.M, p.N, p.K = a.data_ptr(), w.data_ptr(), M, N, K
    p.bias = _lib.ptr(bias)
    if table is not None:
        p.table, p.table_rows, p.table_cols, p.table_ld = table.data_ptr(), table.shape[0], table_cols, table.shape[1]
    if resid is not None:
        p.resid, p.resid_rows, p.ldr = resid.data_ptr(), resid_rows, resid.shape[1]
    if ln_mode:
        p.ln_mode, p.ln_w, p.ln_b, p.ln_eps = ln_mode, ln_w.data_ptr(), ln_b.data_ptr(), ln_eps
    ret = None
    if kv_split_tokens:
        k = torch.empty((M, 128), dtype=d16, device=a.device)
        vT = torch.empty((M // kv_split_tokens, 128, kv_split_tokens), dtype=d16, device=a.device)
        p.kv_split, p.k_out, p.vT_out, p.tokens = 1, k.data_ptr(), vT.data_ptr(), kv_split_tokens
        ret = (k, vT)
    else:
        if out is None:
            out = torch.empty((M, N), dtype=d16, device=a.device)
        p.out, p.ldc = out.data_ptr(), N
        if head_major_tokens:
            p.head_major, p.tokens = 1, head_major_tokens
        ret = out
    _lib.check(_lib.load().msam_wsgemm_bf16(C.byref(p), _lib.stream_ptr()), "msam_wsgemm_bf16")
    return ret


def box_nms_flags(boxes: torch.Tensor, scores: torch.Tensor, valid: torch.Tensor, iou_threshold: float) -> torch.Tensor:
    """Greedy NMS restricted to ``valid`` boxes, without any host synchronisation: returns bool keep flags [N] in the
    ORIGINAL order (boxes with valid == False are never kept and never suppress)."""
    k = int(boxes.shape[0])
    if k == 0:
        return torch.zeros((0,), dtype=torch.bool, device=boxes.device)
    s = torch.where(valid, scores.float(), torch.full_like(scores, float("-inf"), dtype=torch.float32))
    order = torch.sort(s, descending=True, stable=True).indices
    b = boxes.float()[order].contiguous()
    v = valid[order].to(torch.int32).contiguous()
    nblk = (k + 63) // 64
    scratch = torch.empty((k * nblk,), dtype=torch.int64, device=boxes.device)
    keep_sorted = torch.empty((k,), dtype=torch.int32, device=boxes.device)
    _lib.check(_lib.load().msam_box_nms_valid(b.data_ptr(), v.data_ptr(), k, float(iou_threshold), scratch.data_ptr(),
                                              keep_sorted.data_ptr(), _lib.stream_ptr()), "msam_box_nms_valid")
    keep = torch.zeros((k,), dtype=torch.bool, device=boxes.device)
    keep[order] = keep_sorted.bool()
    return keep


_AMG_WS: Dict[Any, torch.Tensor] = {}


def amg_generate_labels(iou: torch.Tensor, stability: torch.Tensor, boxes: torch.Tensor, area: torch.Tensor, bits: torch.Tensor,
                        shape: Tuple[int, int], crop_box, pred_iou_thresh: float, stability_score_thresh: float,
                        box_nms_thresh: float, min_object_size: int = 0, with_background: bool = True):
    """``generate(output_mode="instance_segmentation")`` of a single-crop device state in one library call
    (msam_amg_generate_labels: filters, box NMS, paint, connected components, relabel; N <= 4096 candidates).
    Returns (labels int32 [H, W], flag int32 [1] that reads 0 when the component labelling converged)."""
    _lib.require_gpu()
    lib = _lib.load()
    h, w = int(shape[0]), int(shape[1])
    n = int(iou.shape[0])
    dev = iou.device
    need = int(lib.msam_amg_generate_workspace_bytes(n, h, w))
    if need <= 0:
        raise ValueError(f"amg_generate_labels: 1 <= N <= 4096 candidates, got {n}")
    # one workspace per (device, stream): generate() of tile i runs on a side stream while tile i+1 is decoded
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
    ws = _AMG_WS.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.uint8, device=dev)
        _AMG_WS[key] = ws
    labels = torch.empty((h, w), dtype=torch.int32, device=dev)
    flag = torch.empty((1,), dtype=torch.int32, device=dev)
    cb = (C.c_int32 * 4)(*[int(v) for v in crop_box])
    _lib.check(lib.msam_amg_generate_labels(
        iou.float().contiguous().data_ptr(), stability.float().contiguous().data_ptr(), boxes.to(torch.int32).contiguous().data_ptr(),
        area.to(torch.int32).contiguous().data_ptr(), bits.contiguous().data_ptr(), n, h, w, cb, float(pred_iou_thresh),
        float(stability_score_thresh), float(box_nms_thresh), int(min_object_size), int(bool(with_background)),
        labels.data_ptr(), flag.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_ptr()), "msam_amg_generate_labels")
    return labels, flag


def labels_from_masks(bits: torch.Tensor, order: torch.Tensor, shape: Tuple[int, int], k_dev: Optional[torch.Tensor] = None,
                      min_object_size: int = 0, with_background: bool = False):
    """``util.mask_data_to_segmentation(label_masks=True, merge_exclusively=False)`` of selected masks in one library call
    (msam_labels_from_masks: paint in ``order`` - later masks overwrite -, connected components in the reference's numbering, size /
    background filter, consecutive relabel).  ``order`` int32 [K] indexes ``bits``; with ``k_dev`` (int32[1] on the device) only its
    first k_dev[0] entries are painted.  Returns (labels int32 [H, W], flag int32[1]: 0 = the labelling converged); no host sync."""
    _lib.require_gpu()
    lib = _lib.load()
    h, w = int(shape[0]), int(shape[1])
    dev = bits.device
    need = int(lib.msam_labels_from_masks_workspace_bytes(h, w))
    key = ("lfm", dev.index, torch.cuda.current_stream(dev).cuda_stream)
    ws = _AMG_WS.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.uint8, device=dev)
        _AMG_WS[key] = ws
    labels = torch.empty((h, w), dtype=torch.int32, device=dev)
    flag = torch.empty((1,), dtype=torch.int32, device=dev)
    order = order.to(torch.int32).contiguous()
    _lib.check(lib.msam_labels_from_masks(bits.contiguous().data_ptr(), order.data_ptr(), int(order.numel()), _lib.ptr(k_dev), h, w,
                                          int(min_object_size), int(bool(with_background)), labels.data_ptr(), flag.data_ptr(),
                                          ws.data_ptr(), ws.numel(), _lib.stream_ptr()), "msam_labels_from_masks")
    return labels, flag


def paint_label_image_dev(bits: torch.Tensor, order: torch.Tensor, k_dev: torch.Tensor, height: int, width: int) -> torch.Tensor:
    """paint_label_image with the mask count taken from device memory (k_dev int32[1]); order int32 [N]."""
    label = torch.empty((height, width), dtype=torch.int32, device=bits.device)
    _lib.check(_lib.load().msam_paint_label_image_dev(bits.data_ptr(), order.data_ptr(), k_dev.data_ptr(), height, width,
                                                      label.data_ptr(), _lib.stream_ptr()), "msam_paint_label_image_dev")
    return label


def label_components_async(seg: torch.Tensor, passes: int = 2):
    """label_components without host synchronisation: (roots int32 [H*W], changed_flag int32[1] of the last pass)."""
    h, w = seg.shape
    seg = seg.contiguous()
    roots = torch.empty((h * w,), dtype=torch.int32, device=seg.device)
    flag = torch.zeros((1,), dtype=torch.int32, device=seg.device)
    _lib.check(_lib.load().msam_label_components_async(seg.data_ptr(), h, w, roots.data_ptr(), flag.data_ptr(), passes,
                                                       _lib.stream_ptr()), "msam_label_components_async")
    return roots, flag


def slice_overlaps(labels: torch.Tensor) -> np.ndarray:
    """Overlap table between consecutive slices of a device label volume int32 [Z,H,W] (ids consecutive across z):
    int64 [E, 3] rows (source id in slice z, target id in slice z + 1 - 0 = background -, overlapping pixels), sorted by
    (source, target).  The scatter-add of ``nifty.ground_truth.overlap`` behind the reference's ``compute_edges_from_overlap``
    (multi_dimensional_segmentation.py:357) as an open-addressing hash table in HBM (msam_slice_overlaps); the table is
    enlarged and the pass repeated when it overflows."""
    _lib.require_gpu(labels.device)
    assert labels.dim() == 3 and labels.dtype == torch.int32
    labels = labels.contiguous()
    z, h, w = labels.shape
    cap, max_edges = 1 << 18, 1 << 17
    while True:
        keys = torch.empty((cap,), dtype=torch.int64, device=labels.device)
        counts = torch.empty((cap,), dtype=torch.int32, device=labels.device)
        edges = torch.empty((max_edges, 3), dtype=torch.int32, device=labels.device)
        n = torch.empty((2,), dtype=torch.int32, device=labels.device)
        _lib.check(_lib.load().msam_slice_overlaps(labels.data_ptr(), z, h, w, keys.data_ptr(), counts.data_ptr(), cap, edges.data_ptr(),
                                                   max_edges, n.data_ptr(), _lib.stream_ptr()), "msam_slice_overlaps")
        n_edges, overflow = (int(v) for v in n.cpu().tolist())
        if not overflow and n_edges <= max_edges and 4 * n_edges <= cap:
            break
        if cap >= 1 << 28:
            raise RuntimeError("slice_overlaps: more than 2^26 distinct overlapping pairs")
        cap, max_edges = cap * 4, max_edges * 4
    e = edges[:n_edges].cpu().numpy().astype(np.int64)
    return e[np.lexsort((e[:, 1], e[:, 0]))] if n_edges else e.reshape(0, 3)


def component_sizes(roots: torch.Tensor):
    """(sizes int32 [n] keyed by root index, bg_count int32[1]) for roots int32 [n] (-1 = background)."""
    n = roots.numel()
    sizes = torch.empty((n,), dtype=torch.int32, device=roots.device)
    bg = torch.empty((1,), dtype=torch.int32, device=roots.device)
    _lib.check(_lib.load().msam_component_sizes(roots.data_ptr(), n, sizes.data_ptr(), bg.data_ptr(), _lib.stream_ptr()),
               "msam_component_sizes")
    return sizes, bg


def decoder_image_layer(xin, ktok, vtok, wo, bo, ln_w, ln_b, Nt, *, q_shared=None, wq=None, bq=None, peq=None,
                        rows=None, ln_eps: float = 1e-5, out=None):
    """Fused image-side half of a two-way block (include/msam_hip.h msam_decoder_image_layer)."""
    _lib.require_gpu()
    rows = xin.shape[0] if rows is None else rows
    if out is None:
        out = torch.empty((rows, 256), dtype=_dec16(xin, ktok, vtok, wo, wq, q_shared), device=xin.device)
    p = _lib.ImageLayerParams()
    p.xin, p.q_shared = xin.data_ptr(), _lib.ptr(q_shared)
    p.wq, p.bq, p.peq = _lib.ptr(wq), _lib.ptr(bq), _lib.ptr(peq)
    p.wo, p.bo, p.ln_w, p.ln_b, p.ln_eps = wo.data_ptr(), bo.data_ptr(), ln_w.data_ptr(), ln_b.data_ptr(), ln_eps
    p.ktok, p.vtok, p.Nt, p.out, p.rows = ktok.data_ptr(), vtok.data_ptr(), Nt, out.data_ptr(), rows
    _lib.check(_lib.load().msam_decoder_image_layer(C.byref(p), _lib.stream_ptr()), "msam_decoder_image_layer")
    return out


def t2i_fold_attention(keys, qtok, wk, tabk, wv, bv, *, kv_shared: bool = False, blocked: bool = False):
    """Token -> image attention with folded K / V projections (include/msam_hip.h msam_t2i_fold_attention).
    keys [Pk,4096,256], qtok [P,Nt,128] (Nt <= 8), wk / wv [128,256], tabk [4096,128] in the decoder's 16-bit type, bv fp32 [128]
    -> [P,Nt,128]."""
    _lib.require_gpu()
    lib = _lib.load()
    P, Nt = qtok.shape[0], qtok.shape[1]
    nbytes = int(lib.msam_t2i_fold_workspace_bytes(P))
    work = torch.empty((nbytes,), dtype=torch.uint8, device=keys.device)
    out = torch.empty((P, Nt, 128), dtype=_dec16(keys, qtok, wk, tabk, wv), device=keys.device)
    _lib.check(lib.msam_t2i_fold_attention(keys.data_ptr(), 2 if blocked else int(kv_shared), qtok.data_ptr(), P, Nt, wk.data_ptr(),
                                           tabk.data_ptr(), wv.data_ptr(), bv.data_ptr(), out.data_ptr(), work.data_ptr(),
                                           nbytes, _lib.stream_ptr()), "msam_t2i_fold_attention")
    return out


def i2t_fold_layer(xin, ktok, vtok, wq, tabq, wo, bo, ln_w, ln_b, *, x_shared: bool = False, ln_eps: float = 1e-5, out=None):
    """Folded image->token attention + out_proj + residual + LayerNorm (include/msam_hip.h msam_i2t_fold_layer).
    xin [Px,4096,256], ktok / vtok [P,Nt,128] (Nt <= 8) -> [P,4096,256], all in the decoder's 16-bit type."""
    _lib.require_gpu()
    lib = _lib.load()
    P, Nt = ktok.shape[0], ktok.shape[1]
    nbytes = int(lib.msam_i2t_fold_workspace_bytes(P))
    work = torch.empty((nbytes,), dtype=torch.uint8, device=xin.device)
    if out is None:
        out = torch.empty((P, 4096, 256), dtype=_dec16(xin, ktok, vtok, wq, tabq, wo), device=xin.device)
    _lib.check(lib.msam_i2t_fold_layer(xin.data_ptr(), int(x_shared), ktok.data_ptr(), vtok.data_ptr(), P, Nt, wq.data_ptr(),
                                       tabq.data_ptr(), wo.data_ptr(), bo.data_ptr(), ln_w.data_ptr(), ln_b.data_ptr(),
                                       ln_eps, out.data_ptr(), work.data_ptr(), nbytes, _lib.stream_ptr()),
               "msam_i2t_fold_layer")
    return out


def i2t_fold_operands(ktok, vtok, wq, wo, bo, *, with_kfold: bool = True):
    """Per-prompt operands of an image->token layer in MFMA fragment order (include/msam_hip.h msam_i2t_fold_operands):
    ktok / vtok [P,Nt,128], wq [128,256], wo [256,128] in the decoder's 16-bit type, bo fp32 [256] -> uint8 [P * bytes]."""
    _lib.require_gpu()
    lib = _lib.load()
    _dec16(ktok, vtok, wq, wo)
    P, Nt = ktok.shape[0], ktok.shape[1]
    oper = torch.empty((int(lib.msam_i2t_fold_operand_bytes(P)),), dtype=torch.uint8, device=ktok.device)
    _lib.check(lib.msam_i2t_fold_operands(ktok.data_ptr(), vtok.data_ptr(), P, Nt, wq.data_ptr(), wo.data_ptr(), bo.data_ptr(),
                                          int(with_kfold), oper.data_ptr(), _lib.stream_ptr()), "msam_i2t_fold_operands")
    return oper


def to_blocked(x: torch.Tensor) -> torch.Tensor:
    """[..., 4096, W] row-major -> the blocked layout of csrc/decfold_tok.hip, [..., 256 tiles, W / 32 k-steps, 4 lane groups, 16
    tokens, 8 channels] flattened back to [..., 4096, W] (include/msam_hip.h "BLOCKED layout").  Host-side helper for tests / tools."""
    W = x.shape[-1]
    lead = x.shape[:-2]
    n = len(lead)
    y = x.reshape(*lead, 256, 16, W // 32, 4, 8).permute(*range(n), n, n + 2, n + 3, n + 1, n + 4)
    return y.reshape(*lead, 4096, W).contiguous()


def from_blocked(x: torch.Tensor) -> torch.Tensor:
    """Inverse of :func:`to_blocked`."""
    W = x.shape[-1]
    lead = x.shape[:-2]
    n = len(lead)
    y = x.reshape(*lead, 256, W // 32, 4, 16, 8).permute(*range(n), n, n + 3, n + 1, n + 2, n + 4)
    return y.reshape(*lead, 4096, W).contiguous()


def chain_prepare_tables(src, q0, tabk, tabq1):
    """Blocked copies of the shared tables of the chained kernels (include/msam_hip.h msam_chain_prepare_tables):
    src [4096,256], q0 / tabk / tabq1 [4096,128] in the decoder's 16-bit type -> uint8 blob."""
    _lib.require_gpu()
    lib = _lib.load()
    _dec16(src, q0, tabk, tabq1)
    tables = torch.empty((int(lib.msam_chain_tables_bytes()),), dtype=torch.uint8, device=src.device)
    _lib.check(lib.msam_chain_prepare_tables(src.data_ptr(), q0.data_ptr(), tabk.data_ptr(), tabq1.data_ptr(), tables.data_ptr(),
                                             _lib.stream_ptr()), "msam_chain_prepare_tables")
    return tables


def i2t0_t2i_fused(tables, operands0, ln0_w, ln0_b, qtok, wk, wv, bv, *, ln_eps: float = 1e-5):
    """Layer-0 image->token block on the shared source chained into the layer-1 token->image attention
    (include/msam_hip.h msam_i2t0_t2i_fused): tables from chain_prepare_tables, qtok [P,Nt,128] -> [P,Nt,128]."""
    _lib.require_gpu()
    lib = _lib.load()
    P, Nt = qtok.shape[0], qtok.shape[1]
    nbytes = int(lib.msam_i2t0_t2i_workspace_bytes(P))
    work = torch.empty((nbytes,), dtype=torch.uint8, device=qtok.device)
    out = torch.empty((P, Nt, 128), dtype=_dec16(qtok, wk, wv), device=qtok.device)
    _lib.check(lib.msam_i2t0_t2i_fused(tables.data_ptr(), operands0.data_ptr(), ln0_w.data_ptr(), ln0_b.data_ptr(), ln_eps,
                                       qtok.data_ptr(), P, Nt, wk.data_ptr(), wv.data_ptr(), bv.data_ptr(), out.data_ptr(),
                                       work.data_ptr(), nbytes, _lib.stream_ptr()), "msam_i2t0_t2i_fused")
    return out


def chain_prepare_tables2(src, wv, bv, wk, ln0_w, ln0_b, wo0, bo0):
    """Prompt-independent tables of the second form of the chained attention (include/msam_hip.h msam_chain_prepare_tables2)."""
    _lib.require_gpu()
    lib = _lib.load()
    _dec16(src, wv, wk, wo0)
    t2 = torch.empty((int(lib.msam_chain_tables2_bytes()),), dtype=torch.uint8, device=src.device)
    _lib.check(lib.msam_chain_prepare_tables2(src.data_ptr(), wv.data_ptr(), bv.data_ptr(), wk.data_ptr(), ln0_w.data_ptr(),
                                              ln0_b.data_ptr(), wo0.data_ptr(), bo0.data_ptr(), t2.data_ptr(), _lib.stream_ptr()),
               "msam_chain_prepare_tables2")
    return t2


def chain_prepare_tables2_cached(src, wv, bv, wk, ln0_w, ln0_b, wo0, bo0):
    """The same tables through the two-step path the decoder uses: weight-only part once (msam_chain_prepare_const2), then the
    source-dependent rest (msam_chain_prepare_tables2_c).  Returns (tables2, const2)."""
    _lib.require_gpu()
    lib = _lib.load()
    _dec16(src, wv, wk, wo0)
    c2 = torch.empty((int(lib.msam_chain_const2_bytes()),), dtype=torch.uint8, device=src.device)
    _lib.check(lib.msam_chain_prepare_const2(wv.data_ptr(), bv.data_ptr(), wk.data_ptr(), ln0_w.data_ptr(), ln0_b.data_ptr(),
                                             wo0.data_ptr(), bo0.data_ptr(), c2.data_ptr(), _lib.stream_ptr()), "msam_chain_prepare_const2")
    t2 = torch.empty((int(lib.msam_chain_tables2_bytes()),), dtype=torch.uint8, device=src.device)
    _lib.check(lib.msam_chain_prepare_tables2_c(src.data_ptr(), c2.data_ptr(), t2.data_ptr(), _lib.stream_ptr()),
               "msam_chain_prepare_tables2_c")
    return t2, c2


def t2i_fold_values(vtok0, tables2):
    """Per-prompt M fragments of the second form (include/msam_hip.h msam_t2i_fold_values): vtok0 [P,Nt,128]."""
    _lib.require_gpu()
    lib = _lib.load()
    P, Nt = vtok0.shape[0], vtok0.shape[1]
    mf = torch.empty((int(lib.msam_t2i_fold_values_bytes(P)),), dtype=torch.uint8, device=vtok0.device)
    _lib.check(lib.msam_t2i_fold_values(vtok0.data_ptr(), P, Nt, tables2.data_ptr(), mf.data_ptr(), _lib.stream_ptr()),
               "msam_t2i_fold_values")
    return mf


def i2t_fold_operands_values(ktok, vtok, wq, wo, bo, tables2, *, with_kfold: bool = False):
    """:func:`i2t_fold_operands` and :func:`t2i_fold_values` in one launch (include/msam_hip.h msam_i2t_fold_operands_values)
    -> (operands, mf)."""
    _lib.require_gpu()
    lib = _lib.load()
    _dec16(ktok, vtok, wq, wo)
    P, Nt = ktok.shape[0], ktok.shape[1]
    oper = torch.empty((int(lib.msam_i2t_fold_operand_bytes(P)),), dtype=torch.uint8, device=ktok.device)
    mf = torch.empty((int(lib.msam_t2i_fold_values_bytes(P)),), dtype=torch.uint8, device=ktok.device)
    _lib.check(lib.msam_i2t_fold_operands_values(ktok.data_ptr(), vtok.data_ptr(), P, Nt, wq.data_ptr(), wo.data_ptr(), bo.data_ptr(),
                                                 int(with_kfold), tables2.data_ptr(), oper.data_ptr(), mf.data_ptr(),
                                                 _lib.stream_ptr()), "msam_i2t_fold_operands_values")
    return oper, mf


def i2t0_t2i_fused_v2(tables, tables2, operands0, mf, ln0_w, qtok, wk, *, ln_eps: float = 1e-5):
    """Second form of :func:`i2t0_t2i_fused` (include/msam_hip.h msam_i2t0_t2i_fused_v2) -> [P,Nt,128]."""
    _lib.require_gpu()
    lib = _lib.load()
    P, Nt = qtok.shape[0], qtok.shape[1]
    nbytes = int(lib.msam_i2t0_t2i_v2_workspace_bytes(P))
    work = torch.empty((nbytes,), dtype=torch.uint8, device=qtok.device)
    out = torch.empty((P, Nt, 128), dtype=_dec16(qtok, wk), device=qtok.device)
    _lib.check(lib.msam_i2t0_t2i_fused_v2(tables.data_ptr(), tables2.data_ptr(), operands0.data_ptr(), mf.data_ptr(), ln0_w.data_ptr(),
                                          ln_eps, qtok.data_ptr(), P, Nt, wk.data_ptr(), out.data_ptr(), work.data_ptr(), nbytes,
                                          _lib.stream_ptr()), "msam_i2t0_t2i_fused_v2")
    return out


def i2t01_fused(tables, operands0, ln0_w, ln0_b, operands1, ln1_w, ln1_b, P: int, Nt: int, *, ln_eps: float = 1e-5):
    """Layer-0 image->token block on the shared source chained into the layer-1 image->token block
    (include/msam_hip.h msam_i2t01_fused) -> the layer-1 output stream [P,4096,256] in the BLOCKED layout (from_blocked)."""
    _lib.require_gpu()
    out = torch.empty((P, 4096, 256), dtype=_lib.decoder_dtype(), device=tables.device)
    _lib.check(_lib.load().msam_i2t01_fused(tables.data_ptr(), operands0.data_ptr(), ln0_w.data_ptr(), ln0_b.data_ptr(),
                                            operands1.data_ptr(), ln1_w.data_ptr(), ln1_b.data_ptr(), ln_eps, P, Nt,
                                            out.data_ptr(), _lib.stream_ptr()), "msam_i2t01_fused")
    return out


def upscale_fused(keys, w1, b1, ln_w, ln_b, w2, b2, hyper, mask0: int, nmask: int, *, ln_eps: float = 1e-6, blocked: bool = False,
                  centred: bool = False):
    """Fused output up-scaling + hyper-network product (include/msam_hip.h msam_upscale_fused).
    keys [P,4096,256] (decoder 16-bit type), hyper fp32 [P,4,ld] -> fp32 [P,nmask,256,256].
    centred=True: the caller states that w1 / b1 are centred over the 64 output channels of every sub-pixel (upscale_centre_weights):
    LayerNorm2d's mean is zero by construction and the kernel does not compute it (bit 1 of the C entry point's `keys_blocked`)."""
    _lib.require_gpu()
    _dec16(keys, w1, w2)
    P = keys.shape[0]
    out = torch.empty((P, nmask, 256, 256), dtype=torch.float32, device=keys.device)
    _lib.check(_lib.load().msam_upscale_fused_layout(keys.data_ptr(), int(blocked) | (2 if centred else 0), P, w1.data_ptr(), b1.data_ptr(), ln_w.data_ptr(),
                                                     ln_b.data_ptr(), ln_eps, w2.data_ptr(), b2.data_ptr(), hyper.data_ptr(),
                                                     hyper.shape[-1], mask0, nmask, out.data_ptr(), _lib.stream_ptr()),
               "msam_upscale_fused_layout")
    return out


def upscale_centre_weights(w1_f32: torch.Tensor, b1_f32: torch.Tensor):
    """What modeling.Sam hands the decoder (msam_decoder_t.up1_centred = 1): the first up-scaling layer's GEMM weight [4 * 64 (sub-pixel, channel), 256] minus,
    per sub-pixel, the mean of its 64 rows - on the fp32 values, cast to the decoder's 16-bit type afterwards - and the tiled bias [256] minus its mean."""
    w = w1_f32.float().view(4, 64, -1)
    b = b1_f32.float().view(4, 64)
    return ((w - w.mean(1, keepdim=True)).reshape(256, -1).to(_lib.decoder_dtype()).contiguous(),
            (b - b.mean(1, keepdim=True)).reshape(256).contiguous())


def uncrop_bits(bits: torch.Tensor, crop_box, height: int, width: int) -> torch.Tensor:
    """uncrop_masks on bit masks: [N, ceil(ch/32), cw] of crop_box = [x0, y0, x1, y1] -> [N, ceil(H/32), W] (int32 storage)."""
    _lib.require_gpu()
    x0, y0, x1, y1 = (int(v) for v in crop_box)
    n = int(bits.shape[0])
    out = torch.empty((n, (height + 31) // 32, width), dtype=torch.int32, device=bits.device)
    if n == 0:
        return out
    bits = bits.contiguous()
    lib = _lib.load()
    step = 65535
    for s in range(0, n, step):
        m = min(step, n - s)
        _lib.check(lib.msam_uncrop_bits(bits[s:].data_ptr(), m, y1 - y0, x1 - x0, x0, y0, height, width, out[s:].data_ptr(),
                                        _lib.stream_ptr()), "msam_uncrop_bits")
    return out
