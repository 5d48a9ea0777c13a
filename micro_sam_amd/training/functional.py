"""Differentiable primitives of the mask decoder on libmsam_hip.so (``torch.autograd.Function`` wrappers; torch owns memory,
shapes and the tape, every forward / backward product runs a HIP kernel):

* ``linear``      y = x W^T + b on the MFMA GEMM kernel (``msam_gemm_bf16``), bf16 operands / fp32 accumulation / fp32
                  outputs in BOTH directions: dX = dY W, dW = dY^T X (the reference trains under AMP bf16,
                  ``finetuning/specialists/training/light_microscopy/livecell_multi_gpu_finetuning.py:56-82``); parameters
                  and their gradients stay fp32;
* ``layer_norm``  ``msam_layernorm`` / ``msam_layernorm_backward`` (fp32);
* ``attention``   softmax(q k^T / sqrt(d)) v, ``msam_attention_forward`` / ``msam_attention_backward`` (fp32);
* ``relpos_attention``  the image encoder's attention with its decomposed relative position bias on one token grid,
                  ``msam_relpos_attention_forward`` / ``_backward`` (fp32; gradients also for the two bias tensors).

There is no CPU / eager fallback: on a machine without the library these raise.
"""
from __future__ import annotations

import ctypes as C
import math

import os

import torch

from .. import _lib, ops


def _pad_to(t: torch.Tensor, rows: int, cols: int) -> torch.Tensor:
    if t.shape[0] == rows and t.shape[1] == cols:
        return t.contiguous()
    out = torch.zeros((rows, cols), dtype=t.dtype, device=t.device)
    out[: t.shape[0], : t.shape[1]] = t
    return out


def _mm_nt(a16: torch.Tensor, b16: torch.Tensor, bias: torch.Tensor = None) -> torch.Tensor:
    """fp32 [M, N] = a16 [M, K] @ b16 [N, K]^T (+ bias [N], fused into the GEMM epilogue) on the MFMA GEMM kernel; K is zero-padded to a
    multiple of 64 and N to a multiple of 128 (the kernel's tile constraints), M is arbitrary.  A small output with a very long
    contraction (the weight gradients: M, N <= a few hundred, K = all rows of the batch) is cut into slices of K that run on separate
    workgroups and accumulate with fp32 atomics (``split_k``) - two workgroups would otherwise walk the whole contraction serially."""
    M, K = a16.shape
    N = b16.shape[0]
    Np = (N + 127) // 128 * 128
    tiles = ((M + 127) // 128) * (Np // 128)
    split = 0
    if bias is None and K >= 8192 and tiles < 128:
        split = 1
        while split * 2 * tiles <= 512 and K // (split * 2) >= 1024:
            split *= 2
    unit = 64 * max(split, 1)
    Kp = (K + unit - 1) // unit * unit
    a = _pad_to(a16, M, Kp)
    b = _pad_to(b16, Np, Kp)
    if bias is not None and Np != N:
        bias = torch.nn.functional.pad(bias.float(), (0, Np - N))
    out = ops.gemm(a, b, None if bias is None else bias.float().contiguous(), out_dtype=torch.float32, split_k=split if split > 1 else 0)
    return out if Np == N else out[:, :N]


class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        shape = x.shape
        x2 = x.reshape(-1, shape[-1])
        a16 = x2.to(torch.bfloat16)
        w16 = weight.to(torch.bfloat16)
        y = _mm_nt(a16, w16, None if bias is None else bias.detach())
        ctx.save_for_backward(a16, w16)
        ctx.has_bias = bias is not None
        ctx.in_shape = shape
        return y.reshape(*shape[:-1], weight.shape[0])

    @staticmethod
    def backward(ctx, dy):
        a16, w16 = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.shape[-1]).contiguous()
        dy16 = dy2.to(torch.bfloat16)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = _mm_nt(dy16, w16.t().contiguous()).reshape(ctx.in_shape)            # dY [M,N] @ W [N,K]
        if ctx.needs_input_grad[1]:
            dw = _mm_nt(dy16.t().contiguous(), a16.t().contiguous())                 # dY^T [N,M] @ X [M,K]
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy2.sum(dim=0)
        return dx, dw, db


def linear(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor = None) -> torch.Tensor:
    return _Linear.apply(x, weight, bias)


class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        shape = x.shape
        x2 = x.reshape(-1, shape[-1]).float().contiguous()
        y = ops.layernorm(x2, weight.detach().float().contiguous(), bias.detach().float().contiguous(), eps)
        ctx.save_for_backward(x2, weight)
        ctx.eps = eps
        return y.reshape(shape)

    @staticmethod
    def backward(ctx, dy):
        x2, weight = ctx.saved_tensors
        rows, dim = x2.shape
        dy2 = dy.reshape(rows, dim).float().contiguous()
        dx = torch.empty_like(x2)
        dw = torch.zeros((dim,), dtype=torch.float32, device=x2.device)
        db = torch.zeros((dim,), dtype=torch.float32, device=x2.device)
        w = weight.detach().float().contiguous()
        _lib.check(_lib.load().msam_layernorm_backward(x2.data_ptr(), w.data_ptr(), dy2.data_ptr(), float(ctx.eps), rows, dim,
                                                       dx.data_ptr(), dw.data_ptr(), db.data_ptr(), _lib.stream_ptr()),
                   "msam_layernorm_backward")
        return dx.reshape(dy.shape), dw, db, None


def layer_norm(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, eps: float) -> torch.Tensor:
    """LayerNorm over the last dim (64, 128 or 256 channels; 768 / 1024 / 1280 for the image encoder)."""
    return _LayerNorm.apply(x, weight, bias, eps)


class _Attention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v):
        # q [B, H, Nq, D], k / v [B, H, Nk, D]
        B, H, Nq, D = q.shape
        Nk = k.shape[2]
        qc, kc, vc = q.float().contiguous(), k.float().contiguous(), v.float().contiguous()
        out = torch.empty_like(qc)
        lse = torch.empty((B, H, Nq), dtype=torch.float32, device=q.device)
        scale = 1.0 / math.sqrt(D)
        _lib.check(_lib.load().msam_attention_forward(qc.data_ptr(), kc.data_ptr(), vc.data_ptr(), B * H, Nq, Nk, D, scale,
                                                      out.data_ptr(), lse.data_ptr(), _lib.stream_ptr()), "msam_attention_forward")
        ctx.save_for_backward(qc, kc, vc, out, lse)
        ctx.scale = scale
        return out

    @staticmethod
    def backward(ctx, dout):
        qc, kc, vc, out, lse = ctx.saved_tensors
        B, H, Nq, D = qc.shape
        Nk = kc.shape[2]
        do = dout.float().contiguous()
        dq, dk, dv = torch.empty_like(qc), torch.empty_like(kc), torch.empty_like(vc)
        delta = torch.empty_like(lse)
        _lib.check(_lib.load().msam_attention_backward(qc.data_ptr(), kc.data_ptr(), vc.data_ptr(), out.data_ptr(), do.data_ptr(),
                                                       lse.data_ptr(), B * H, Nq, Nk, D, ctx.scale, dq.data_ptr(), dk.data_ptr(),
                                                       dv.data_ptr(), delta.data_ptr(), _lib.stream_ptr()), "msam_attention_backward")
        return dq, dk, dv


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """softmax(q k^T / sqrt(D)) v for q [B, H, Nq, D], k / v [B, H, Nk, D], D in (16, 32)."""
    return _Attention.apply(q, k, v)


class _RelPosAttention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, bias_h, bias_w, scale):
        # q / k / v [BH, N, D] with N = Gh * Gw tokens of one grid; bias_h [BH, N, Gh], bias_w [BH, N, Gw]
        BH, N, D = q.shape
        Gh, Gw = bias_h.shape[2], bias_w.shape[2]
        if Gh * Gw != N:
            raise ValueError(f"relpos_attention: {N} tokens are not a {Gh} x {Gw} grid")
        qc, kc, vc = q.float().contiguous(), k.float().contiguous(), v.float().contiguous()
        bh, bw = bias_h.float().contiguous(), bias_w.float().contiguous()
        out = torch.empty_like(qc)
        lse = torch.empty((BH, N), dtype=torch.float32, device=q.device)
        _lib.check(_lib.load().msam_relpos_attention_forward(qc.data_ptr(), kc.data_ptr(), vc.data_ptr(), bh.data_ptr(), bw.data_ptr(),
                                                             BH, Gh, Gw, D, float(scale), out.data_ptr(), lse.data_ptr(),
                                                             _lib.stream_ptr()), "msam_relpos_attention_forward")
        ctx.save_for_backward(qc, kc, vc, bh, bw, out, lse)
        ctx.scale = float(scale)
        return out

    @staticmethod
    def backward(ctx, dout):
        qc, kc, vc, bh, bw, out, lse = ctx.saved_tensors
        BH, N, D = qc.shape
        Gh, Gw = bh.shape[2], bw.shape[2]
        do = dout.float().contiguous()
        dq, dk, dv = torch.empty_like(qc), torch.empty_like(kc), torch.empty_like(vc)
        dbh, dbw = torch.empty_like(bh), torch.empty_like(bw)
        delta = torch.empty_like(lse)
        _lib.check(_lib.load().msam_relpos_attention_backward(
            qc.data_ptr(), kc.data_ptr(), vc.data_ptr(), bh.data_ptr(), bw.data_ptr(), out.data_ptr(), do.data_ptr(), lse.data_ptr(),
            BH, Gh, Gw, D, ctx.scale, dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), dbh.data_ptr(), dbw.data_ptr(), delta.data_ptr(),
            _lib.stream_ptr()), "msam_relpos_attention_backward")
        return dq, dk, dv, dbh, dbw, None


def relpos_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, bias_h: torch.Tensor, bias_w: torch.Tensor,
                     scale: float) -> torch.Tensor:
    """softmax_j(scale q_i k_j + bias_h[i, j // Gw] + bias_w[i, j % Gw]) v for q / k / v [BH, Gh * Gw, D] (D = 64 or 80)."""
    return _RelPosAttention.apply(q, k, v, bias_h, bias_w, scale)


# Which implementation the image encoder's attention takes under autograd (training/encoders.py):
#   "gemm"   (default) the two products as plain library batched GEMMs on bf16 operands (torch.bmm = rocBLAS / hipBLASLt on MFMA; what
#            the reference's AMP does), scores + decomposed bias + softmax in fp32 torch operators, backward by autograd.  The score
#            matrix is materialised (24 heads x 4096^2 fp32 = 1.6 GB per global block of vit_b at batch 2: 288 GB of HBM hold it) - the
#            one-thread-per-row fp32 kernels were 34 % of a whole-model step (profiles/r03_train_profile.md);
#   "kernel" the hand-written fp32 kernels (msam_relpos_attention_forward / _backward): no score matrix in HBM, exact fp32, slow -
#            the checked reference of the composite (tests/test_gpu_training_encoders.py) and the low-memory option.
RELPOS_ATTENTION_IMPL = os.environ.get("MSAM_RELPOS_IMPL", "gemm")


def relpos_attention_gemm(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, bias_h: torch.Tensor, bias_w: torch.Tensor,
                          scale: float) -> torch.Tensor:
    BH, N, _ = q.shape
    gh, gw = bias_h.shape[-1], bias_w.shape[-1]
    s = torch.bmm((q * scale).to(torch.bfloat16), k.to(torch.bfloat16).transpose(1, 2)).float()
    s = (s.view(BH, N, gh, gw) + bias_h.unsqueeze(-1) + bias_w.unsqueeze(-2)).view(BH, N, N)
    p = torch.softmax(s, dim=-1)
    return torch.bmm(p.to(torch.bfloat16), v.to(torch.bfloat16)).float()


def relpos_attention_auto(q, k, v, bias_h, bias_w, scale: float) -> torch.Tensor:
    if RELPOS_ATTENTION_IMPL == "gemm":
        return relpos_attention_gemm(q, k, v, bias_h, bias_w, scale)
    return relpos_attention(q, k, v, bias_h, bias_w, scale)
