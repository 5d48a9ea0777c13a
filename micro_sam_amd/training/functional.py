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
import weakref

import torch

from .. import _lib, ops


def _pad_to(t: torch.Tensor, rows: int, cols: int) -> torch.Tensor:
    if t.shape[0] == rows and t.shape[1] == cols:
        return t.contiguous()
    out = torch.zeros((rows, cols), dtype=t.dtype, device=t.device)
    out[: t.shape[0], : t.shape[1]] = t
    return out


def _mm_nt(a16: torch.Tensor, b16: torch.Tensor, bias: torch.Tensor = None, n: int = None) -> torch.Tensor:
    """fp32 [M, N] = a16 [M, K] @ b16 [N, K]^T (+ bias [N], fused into the GEMM epilogue) on the MFMA GEMM kernel; K is zero-padded to a
    multiple of 64 and N to a multiple of 128 (the kernel's tile constraints; ``n`` = valid rows of an operand padded beforehand), M is
    arbitrary.  A small output with a very long
    contraction (the weight gradients: M, N <= a few hundred, K = all rows of the batch) is cut into slices of K that run on separate
    workgroups and accumulate with fp32 atomics (``split_k``) - two workgroups would otherwise walk the whole contraction serially."""
    M, K = a16.shape
    N = b16.shape[0] if n is None else n             # n: rows of b16 that count (b16 already zero-padded to the tile, _weight16)
    K = max(K, b16.shape[1])
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


# bf16 copies of the weights, W [N, K] and W^T [K, N], per parameter and parameter version: the mask decoder's weights are used by
# 2 images x 8 sub-iterations per optimiser step (reference sam_trainer.py:252-292), the casts and the transposition are the same 16 times
_W16 = {}


def _weight16(weight: torch.Tensor):
    base = weight._base if weight._base is not None else weight
    if not isinstance(base, torch.nn.Parameter):
        w16 = weight.detach().to(torch.bfloat16).contiguous()
        return w16, w16.t().contiguous()
    key = (id(base), tuple(weight.shape), tuple(weight.stride()), weight.storage_offset())
    hit = _W16.get(key)
    stamp = (base._version, base.data_ptr(), base.device.index)       # param.data = ... / .to(device) keep the counter: address + device too
    if hit is not None and hit[0]() is base and hit[1] == stamp:
        return hit[2], hit[3]
    if len(_W16) > 4096:                     # parameters of discarded models
        for k in [k for k, v in _W16.items() if v[0]() is None]:
            del _W16[k]
    w16 = weight.detach().to(torch.bfloat16)
    N, K = w16.shape
    w16t = _pad_to(w16.t(), (K + 127) // 128 * 128, (N + 63) // 64 * 64)          # padded to the product kernel's tiles once, not per call
    w16 = _pad_to(w16, (N + 127) // 128 * 128, (K + 63) // 64 * 64)
    _W16[key] = (weakref.ref(base), stamp, w16, w16t)
    return w16, w16t


def _fusable(t: torch.Tensor) -> bool:
    """msam_cast_transpose takes it: rows contiguous, widths in fours, fp32 or bf16 (everything the model's linears see)."""
    return (t.dim() == 2 and t.stride(1) == 1 and t.shape[1] % 4 == 0 and t.stride(0) % 4 == 0 and t.shape[0] <= 4_000_000
            and t.dtype in (torch.float32, torch.bfloat16) and t.is_cuda and t.data_ptr() % 16 == 0)


class _Linear(torch.autograd.Function):
    """y = x W^T + b.  The weight gradient dW = dY^T X contracts over the rows, and the product kernel wants both operands contiguous
    along the contraction: X^T is written next to the bf16 copy of X in the forward pass, dY^T / the bf16 copy of dY / the bias
    gradient in ONE pass over dY in the backward pass (ops.cast_transpose) - torch's cast + strided transpose copy + sum were 30 % of a
    fine-tuning step (profiles/r03_experiments.md section 8)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        shape = x.shape
        x2 = x.reshape(-1, shape[-1])
        need_dw = weight.requires_grad
        if _fusable(x2):
            a16, a16t, _ = ops.cast_transpose(x2, True, need_dw, False)
        else:
            a16 = x2.to(torch.bfloat16)
            a16t = a16.t().contiguous() if need_dw else None
        w16, w16t = _weight16(weight)
        y = _mm_nt(a16, w16, None if bias is None else bias.detach(), n=weight.shape[0])
        ctx.save_for_backward(a16t if need_dw else None, w16t)
        ctx.has_bias = bias is not None
        ctx.in_shape = shape
        return y.reshape(*shape[:-1], weight.shape[0])

    @staticmethod
    def backward(ctx, dy):
        a16t, w16t = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.shape[-1])
        want_dw = ctx.needs_input_grad[1] and a16t is not None
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        if _fusable(dy2):
            dy16, dy16t, db = ops.cast_transpose(dy2, ctx.needs_input_grad[0], want_dw, want_db)
        else:
            dy2 = dy2.contiguous()
            dy16 = dy2.to(torch.bfloat16)
            dy16t = dy16.t().contiguous() if want_dw else None
            db = dy2.float().sum(dim=0) if want_db else None
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = _mm_nt(dy16, w16t, n=ctx.in_shape[-1]).reshape(ctx.in_shape)        # dY [M,N] @ W [N,K]
        if want_dw:
            dw = _mm_nt(dy16t, a16t)                                                 # dY^T [N,M] @ X [M,K]
        return dx, dw, db if want_db else None


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
