"""The convolutional decoder of automatic instance segmentation on the library's fp32 kernels (``csrc/strict.hip``): what
``DecoderAdapter.forward`` (reference ``micro_sam/instance_segmentation.py:710-735``) computes from precomputed image embeddings - three
distance / foreground maps at the original image size - without a torch operator on the data path.

Everything is channels-last (one row per pixel), so that each layer is ONE product on the f32-input MFMA kernel:

  Conv2d 3 x 3, padding 1        implicit GEMM - the tile loader gathers the nine taps (``msam_sgemm_t.conv_*``; no im2col matrix: it would
                                 be 4.8 GB for the 1024^2 x 128-channel head); bias, BatchNorm2d on its running statistics (column scale /
                                 shift) and ReLU in the epilogue
  ConvTranspose2d 2 x 2, stride 2   a product with N = 4 C_out columns whose epilogue scatters the four sub-pixels (``shuffle_*``)
  Upsampler2d                    ``msam_strict_resize_bilinear`` (x 2) + a 1 x 1 product
  InstanceNorm2d                 ``msam_strict_instance_norm`` (chunked two-pass statistics merged in double)
  torch.cat([x, skip], dim=1)    no copy: the two producers write the two column ranges of one buffer (``ldc`` = total width)
  out_conv + Sigmoid             product with the activation in its epilogue; ``postprocess_masks`` = one bilinear pass over the un-padded
                                 window, written as NCHW

Parameters stay in the torch module tree (``models/unetr.py``: checkpoints load unchanged, the tree is what trains); this class holds fp32
GEMM-layout copies, rebuilt when a parameter's version moves.  Checked against ``oracle/unetr_ref.py`` (tests/test_gpu_ais.py; on the CPU
through the host build, tests/test_unetr_hip_host.py)."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Tuple

import torch

from .. import _lib
from ..strict import ACT_NONE, ACT_RELU, gemm

ACT_SIGMOID = _lib.ACT_SIGMOID
IMG_SIZE = 1024


def _f32(t: torch.Tensor, dev) -> torch.Tensor:
    return t.detach().to(device=dev, dtype=torch.float32).contiguous()


def conv_gemm(x: torch.Tensor, pitch: int, B: int, H: int, W: int, Cin: int, w: torch.Tensor, bias, scale=None, shift=None, act: int = ACT_NONE,
              out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """3 x 3 / padding 1 convolution of channels-last x [B*H*W, >= Cin] (pixel pitch ``pitch``) with w [Cout, 9 Cin] (columns ky, kx, c)."""
    M, N = B * H * W, w.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=w.device)
    p = _lib.SGemmParams()
    p.A, p.lda, p.W, p.ldw, p.M, p.N, p.K = x.data_ptr(), pitch, w.data_ptr(), w.stride(0), M, N, 9 * Cin
    p.bias = None if bias is None else bias.data_ptr()
    if scale is not None:
        p.col_scale, p.col_shift = scale.data_ptr(), shift.data_ptr()
    p.act = act
    p.conv_h, p.conv_w, p.conv_c = H, W, Cin
    p.out, p.ldc = out.data_ptr(), out.stride(0)
    _lib.check(_lib.load().msam_strict_gemm(C.byref(p), _lib.stream_ptr()), "msam_strict_gemm(conv3x3)")
    return out


def deconv_gemm(x: torch.Tensor, B: int, H: int, W: int, w: torch.Tensor, bias4: torch.Tensor, Cout: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """ConvTranspose2d(kernel 2, stride 2) of channels-last x [B*H*W, Cin] with w [4 Cout (ky, kx, co), Cin] -> [B*2H*2W, Cout]."""
    M = B * H * W
    if out is None:
        out = torch.empty((4 * M, Cout), dtype=torch.float32, device=w.device)
    p = _lib.SGemmParams()
    p.A, p.lda, p.W, p.ldw, p.M, p.N, p.K = x.data_ptr(), x.stride(0), w.data_ptr(), w.stride(0), M, 4 * Cout, w.shape[1]
    p.bias = bias4.data_ptr()
    p.shuffle_h, p.shuffle_w, p.shuffle_c = H, W, Cout
    p.out, p.ldc = out.data_ptr(), out.stride(0)
    _lib.check(_lib.load().msam_strict_gemm(C.byref(p), _lib.stream_ptr()), "msam_strict_gemm(deconv2x2)")
    return out


def instance_norm(x: torch.Tensor, B: int, HW: int, Cn: int, eps: float = 1e-5) -> torch.Tensor:
    out = torch.empty((B * HW, Cn), dtype=torch.float32, device=x.device)
    nchunk = (HW + 2047) // 2048
    ws = torch.empty(2 * B * nchunk * Cn + 2 * B * Cn, dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().msam_strict_instance_norm(x.data_ptr(), x.stride(0), B, HW, Cn, float(eps), out.data_ptr(), ws.data_ptr(), ws.numel(),
                                                     _lib.stream_ptr()), "msam_strict_instance_norm")
    return out


def resize(x: torch.Tensor, B: int, h: int, w: int, pitch_h: int, pitch_w: int, Cn: int, H2: int, W2: int, scale_h: float, scale_w: float,
           nchw: bool = False) -> torch.Tensor:
    out = torch.empty((B, Cn, H2, W2) if nchw else (B * H2 * W2, Cn), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().msam_strict_resize_bilinear(x.data_ptr(), B, h, w, pitch_h, pitch_w, x.stride(0), Cn, H2, W2, float(scale_h), float(scale_w),
                                                       1 if nchw else 0, out.data_ptr(), _lib.stream_ptr()), "msam_strict_resize_bilinear")
    return out


class HipUnetrDecoder:
    """The decoder part of a ``models.unetr.UNETR`` / ``DecoderAdapter`` (same attribute names) on the HIP kernels."""

    def __init__(self, module) -> None:
        self.m = module
        self._w = None
        self._key = None

    def _param_key(self):
        return tuple((id(p), p._version, p.data_ptr()) for p in list(self.m.parameters()) + list(self.m.buffers()))

    def _weights(self, dev):
        key = (self._param_key(), str(dev))
        if self._w is not None and key == self._key:
            return self._w
        m = self.m

        def conv3(conv):                        # Conv2d [Cout, Cin, 3, 3] -> [Cout, (ky, kx, c)]
            return (_f32(conv.weight.permute(0, 2, 3, 1).reshape(conv.weight.shape[0], -1), dev), _f32(conv.bias, dev), conv.weight.shape[1])

        def up(mod):
            if hasattr(mod, "block"):           # SingleDeconv2DBlock: ConvTranspose2d [Cin, Cout, 2, 2] -> rows (ky, kx, co)
                wt = mod.block.weight
                return ("deconv", _f32(wt.permute(2, 3, 1, 0).reshape(4 * wt.shape[1], wt.shape[0]), dev), _f32(mod.block.bias.repeat(4), dev),
                        wt.shape[1])
            return ("bilinear", _f32(mod.conv.weight.reshape(mod.conv.weight.shape[0], -1), dev), _f32(mod.conv.bias, dev), mod.conv.weight.shape[0])

        def conv_block(blk):
            return [conv3(blk.block[1]), conv3(blk.block[4])]

        def deconv_block(blk):
            bn = blk.block[2]
            alpha = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
            beta = bn.bias.detach().float() - bn.running_mean.detach().float() * alpha
            return dict(up=up(blk.block[0]), conv=conv3(blk.block[1].block), scale=_f32(alpha, dev), shift=_f32(beta, dev))
        w = dict(deconv=[deconv_block(getattr(m, f"deconv{i}")) for i in (1, 2, 3, 4)], base=conv_block(m.base),
                 blocks=[conv_block(b) for b in m.decoder.blocks], samplers=[up(s_) for s_ in m.decoder.samplers],
                 deconv_out=up(m.deconv_out), head=conv_block(m.decoder_head),
                 out=(_f32(m.out_conv.weight.reshape(m.out_conv.weight.shape[0], -1), dev), _f32(m.out_conv.bias, dev)))
        act = m.final_activation
        if act is None:
            w["final_act"] = ACT_NONE
        elif isinstance(act, torch.nn.Sigmoid):
            w["final_act"] = ACT_SIGMOID
        else:
            raise NotImplementedError(f"micro_sam_amd: final activation {type(act).__name__} of the UNETR decoder is not on the HIP path (Sigmoid / None)")
        self._w, self._key = w, key
        return w

    # -- layers
    def _up(self, spec, x, B, H, W, out=None):
        kind, wt, bias, cout = spec
        if kind == "deconv":
            return deconv_gemm(x, B, H, W, wt, bias, cout, out=out)
        y = resize(x, B, H, W, H, W, x.shape[1], 2 * H, 2 * W, 0.5, 0.5)
        return gemm(y, wt, bias, out=out)

    def _conv_block(self, specs, x, B, H, W):
        for wt, bias, cin in specs:
            xn = instance_norm(x, B, H * W, cin)
            x = conv_gemm(xn, cin, B, H, W, cin, wt, bias, act=ACT_RELU)
            del xn
        return x

    def _deconv_block(self, spec, x, B, H, W, out):
        y = self._up(spec["up"], x, B, H, W)
        wt, bias, cin = spec["conv"]
        return conv_gemm(y, y.stride(0), B, 2 * H, 2 * W, cin, wt, bias, spec["scale"], spec["shift"], ACT_RELU, out=out)

    @torch.no_grad()
    def decode_rows(self, z12: torch.Tensor) -> Tuple[torch.Tensor, int]:
        """Image embeddings [B, 256, 64, 64] -> channels-last decoder output rows [B * 1024 * 1024, 4] (the first ``out_channels`` columns
        are the maps; 4 = the row pitch) - ``DecoderAdapter._forward_impl`` without the final layout change.  (Any square grid G works:
        the output is 16 G on a side.)"""
        dev = z12.device
        _lib.require_gpu(dev)
        w = self._weights(dev)
        B, Cz, G = z12.shape[0], z12.shape[1], z12.shape[2]
        if z12.dim() != 4 or z12.shape[3] != G or Cz % 4:
            raise ValueError(f"expected square image embeddings [B, C, G, G] with C % 4 == 0, got {tuple(z12.shape)}")
        rows = z12.to(torch.float32).permute(0, 2, 3, 1).reshape(B * G * G, Cz).contiguous()      # NCHW -> one row per pixel (a copy, no arithmetic)
        # widths: sampler / deconv_out outputs (first columns of a concatenation) and the skip inputs (last columns)
        cs = [s_[3] for s_ in w["samplers"]] + [w["deconv_out"][3]]
        ck = [d["conv"][0].shape[0] for d in w["deconv"]]
        sizes = [2 * G, 4 * G, 8 * G, 16 * G]
        cat = [torch.empty((B * sizes[i] * sizes[i], cs[i] + ck[i]), dtype=torch.float32, device=dev) for i in range(4)]
        x, H = rows, G
        for i in range(4):                                                               # z9, z6, z3, z0: the skip halves of the four buffers
            x = self._deconv_block(w["deconv"][i], x, B, H, H, out=cat[i][:, cs[i]:])
            H *= 2
        x = self._conv_block(w["base"], rows, B, G, G)
        H = G
        for i in range(3):
            self._up(w["samplers"][i], x, B, H, H, out=cat[i][:, :cs[i]])
            H *= 2
            x = self._conv_block(w["blocks"][i], cat[i], B, H, H)
            cat[i] = None
        self._up(w["deconv_out"], x, B, H, H, out=cat[3][:, :cs[3]])
        H *= 2
        x = self._conv_block(w["head"], cat[3], B, H, H)
        cat[3] = None
        wo, bo = w["out"]
        nout = wo.shape[0]
        if nout > 4:
            raise NotImplementedError("micro_sam_amd: more than 4 output channels of the UNETR decoder")
        out = torch.zeros((B * H * H, 4), dtype=torch.float32, device=dev)
        gemm(x, wo, bo, act=w["final_act"], out=out[:, :nout])
        return out, nout

    @torch.no_grad()
    def forward(self, input_: torch.Tensor, input_shape, original_shape) -> torch.Tensor:
        """``DecoderAdapter.forward``: decoder + ``postprocess_masks`` -> [B, out_channels, *original_shape] fp32."""
        rows, nout = self.decode_rows(input_)
        B = input_.shape[0]
        if input_.shape[2] * 16 != IMG_SIZE:
            raise ValueError(f"postprocess_masks expects the decoder output at {IMG_SIZE}^2 (embeddings of a 64 x 64 grid), got {tuple(input_.shape)}")
        ih, iw = int(input_shape[0]), int(input_shape[1])
        oh, ow = int(original_shape[0]), int(original_shape[1])
        # interpolate(1024 -> 1024) is the identity; crop the padding (a window of the 1024^2 rows), resize to the original size, NCHW
        res = resize(rows, B, ih, iw, IMG_SIZE, IMG_SIZE, 4, oh, ow, ih / oh, iw / ow, nchw=True)
        return res[:, :nout]
