"""The AIS decoder's HIP path on the CPU (csrc/strict.hip behind the host shim): every layer primitive of micro_sam_amd/models/unetr_hip.py
against the torch operator it replaces, and the whole decoder graph (both up-sampler flavours, a small grid) against oracle/unetr_ref.py -
the independent restatement of the reference's DecoderAdapter graph (micro_sam/instance_segmentation.py:710-733).  TEST INFRASTRUCTURE; the
device run is tests/test_gpu_ais.py."""
import os

import pytest
import torch
import torch.nn.functional as F

from hip_host_shim import build_library

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def host(tmp_path_factory):
    from micro_sam_amd import _lib
    lib = build_library(str(tmp_path_factory.mktemp("host_unetr")), ROOT)
    for name, (res, args) in _lib._PROTOS.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    saved = (_lib._lib, _lib.require_gpu, _lib.stream_ptr, _lib.ptr)
    _lib._lib = lib
    _lib.require_gpu = lambda device=None: torch.device("cpu") if device is None else torch.device(device)
    _lib.stream_ptr = lambda: None
    _lib.ptr = lambda t: None if t is None else t.data_ptr()
    try:
        yield lib
    finally:
        _lib._lib, _lib.require_gpu, _lib.stream_ptr, _lib.ptr = saved


def _rows(x):                     # NCHW -> channels-last rows
    return x.permute(0, 2, 3, 1).reshape(-1, x.shape[1]).contiguous()


def _nchw(rows, B, H, W):
    return rows.reshape(B, H, W, -1).permute(0, 3, 1, 2)


@pytest.mark.parametrize("small_below", [512, 0])
def test_layer_primitives(host, small_below):
    """(small_below: the implicit-GEMM convolution on the 64 x 64 tile these sizes get by default, and on the 128 x 128 tile of the
    1024^2 feature maps)"""
    from micro_sam_amd.models import unetr_hip as UH
    assert host.msam_tune_set(b"sgemm_small_below", small_below) == 0
    try:
        _layer_primitives(UH)
    finally:
        assert host.msam_tune_set(b"sgemm_small_below", 512) == 0


def _layer_primitives(UH):
    g = torch.Generator().manual_seed(0)
    B, H, W, Cin, Cout = 2, 9, 13, 8, 12
    x = torch.randn(B, Cin, H, W, generator=g)
    wt, bias = torch.randn(Cout, Cin, 3, 3, generator=g) / 6, torch.randn(Cout, generator=g)
    # 3 x 3 convolution + BatchNorm (running statistics) + ReLU in the epilogue, input = a column slice of a wider buffer
    wide = torch.randn(B * H * W, Cin + 4, generator=g)
    wide[:, 4:] = _rows(x)
    scale, shift = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g)
    out = UH.conv_gemm(wide[:, 4:], wide.stride(0), B, H, W, Cin, wt.permute(0, 2, 3, 1).reshape(Cout, -1).contiguous(), bias, scale, shift, UH.ACT_RELU)
    ref = F.relu(F.conv2d(x.double(), wt.double(), bias.double(), padding=1) * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1))
    assert (_nchw(out, B, H, W) - ref).abs().max().item() <= 1e-5
    # transposed 2 x 2 / stride 2 convolution written into the first columns of a concatenation buffer
    wd, bd = torch.randn(Cin, Cout, 2, 2, generator=g) / 3, torch.randn(Cout, generator=g)
    cat = torch.full((B * 4 * H * W, Cout + 4), float("nan"))
    UH.deconv_gemm(_rows(x), B, H, W, wd.permute(2, 3, 1, 0).reshape(4 * Cout, Cin).contiguous(), bd.repeat(4), Cout, out=cat[:, :Cout])
    ref = F.conv_transpose2d(x.double(), wd.double(), bd.double(), stride=2)
    assert (_nchw(cat[:, :Cout], B, 2 * H, 2 * W) - ref).abs().max().item() <= 1e-5 and torch.isnan(cat[:, Cout:]).all()
    # InstanceNorm2d over more than one statistics chunk, large mean
    xi = torch.randn(2, 8, 70, 70, generator=g) * 3 + 40
    out = UH.instance_norm(_rows(xi), 2, 4900, 8)
    assert (_nchw(out, 2, 70, 70) - F.instance_norm(xi.double(), eps=1e-5)).abs().max().item() <= 2e-5
    # bilinear resize: x2 (Upsampler2d), a cropped window to another size (postprocess_masks), NCHW output
    xr = torch.randn(2, 4, 12, 10, generator=g)
    up = UH.resize(_rows(xr), 2, 12, 10, 12, 10, 4, 24, 20, 0.5, 0.5)
    assert (_nchw(up, 2, 24, 20) - F.interpolate(xr, scale_factor=2, mode="bilinear", align_corners=False)).abs().max().item() <= 1e-6
    win = UH.resize(_rows(xr), 2, 9, 7, 12, 10, 4, 31, 17, 9 / 31, 7 / 17, nchw=True)
    # (non-integer scales: torch's own CPU result differs between builds in the last bit of the source index - fused vs separately rounded
    #  scale * (dst + 0.5) - 0.5, DESIGN.md section 3 - so this is a tolerance, not an equality)
    assert (win - F.interpolate(xr[:, :, :9, :7], (31, 17), mode="bilinear", align_corners=False)).abs().max().item() <= 5e-6
    same = UH.resize(_rows(xr), 2, 12, 10, 12, 10, 4, 12, 10, 1.0, 1.0, nchw=True)
    assert torch.equal(same, xr)


@pytest.mark.parametrize("transpose", [True, False])
def test_decoder_graph_is_the_oracles(host, transpose):
    """HipUnetrDecoder.decode_rows on a 6 x 6 grid (-> 96 x 96) with small widths, both up-sampler flavours, random BatchNorm statistics,
    against oracle/unetr_ref.decode (and against the torch module tree the parameters live in)."""
    from micro_sam_amd.models import unetr as U
    from micro_sam_amd.models.unetr_hip import HipUnetrDecoder
    from oracle import unetr_ref as R
    torch.manual_seed(3 + int(transpose))
    widths = {"use_conv_transpose": transpose, "base": (16, 32), "blocks": [(32, 16), (16, 8), (8, 8)], "samplers": [(32, 16), (16, 8), (8, 4)],
              "deconv": [(16, 16), (16, 8), (8, 4), (4, 4)], "deconv_out": (8, 4), "head": (8, 8), "out": (8, 3)}
    net = U.UNETR(torch.nn.Identity(), widths)
    net.eval()
    with torch.no_grad():
        for mod in net.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.running_mean.normal_(0, 0.5); mod.running_var.uniform_(0.5, 2.0); mod.weight.uniform_(0.5, 1.5); mod.bias.normal_(0, 0.3)
    z = torch.randn(2, 16, 6, 6)
    sd = {k: v.detach() for k, v in net.state_dict().items() if not k.startswith("encoder")}
    ref = R.decode(sd, z)
    with torch.no_grad():
        assert (net.decode(z) - ref).abs().max().item() <= 1e-5                       # module tree == oracle (two independent restatements)
    rows, nout = HipUnetrDecoder(net).decode_rows(z)
    got = rows.reshape(2, 96, 96, 4).permute(0, 3, 1, 2)[:, :nout]
    assert nout == 3 and (got - ref).abs().max().item() <= 2e-5, (got - ref).abs().max().item()
