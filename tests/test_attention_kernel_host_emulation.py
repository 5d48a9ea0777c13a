"""window_attention_kernel (csrc/attention.hip) executed on the CPU behind tests/hip_host_shim.py (one host thread per lane, the MFMA in
its register layout): staging of K / V^T with the bias-only padding tokens of the 70 x 70 padded grid, rel-pos terms by MFMA, masked
softmax, P V - against a plain fp64 restatement of SAM's windowed attention with decomposed relative position bias on the same 16-bit
q / k / v.  The device run of the same kernel: tests/test_gpu_kernels.py::test_vit_attention_vs_oracle."""
import ctypes
import math
import os

import numpy as np
import pytest
import torch

from hip_host_shim import build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "micro_sam_amd", "csrc", "attention.hip")

ENTRY = r"""
extern "C" void emu_global_attention(int f16, const u16* q, const u16* k, const u16* v, const u16* relh, const u16* relw, int grid, int heads,
                                     float scale, u16* out) {          // grid <= B * heads * 32 workgroups of 128 queries (two image rows) each
    if (f16) launch_grid(grid, 1, [=] { global_attention_kernel<64, true>(q, k, v, relh, relw, heads, scale, out); });
    else launch_grid(grid, 1, [=] { global_attention_kernel<64, false>(q, k, v, relh, relw, heads, scale, out); });
}
extern "C" void emu_window_attention(int f16, const u16* q, const u16* k, const u16* v, const u16* relh, const u16* relw, const float* bias,
                                     int B, int heads, float scale, u16* out) {
    if (f16) launch_grid(B * 25 * heads, 1, [=] { window_attention_kernel<64, true>(q, k, v, relh, relw, bias, heads, scale, out); });
    else launch_grid(B * 25 * heads, 1, [=] { window_attention_kernel<64, false>(q, k, v, relh, relw, bias, heads, scale, out); });
}
"""


@pytest.fixture(scope="module")
def emu(tmp_path_factory):
    text = open(SRC).read()
    start = text.index("constexpr int TOK = 4096;")
    end = text.index("}  // namespace", start)
    body = text[start:end]
    assert "window_attention_kernel" in body and "global_attention_kernel" in body
    # the transposing LDS read is a clang builtin on an address-space pointer: the shim's restatement takes its place
    a = body.index("typedef short gs16x4_t")
    b = body.index("template <int HD, bool F16 = false>", a)
    body = body[:a] + "static inline uint2 g_tr16(const unsigned char* p) { return ds_read_tr16_b64_emu(p); }\n" + body[b:]
    lib = build(str(tmp_path_factory.mktemp("emu_attn")), "attn", body, ENTRY)
    lib.emu_window_attention.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 6 + [ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_void_p]
    lib.emu_global_attention.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 5 + [ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_void_p]
    return lib


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _bits(t, f16):
    return (t.to(torch.float16) if f16 else t.to(torch.bfloat16)).view(torch.int16).numpy().view(np.uint16).copy()


def _round(t, f16):
    return (t.to(torch.float16) if f16 else t.to(torch.bfloat16)).double()


def _reference(q, k, v, rel_h, rel_w, bias, heads, scale, f16):
    """[heads, 4096, 64] 16-bit-rounded q / k / v of ONE image -> [4096, heads * 64]: 5 x 5 windows of 14 x 14 over the grid padded to
    70 x 70, padding tokens carry the projection bias (the layer input is zero there), softmax over all 196 tokens of a window."""
    hd = 64
    D = heads * hd
    out = torch.zeros(64, 64, heads, hd, dtype=torch.float64)
    idx = torch.arange(14)
    dh = (idx[:, None] - idx[None, :]) + 13                                   # [q, k] -> row of the table
    for h in range(heads):
        grids = []
        for t, off in ((q, 0), (k, D), (v, 2 * D)):
            gpad = _round(bias[off + h * hd: off + (h + 1) * hd], f16).expand(70, 70, hd).clone()
            gpad[:64, :64] = t[h].double().reshape(64, 64, hd)
            grids.append(gpad)
        rh, rw = _round(rel_h, f16), _round(rel_w, f16)
        for wy in range(5):
            for wx in range(5):
                qs, ks, vs = (gr[wy * 14:(wy + 1) * 14, wx * 14:(wx + 1) * 14].reshape(196, hd) for gr in grids)
                s = scale * qs @ ks.t()
                th = torch.einsum("qc,qkc->qk", qs.reshape(14, 14, hd).reshape(196, hd), rh[dh][idx.repeat_interleave(14)])      # [196, 14 kh]
                tw = torch.einsum("qc,qkc->qk", qs, rw[dh][idx.repeat(14)])                                                        # [196, 14 kw]
                s = (s.reshape(196, 14, 14) + th[:, :, None] + tw[:, None, :]).reshape(196, 196)
                o = torch.softmax(s, dim=-1) @ vs
                o = o.reshape(14, 14, hd)
                ys, xs = min(14, 64 - wy * 14), min(14, 64 - wx * 14)
                out[wy * 14: wy * 14 + ys, wx * 14: wx * 14 + xs, h] = o[:ys, :xs]
    return out.reshape(4096, D)


@pytest.mark.parametrize("f16", [False, True])
def test_window_attention_kernel_source_on_the_cpu(emu, f16):
    g = torch.Generator().manual_seed(3 + int(f16))
    heads, hd = 2, 64
    D = heads * hd
    q, k, v = (torch.randn(heads, 4096, hd, generator=g) * 0.8 for _ in range(3))
    q, k, v = ((t.to(torch.float16) if f16 else t.to(torch.bfloat16)) for t in (q, k, v))
    rel_h, rel_w = torch.randn(27, hd, generator=g) * 0.1, torch.randn(27, hd, generator=g) * 0.1
    bias = torch.randn(3 * D, generator=g) * 0.3
    scale = 1.0 / math.sqrt(hd)
    out = np.zeros((4096, D), np.uint16)
    qa, ka, va = (t.view(torch.int16).numpy().view(np.uint16).copy() for t in (q, k, v))
    rha, rwa = _bits(rel_h, f16), _bits(rel_w, f16)
    ba = bias.numpy().astype(np.float32).copy()
    emu.emu_window_attention(int(f16), _ptr(qa), _ptr(ka), _ptr(va), _ptr(rha), _ptr(rwa), _ptr(ba), 1, heads, ctypes.c_float(scale), _ptr(out))
    got = torch.from_numpy(out.view(np.int16)).view(torch.float16 if f16 else torch.bfloat16).double()
    ref = _reference(q, k, v, rel_h, rel_w, bias, heads, scale, f16)
    err = (got - ref).abs().max().item()
    assert err <= (4e-3 if f16 else 2.5e-2) * ref.abs().max().item(), err        # 16-bit probabilities and outputs


@pytest.mark.parametrize("f16", [False])              # (the fp16 instantiation differs in the MFMA / pack helpers only: covered by the window test)
def test_global_attention_kernel_source_on_the_cpu(emu, f16):
    """Global attention (4096 keys): rel_h / rel_w tables by MFMA in the prologue, 128 key tiles with the base-2 online softmax on
    float pairs, accumulators rescaled only when the running maximum moves, V^T through the transposing LDS read - against the fp64
    attention with decomposed relative position bias on the same 16-bit q / k / v.  Scores are scaled up so that the running maximum
    does move between tiles for some queries and stays put for others (both sides of the conditional rescale)."""
    g = torch.Generator().manual_seed(11 + int(f16))
    heads, hd = 1, 64
    q, k, v = (torch.randn(heads, 4096, hd, generator=g) for _ in range(3))
    q = q * 1.5
    q, k, v = ((t.to(torch.float16) if f16 else t.to(torch.bfloat16)) for t in (q, k, v))
    rel_h, rel_w = torch.randn(127, hd, generator=g) * 0.15, torch.randn(127, hd, generator=g) * 0.15
    scale = 1.0 / math.sqrt(hd)
    out = np.zeros((4096, heads * hd), np.uint16)
    qa, ka, va = (t.view(torch.int16).numpy().view(np.uint16).copy() for t in (q, k, v))
    rha, rwa = _bits(rel_h, f16), _bits(rel_w, f16)
    n_wg = 5                                   # the first five of the 32 workgroups: image rows 0..9, all 4096 keys each
    emu.emu_global_attention(int(f16), _ptr(qa), _ptr(ka), _ptr(va), _ptr(rha), _ptr(rwa), n_wg, heads, ctypes.c_float(scale), _ptr(out))
    got = torch.from_numpy(out.view(np.int16)).view(torch.float16 if f16 else torch.bfloat16).double()
    qd, kd, vd = q[0].double(), k[0].double(), v[0].double()
    rh, rw = _round(rel_h, f16), _round(rel_w, f16)
    idx = torch.arange(64)
    d = (idx[:, None] - idx[None, :]) + 63
    qg = qd.reshape(64, 64, hd)
    th = torch.einsum("hwc,hkc->hwk", qg, rh[d])                               # [qh, qw, kh]
    tw = torch.einsum("hwc,wkc->hwk", qg, rw[d])                               # [qh, qw, kw]
    s = (scale * qd @ kd.t()).reshape(64, 64, 64, 64) + th[:, :, :, None] + tw[:, :, None, :]
    ref = torch.softmax(s.reshape(4096, 4096), dim=-1) @ vd
    rows = n_wg * 128
    assert float(got[rows:].abs().max()) == 0.0           # untouched
    err = (got[:rows] - ref[:rows]).abs().max().item()
    assert err <= (4e-3 if f16 else 2.5e-2) * ref.abs().max().item(), err
