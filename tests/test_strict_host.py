"""The strict (fp32) precision mode on the CPU: csrc/strict.hip compiled for the host (tests/hip_host_shim: the f32-input MFMA is emulated
as the k-ordered fmaf chain the device executes) and driven (1) kernel by kernel through the C ABI against fp64 / torch fp32 references and
(2) through the PRODUCT's own host layer (micro_sam_amd/strict.py behind Sam.set_precision("strict")) against the oracle's fp32 path -
the reference CPU path - for the mask decoder (points, boxes, mask prompts) and for a two-block ViT (one windowed block with the
zero-padded border, one global block).  TEST INFRASTRUCTURE: the product has no CPU path; the GPU suite repeats the end-to-end
comparisons on the device (tests/test_gpu_strict.py)."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from hip_host_shim import build_library

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
vp = C.c_void_p


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    from micro_sam_amd import _lib
    host = build_library(str(tmp_path_factory.mktemp("host_strict")), ROOT)
    for name, (res, args) in _lib._PROTOS.items():
        fn = getattr(host, name)
        fn.restype, fn.argtypes = res, args
    return host


def _p(t):
    return None if t is None else t.data_ptr()


def _gemm(lib, a, w, bias=None, act=0, a2=None, a2_rows=0, res=None, res_rows=0, split=False):
    from micro_sam_amd import _lib as L
    M, K = a.shape
    N = w.shape[0]
    out = torch.full((M, N), float("nan"))
    p = L.SGemmParams()
    p.A, p.lda, p.W, p.ldw, p.M, p.N, p.K = a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), M, N, K
    if a2 is not None:
        p.A2, p.lda2, p.a2_rows = a2.data_ptr(), a2.stride(0), a2_rows
    p.bias, p.act = _p(bias), act
    if res is not None:
        p.res, p.ldr, p.res_rows = res.data_ptr(), res.stride(0), res_rows
    p.out, p.ldc = out.data_ptr(), out.stride(0)
    if split:
        from micro_sam_amd import strict
        strict.forget_scales()
        p.split16, p.a_scale, p.w_scale = 1, 1.0, strict.weight_scale(w)
    assert lib.msam_strict_gemm(C.byref(p), None) == 0, lib.msam_last_error()
    return out


@pytest.mark.parametrize("split", [False, True])
@pytest.mark.parametrize("M,N,K", [(150, 200, 36), (5, 4, 256), (260, 32, 4), (128, 128, 64)])
@pytest.mark.parametrize("small_below", [0, 512])
def test_strict_gemm_every_epilogue(lib, M, N, K, small_below, split):
    g = torch.Generator().manual_seed(M + N + K)
    a, w = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / K ** 0.5
    bias, a2, res = torch.randn(N, generator=g), torch.randn(7, K, generator=g), torch.randn(11, N, generator=g)
    assert lib.msam_tune_set(b"sgemm_small_below", small_below) == 0          # 0: the 128 x 128 tile, 512: the 64 x 64 tile at these sizes
    try:
        _every_epilogue(lib, M, N, K, a, w, bias, a2, res, split)
    finally:
        assert lib.msam_tune_set(b"sgemm_small_below", 512) == 0


def _every_epilogue(lib, M, N, K, a, w, bias, a2, res, split=False):
    """split: the split16 form of the product (fp16 operand pairs, msam_sgemm_t.split16) - held to the SAME tolerance as the fp32 product."""
    y = a.double() @ w.double().t()
    tol = 3e-6 * max(1.0, y.abs().max().item())
    assert (_gemm(lib, a, w, split=split) - y).abs().max().item() <= tol
    rows = torch.arange(M)
    y2 = (a.double() + a2.double()[rows % 7]) @ w.double().t() + bias.double()
    out = _gemm(lib, a, w, bias, act=1, a2=a2, a2_rows=7, res=res, res_rows=11, split=split)
    want = torch.nn.functional.gelu(y2) + res.double()[rows % 11]
    assert (out - want).abs().max().item() <= 3e-6 * max(1.0, want.abs().max().item())
    out = _gemm(lib, a, w, bias, act=2, res=a.new_ones(M, N), res_rows=0, split=split)
    assert (out - (torch.relu(y + bias.double()) + 1)).abs().max().item() <= tol


def test_strict_gemm_is_the_fp32_fmaf_chain_and_refuses_bad_shapes(lib):
    """The emulated f32 MFMA is a k-ordered fmaf chain (what the device executes), so a product whose exact result needs more than 24
    bits differs from fp64 by fp32 roundings only - and K % 4 != 0 / misaligned rows are refused, not mis-read."""
    from micro_sam_amd import _lib as L
    a, w = torch.randn(4, 6), torch.randn(4, 6)
    p = L.SGemmParams()
    out = torch.zeros(4, 4)
    p.A, p.lda, p.W, p.ldw, p.M, p.N, p.K, p.out, p.ldc = a.data_ptr(), 6, w.data_ptr(), 6, 4, 4, 6, out.data_ptr(), 4
    assert lib.msam_strict_gemm(C.byref(p), None) == 1 and b"multiples of 4" in lib.msam_last_error()


@pytest.mark.parametrize("rows,dim,gelu", [(37, 768, 0), (50, 64, 1), (9, 256, 1), (6, 1280, 0)])
def test_strict_layernorm(lib, rows, dim, gelu):
    g = torch.Generator().manual_seed(dim)
    x = torch.randn(rows, dim, generator=g) * 2 + 0.7
    w, b = torch.randn(dim, generator=g) * 0.2 + 1, torch.randn(dim, generator=g)
    out = torch.empty_like(x)
    assert lib.msam_strict_layernorm(x.data_ptr(), w.data_ptr(), b.data_ptr(), 1e-6, rows, dim, out.data_ptr(), gelu, 0, None) == 0
    ref = torch.nn.functional.layer_norm(x.double(), (dim,), w.double(), b.double(), 1e-6)
    if gelu:
        ref = torch.nn.functional.gelu(ref)
    assert (out - ref).abs().max().item() <= 4e-6 * max(1.0, ref.abs().max().item())


def test_strict_layernorm_nchw_output(lib):
    x = torch.randn(2 * 16, 256)
    w, b = torch.rand(256) + 0.5, torch.randn(256)
    out = torch.empty(2, 256, 16)
    assert lib.msam_strict_layernorm(x.data_ptr(), w.data_ptr(), b.data_ptr(), 1e-6, 32, 256, out.data_ptr(), 0, 16, None) == 0
    ref = torch.nn.functional.layer_norm(x, (256,), w, b, 1e-6).reshape(2, 16, 256).permute(0, 2, 1)
    assert (out - ref).abs().max().item() <= 1e-5


@pytest.mark.parametrize("B,H,Nq,Nk,D,shared", [(3, 8, 4096 // 64, 7, 16, False), (2, 8, 7, 7, 32, False), (2, 8, 7, 300, 16, True),
                                               (2, 2, 12, 520, 16, False), (2, 8, 5, 520, 16, False), (1, 8, 16, 16, 32, False), (2, 8, 64, 13, 16, True)])
def test_strict_attention(lib, B, H, Nq, Nk, D, shared):
    """Both kernels (short key side: one thread per query and head; long key side: a workgroup per (batch, head)), with the shared
    (batch stride 0) operands of the first decoder layer."""
    g = torch.Generator().manual_seed(Nq * Nk + D)
    long_keys = Nk > 16
    q = torch.randn(1 if (shared and not long_keys) else B, Nq, H * D, generator=g)
    k = torch.randn(1 if (shared and long_keys) else B, Nk, H * D, generator=g)
    v = torch.randn_like(k)
    out = torch.full((B, Nq, H * D), float("nan"))
    denom = float(D) ** 0.5
    assert lib.msam_strict_attention(q.data_ptr(), H * D, 0 if q.shape[0] == 1 and B > 1 else Nq * H * D, k.data_ptr(), H * D,
                                     0 if k.shape[0] == 1 and B > 1 else Nk * H * D, v.data_ptr(), H * D,
                                     0 if k.shape[0] == 1 and B > 1 else Nk * H * D, B, H, Nq, Nk, D, denom, out.data_ptr(), H * D, Nq * H * D,
                                     None) == 0, lib.msam_last_error()

    def heads(t, n):
        return t.double().expand(B, -1, -1).reshape(B, n, H, D).transpose(1, 2)
    ref = torch.softmax(heads(q, Nq) @ heads(k, Nk).transpose(-1, -2) / denom, dim=-1) @ heads(v, Nk)
    ref = ref.transpose(1, 2).reshape(B, Nq, H * D)
    assert (out - ref).abs().max().item() <= 5e-6


def _relpos_ref(qkv, bqkv, rel_h, rel_w, B, heads, hd, G, window, scale):
    """oracle/sam_ref._attention_relpos + the window partition of image_encoder, fp64."""
    from oracle import sam_ref as S
    D = heads * hd
    x = qkv.double().reshape(B, G, G, 3 * D)
    if window:
        pad = (window - G % window) % window
        x = torch.nn.functional.pad(x, (0, 0, 0, pad, 0, pad))
        x[:, G:, :, :] = bqkv.double()
        x[:, :, G:, :] = bqkv.double()
        x, pad_hw = S._window_partition(x, window)
    Bp, S_, _, _ = x.shape
    t = x.reshape(Bp, S_ * S_, 3, heads, hd).permute(2, 0, 3, 1, 4).reshape(3, Bp * heads, S_ * S_, hd)
    q, k, v = t.unbind(0)
    attn = (q * scale) @ k.transpose(-2, -1)
    Rh, Rw = S._get_rel_pos(S_, S_, rel_h.double()), S._get_rel_pos(S_, S_, rel_w.double())
    rq = q.reshape(Bp * heads, S_, S_, hd)
    attn = (attn.view(-1, S_, S_, S_, S_) + torch.einsum("bhwc,hkc->bhwk", rq, Rh)[:, :, :, :, None]
            + torch.einsum("bhwc,wkc->bhwk", rq, Rw)[:, :, :, None, :]).view(-1, S_ * S_, S_ * S_)
    o = (attn.softmax(dim=-1) @ v).view(Bp, heads, S_, S_, hd).permute(0, 2, 3, 1, 4).reshape(Bp, S_, S_, D)
    if window:
        o = S._window_unpartition(o, window, pad_hw, (G, G))
    return o.reshape(B * G * G, D)


@pytest.mark.parametrize("B,heads,hd,G,window", [(2, 2, 64, 20, 14), (1, 1, 80, 30, 14), (1, 1, 64, 64, 0), (1, 1, 80, 64, 0)])
def test_strict_relpos_attention(lib, B, heads, hd, G, window):
    """Windowed attention on a grid that is not a multiple of the window (the border windows see the qkv bias as their padding tokens,
    exactly as zero-padding AFTER norm1 does in the reference) and the global 64 x 64 form; head_dim 64 and 80 (vit_h)."""
    g = torch.Generator().manual_seed(G + hd)
    D = heads * hd
    S_ = window if window else G
    qkv = torch.randn(B * G * G, 3 * D, generator=g)
    bqkv = torch.randn(3 * D, generator=g) * 0.5
    rel_h, rel_w = torch.randn(2 * S_ - 1, hd, generator=g) * 0.2, torch.randn(2 * S_ - 1, hd, generator=g) * 0.2
    scale = hd ** -0.5
    out = torch.full((B * G * G, D), float("nan"))
    assert lib.msam_strict_relpos_attention(qkv.data_ptr(), bqkv.data_ptr(), rel_h.data_ptr(), rel_w.data_ptr(), B, heads, hd, G, window,
                                            scale, out.data_ptr(), None) == 0, lib.msam_last_error()
    ref = _relpos_ref(qkv, bqkv, rel_h, rel_w, B, heads, hd, G, window, scale)
    assert torch.isfinite(out).all() and (out - ref).abs().max().item() <= 2e-5
    # both forms have two kernels: the transposed f32-MFMA formulation (srelpos_mfma_kernel / srelpos_win_mfma_kernel; the default) and the
    # vector-unit one
    out_v = torch.full((B * G * G, D), float("nan"))
    assert lib.msam_tune_set(b"srel_mfma", 0) == 0
    try:
        assert lib.msam_strict_relpos_attention(qkv.data_ptr(), bqkv.data_ptr(), rel_h.data_ptr(), rel_w.data_ptr(), B, heads, hd, G, window,
                                                scale, out_v.data_ptr(), None) == 0, lib.msam_last_error()
    finally:
        assert lib.msam_tune_set(b"srel_mfma", 2) == 0
    assert (out_v - ref).abs().max().item() <= 2e-5
    assert not torch.equal(out, out_v) and (out - out_v).abs().max().item() <= 3e-5      # two kernels, two summation orders
    # the split16 form of the MFMA kernels (q . k and p @ v on fp16 operand pairs): the same tolerance against fp64
    out_s = torch.full((B * G * G, D), float("nan"))
    assert lib.msam_split16_relpos_attention(qkv.data_ptr(), bqkv.data_ptr(), rel_h.data_ptr(), rel_w.data_ptr(), B, heads, hd, G, window,
                                             scale, out_s.data_ptr(), None) == 0, lib.msam_last_error()
    assert torch.isfinite(out_s).all() and (out_s - ref).abs().max().item() <= 2e-5, (out_s - ref).abs().max().item()


def test_strict_gemm_tile_variants_give_the_same_bits(lib):
    """sgemm_kernel<., 1, 2> (128 x 128 tile, one LDS stage, three workgroups per CU: the default of the large launches), <., 2, 2> (two
    stages) and <., 1, 1> (64 x 64 tiles: launches that would not fill the chip) walk k in the same order."""
    g = torch.Generator().manual_seed(11)
    a, w, b = torch.randn(300, 200, generator=g), torch.randn(150, 200, generator=g), torch.randn(150, generator=g)
    res = torch.randn(100, 150, generator=g)
    outs = []
    for bufs, small in ((1, 0), (2, 0), (1, 512)):
        assert lib.msam_tune_set(b"sgemm_bufs", bufs) == 0 and lib.msam_tune_set(b"sgemm_small_below", small) == 0
        try:
            outs.append(_gemm(lib, a, w, b, act=1, res=res, res_rows=100))
        finally:
            assert lib.msam_tune_set(b"sgemm_bufs", 1) == 0 and lib.msam_tune_set(b"sgemm_small_below", 512) == 0
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    ref = torch.nn.functional.gelu(a.double() @ w.double().T + b.double()) + res.double().repeat(3, 1)
    assert (outs[0] - ref).abs().max().item() <= 1e-4


def test_strict_gathers_and_hyper_product(lib):
    g = torch.Generator().manual_seed(3)
    # patch gather, fp32 and uint8 (Sam.preprocess fused)
    img = torch.randn(1, 3, 1024, 1024, generator=g)
    out = torch.empty(4096, 768)
    assert lib.msam_strict_patchify(img.data_ptr(), None, 1, 0, 0, out.data_ptr(), None) == 0
    want = img.reshape(1, 3, 64, 16, 64, 16).permute(0, 2, 4, 1, 3, 5).reshape(4096, 768)
    assert torch.equal(out, want)
    u8 = torch.randint(0, 256, (1, 700, 1000, 3), generator=g, dtype=torch.uint8)
    assert lib.msam_strict_patchify(None, u8.data_ptr(), 1, 700, 1000, out.data_ptr(), None) == 0
    from oracle import sam_ref as S
    pre = S.preprocess(u8.permute(0, 3, 1, 2).float())
    assert torch.equal(out, pre.reshape(1, 3, 64, 16, 64, 16).permute(0, 2, 4, 1, 3, 5).reshape(4096, 768))
    # 3 x 3 gather
    x = torch.randn(1, 64, 64, 8, generator=g)
    cols = torch.empty(4096, 72)
    assert lib.msam_strict_im2col3x3(x.data_ptr(), 1, 8, cols.data_ptr(), None) == 0
    wt = torch.randn(5, 8, 3, 3, generator=g)
    ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), wt.double(), padding=1)[0].permute(1, 2, 0).reshape(4096, 5)
    got = cols.double() @ wt.permute(0, 2, 3, 1).reshape(5, 72).double().t()
    assert (got - ref).abs().max().item() <= 1e-9
    # source stream
    emb, nm = torch.randn(256, 4096, generator=g), torch.randn(256, generator=g)
    src = torch.empty(4096, 256)
    assert lib.msam_strict_source(emb.data_ptr(), nm.data_ptr(), 0, 1, src.data_ptr(), None) == 0
    assert torch.equal(src, (emb + nm[:, None]).t())
    dn = torch.randn(2, 256, 4096, generator=g)
    src2 = torch.empty(2, 4096, 256)
    assert lib.msam_strict_source(emb.data_ptr(), dn.data_ptr(), 256 * 4096, 2, src2.data_ptr(), None) == 0
    assert torch.equal(src2, (emb[None] + dn).transpose(1, 2))
    # hyper product + un-shuffle of the two transposed convolutions' sub-pixels against conv_transpose2d itself
    P = 1
    keys = torch.randn(P, 4096, 16, generator=g)                      # a 16-channel stand-in for the image-token stream
    w1, w2 = torch.randn(16, 64, 2, 2, generator=g) / 4, torch.randn(64, 32, 2, 2, generator=g) / 8
    u1 = keys.reshape(P * 4096, 16) @ w1.permute(2, 3, 1, 0).reshape(256, 16).t()            # [P*4096, (ky,kx,co)]
    u2 = u1.reshape(P * 4096 * 4, 64) @ w2.permute(2, 3, 1, 0).reshape(128, 64).t()          # [(P*4096*4), (ky2,kx2,c2)]
    hyper = torch.randn(P, 4, 32, generator=g)
    low = torch.full((P, 3, 256, 256), float("nan"))
    assert lib.msam_strict_hyper_masks(u2.contiguous().data_ptr(), hyper.data_ptr(), 32, 1, 3, P, low.data_ptr(), None) == 0
    up = torch.nn.functional.conv_transpose2d(
        torch.nn.functional.conv_transpose2d(keys.transpose(1, 2).reshape(P, 16, 64, 64).double(), w1.double(), stride=2), w2.double(), stride=2)
    ref = (hyper.double()[:, 1:4] @ up.reshape(P, 32, 65536)).reshape(P, 3, 256, 256)
    assert (low - ref).abs().max().item() <= 1e-4 * ref.abs().max().item()


# ------------------------------------------------------------------------------------------------ the product's host layer

@pytest.fixture(scope="module")
def host_sam(lib):
    from micro_sam_amd import _lib, modeling
    from micro_sam_amd.synthetic import synthetic_state_dict
    os.environ["MSAM_EMU_CUS"] = "4"
    saved = (_lib._lib, _lib.require_gpu, _lib.stream_ptr, _lib.ptr)
    _lib._lib = lib
    _lib.require_gpu = lambda device=None: torch.device("cpu") if device is None else torch.device(device)
    _lib.stream_ptr = lambda: None
    _lib.ptr = lambda t: None if t is None else t.data_ptr()
    sd = synthetic_state_dict("vit_b", 0, variant="cells")
    g = torch.Generator().manual_seed(77)
    for k in list(sd):                                   # generic (not exactly representable) decoder weights: a 1 % perturbation
        if k.startswith(("mask_decoder.", "prompt_encoder.")) and sd[k].dtype == torch.float32 and "gaussian" not in k:
            sd[k] = sd[k] * (1 + 0.01 * torch.randn(sd[k].shape, generator=g))
    sam = modeling.build_sam("vit_b")
    sam.load_state_dict(sd)
    sam.eval()
    try:
        yield lib, sam, sd
    finally:
        _lib._lib, _lib.require_gpu, _lib.stream_ptr, _lib.ptr = saved
        os.environ.pop("MSAM_EMU_CUS", None)


# (the mask-prompt case runs in the strict mode only: the split16 mode shares its per-prompt-source path with "box+points" - CPU suite time)
@pytest.mark.parametrize("kind,mode", [("points", "strict"), ("box+points", "strict"), ("mask", "strict"), ("points", "split16"), ("box+points", "split16")])
def test_strict_decode_is_the_oracles_fp32_decoder(host_sam, kind, mode):
    """Sam.set_precision("strict") -> Sam.decode = the reference's un-folded two-way transformer + up-scaling in fp32: low-res logits and
    IoU predictions equal the oracle's fp32 path to fp32 rounding (the default 16-bit path: 3e-2 of the logit scale)."""
    from oracle import sam_ref as S
    _, sam, sd = host_sam
    g = torch.Generator().manual_seed(len(kind))
    feats = torch.randn(1, 256, 64, 64, generator=g) * 0.6
    P = 2
    pts = torch.rand(P, 2 if kind == "box+points" else 1, 2, generator=g) * 1024
    lbl = torch.ones(P, pts.shape[1], dtype=torch.int)
    boxes = mask_in = None
    if kind == "box+points":
        x0 = torch.rand(P, 2, generator=g) * 500
        boxes = torch.cat([x0, x0 + 300], dim=1)
    if kind == "mask":
        mask_in = torch.randn(P, 1, 256, 256, generator=g) * 6
    with torch.no_grad():
        _, iou_r, low_r = S.predict_torch(sd, feats, (1024, 1024), (1024, 1024), pts, lbl, boxes, mask_in, return_logits=True, precision="fp32")
    from micro_sam_amd import strict
    sam.set_precision(mode)                             # split16: every product on fp16 operand pairs - the same tolerances hold
    strict.FUSED_KV = kind == "box+points"              # (one case through the one-launch k | v projection, off by default)
    try:
        low, iou = sam.decode(feats, pts, lbl, boxes, mask_in)
    finally:
        sam.set_precision("default")
        strict.FUSED_KV = False
    scale = low_r.abs().max().item()
    d = (low - low_r).abs()
    # fp32 rounding through ~40 dependent products (logits of +-130: one ulp is 1.5e-5): max <= 1e-4, mean <= 2e-6 of the scale
    tol, tol_mean = 1e-4, 2e-6
    assert torch.isfinite(low).all() and d.max().item() <= tol * scale and d.mean().item() <= tol_mean * scale, \
        (d.max().item() / scale, d.mean().item() / scale)
    assert (iou - iou_r).abs().max().item() <= 2e-5


def test_strict_gemm_a2_cols_is_two_products(host_sam):
    """msam_sgemm_t.a2_cols: `(x + pe) Wk^T | x Wv^T` over [Wk; Wv] in one launch = the two launches, bit for bit; bad values are refused."""
    from micro_sam_amd import _lib, strict
    g = torch.Generator().manual_seed(4)
    x, pe = torch.randn(700, 96, generator=g), torch.randn(100, 96, generator=g)
    wk, wv, bk, bv = torch.randn(128, 96, generator=g), torch.randn(256, 96, generator=g), torch.randn(128, generator=g), torch.randn(256, generator=g)
    kv = strict.gemm(x, torch.cat([wk, wv]), torch.cat([bk, bv]), a2=pe, a2_rows=100, a2_cols=128)
    assert torch.equal(kv[:, :128], strict.gemm(x, wk, bk, a2=pe, a2_rows=100)) and torch.equal(kv[:, 128:], strict.gemm(x, wv, bv))
    with pytest.raises(ValueError, match="a2_cols"):
        strict.gemm(x, torch.cat([wk, wv]), torch.cat([bk, bv]), a2=pe, a2_rows=100, a2_cols=100)


@pytest.mark.parametrize("shared,Tk,split", [(True, 7, False), (False, 9, False), (False, 16, False), (True, 7, True), (False, 16, True), (False, 5, True)])
def test_strict_i2t_block_is_the_four_launches(host_sam, shared, Tk, split):
    """msam_strict_i2t_block (projection, 8-head attention over <= 16 tokens, projection + residual, LayerNorm in one launch, transposed
    MFMA orientation) against the same step as four launches; layer 0's shared stream and the in-place per-prompt stream."""
    from micro_sam_amd import strict
    g = torch.Generator().manual_seed(Tk)
    B = 2
    keys = torch.randn(4096 if shared else B * 4096, 256, generator=g)
    pos = torch.randn(4096, 256, generator=g)
    wq, wo = (torch.randn(128, 256, generator=g) / 16, torch.randn(128, generator=g) * 0.1), (torch.randn(256, 128, generator=g) / 11, torch.randn(256, generator=g) * 0.1)
    tok_k, tok_v = torch.randn(B * Tk, 128, generator=g), torch.randn(B * Tk, 128, generator=g)
    norm = (torch.rand(256, generator=g) + 0.5, torch.randn(256, generator=g) * 0.2, 1e-5)
    strict.forget_scales()
    prepared = strict.PREPARED_WEIGHTS
    strict.PREPARED_WEIGHTS = split and Tk == 16        # one case through the prepared weight pairs (msam_split16_prepare_pairs; off by default)
    with strict.split_mode(split):                      # split: the split16 form of both (fp16 operand pairs; same tolerance: fp32-level accuracy)
        q = strict.gemm(keys, *wq, a2=pos, a2_rows=4096)
        att = strict.attention(q, tok_k, tok_v, B, 8, 4096, Tk, 16, 4.0, q_shared=shared)
        want = strict.gemm(att, *wo, res=keys, res_rows=4096 if shared else 0)
        strict.layer_norm(want, *norm, out=want)
        got = strict.i2t_block(keys.clone(), shared, pos, wq, tok_k, tok_v, wo, norm, B, Tk)
        if split and Tk <= 8:
            # Tk <= 8 went through the folded kernel (msam_split16_i2t_block); the un-folded one-launch form must agree with it
            strict.FOLDED_I2T = False
            try:
                got_unfolded = strict.i2t_block(keys.clone(), shared, pos, wq, tok_k, tok_v, wo, norm, B, Tk)
            finally:
                strict.FOLDED_I2T = True
            assert (got - got_unfolded).abs().max().item() <= 2e-5
    strict.PREPARED_WEIGHTS = prepared
    assert got.shape == want.shape and torch.isfinite(got).all()
    assert (got - want).abs().max().item() <= 2e-5, (got - want).abs().max().item()
    # fp64 statement of the step
    kd = keys.double().reshape(-1, 4096, 256).expand(B, -1, -1)
    qd = ((kd + pos.double()) @ wq[0].double().T + wq[1].double()).reshape(B, 4096, 8, 16).transpose(1, 2)
    kk, vv = (t.double().reshape(B, Tk, 8, 16).transpose(1, 2) for t in (tok_k, tok_v))
    ad = (torch.softmax(qd @ kk.transpose(-1, -2) / 4.0, dim=-1) @ vv).transpose(1, 2).reshape(B, 4096, 128)
    ref = torch.nn.functional.layer_norm(kd + ad @ wo[0].double().T + wo[1].double(), (256,), norm[0].double(), norm[1].double(), 1e-5)
    assert (got.double() - ref.reshape(-1, 256)).abs().max().item() <= 2e-5


@pytest.mark.parametrize("mask0,nmask", [(1, 3), (0, 1)])
def test_split16_upscale2_is_the_three_launches(host_sam, mask0, nmask):
    """msam_strict_upscale2 (LayerNorm2d, GELU, second transposed convolution, GELU, hyper product, un-shuffle in one launch; transposed
    split16 product, the stream read straight from global memory) against the same steps as three launches and against fp64."""
    from micro_sam_amd import _lib as L
    from micro_sam_amd import strict
    lib, _, _ = host_sam
    g = torch.Generator().manual_seed(mask0 + 10 * nmask)
    P = 1
    u1 = torch.randn(P * 16384, 64, generator=g) * torch.exp(0.5 * torch.randn(P * 16384, 1, generator=g))
    lnw, lnb = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.2
    w2, b2 = torch.randn(128, 64, generator=g) / 8, torch.randn(128, generator=g) * 0.1
    hyper = torch.randn(P, 4, 32, generator=g)
    low = torch.full((P, nmask, 256, 256), float("nan"))
    strict.forget_scales()
    q = L.SUp2Params()
    q.u1, q.ln_weight, q.ln_bias, q.ln_eps = u1.data_ptr(), lnw.data_ptr(), lnb.data_ptr(), 1e-6
    q.w2, q.b2, q.w_scale = w2.data_ptr(), b2.data_ptr(), strict.weight_scale(w2)
    q.hyper, q.hyper_ld, q.mask0, q.nmask, q.low_res, q.P = hyper.data_ptr(), 32, mask0, nmask, low.data_ptr(), P
    assert lib.msam_strict_upscale2(C.byref(q), None) == 0, lib.msam_last_error()
    with strict.split_mode(True):
        t = strict.layer_norm(u1.clone(), lnw, lnb, 1e-6, gelu=True)
        t = strict.gemm(t, w2, b2, act=strict.ACT_GELU)
    want = torch.full((P, nmask, 256, 256), float("nan"))
    assert lib.msam_strict_hyper_masks(t.data_ptr(), hyper.data_ptr(), 32, mask0, nmask, P, want.data_ptr(), None) == 0
    assert torch.isfinite(low).all()
    scale = want.abs().max().item()
    assert (low - want).abs().max().item() <= 2e-6 * scale, (low - want).abs().max().item() / scale
    # fp64 statement
    xd = torch.nn.functional.gelu(torch.nn.functional.layer_norm(u1.double(), (64,), lnw.double(), lnb.double(), 1e-6))
    up = torch.nn.functional.gelu(xd @ w2.double().T + b2.double()).reshape(P, 64, 64, 2, 2, 2, 2, 32)      # p, ty, tx, ky, kx, ky2, kx2, c
    ref = torch.einsum("pmc,pyxabdec->pmyadxbe", hyper.double()[:, mask0:mask0 + nmask], up).reshape(P, nmask, 256, 256)
    assert (low.double() - ref).abs().max().item() <= 3e-6 * scale
    q.nmask = 5
    assert lib.msam_strict_upscale2(C.byref(q), None) == 1


def test_strict_module_call_and_single_mask(host_sam):
    """``sam.mask_decoder(...)`` (the stand-alone module call of trainable_sam.py:100-106) in strict mode, multimask_output=False."""
    from oracle import sam_ref as S
    _, sam, sd = host_sam
    g = torch.Generator().manual_seed(21)
    feats = torch.randn(1, 256, 64, 64, generator=g) * 0.6
    pts, lbl = torch.rand(1, 1, 2, generator=g) * 1024, torch.ones(1, 1, dtype=torch.int)
    with torch.no_grad():
        sparse, dense = S.prompt_encoder(sd, (pts, lbl), None, None)
        low_r, iou_r = S.mask_decoder(sd, feats, S.get_dense_pe(sd), sparse, dense, False, "fp32")
    sam.set_precision("strict")
    try:
        low, iou = sam.mask_decoder(feats, sam.prompt_encoder.get_dense_pe(), sparse, dense, False)
    finally:
        sam.set_precision("default")
    assert low.shape == (1, 1, 256, 256) and (low - low_r).abs().max().item() <= 1e-4 * low_r.abs().max().item()
    assert (iou - iou_r).abs().max().item() <= 2e-5


def test_strict_encoder_is_the_oracles_fp32_encoder(host_sam):
    """A two-block ViT (64 channels, one head; block 0 windowed, block 1 global) through ImageEncoderViT.set_precision("fp32") - patch
    gather with the fused Sam.preprocess, pos_embed, both attention forms, MLP, neck - against the oracle's fp32 image_encoder."""
    from micro_sam_amd import modeling
    from oracle import sam_ref as S
    g = torch.Generator().manual_seed(5)
    enc = modeling.ImageEncoderViT(64, 2, 1, (1,))
    with torch.no_grad():
        for p in enc.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * (0.05 if p.dim() > 1 else 0.3) + (1.0 if p.dim() == 1 and p.shape[0] in (64, 256) and False else 0.0))
        for blk in enc.blocks:
            blk.norm1.weight.add_(1.0); blk.norm2.weight.add_(1.0)
        enc.neck[1].weight.add_(1.0); enc.neck[3].weight.add_(1.0)
    sd = {"image_encoder." + k: v.detach() for k, v in enc.state_dict().items()}
    S.VIT_CONFIGS["vit_x"] = dict(embed_dim=64, depth=2, num_heads=1, global_attn_indexes=(1,))
    u8 = torch.randint(0, 256, (1, 1024, 900, 3), generator=g, dtype=torch.uint8)
    try:
        with torch.no_grad():
            ref = S.image_encoder(sd, S.preprocess(u8.permute(0, 3, 1, 2).float()), "vit_x", "fp32")
    finally:
        S.VIT_CONFIGS.pop("vit_x")
    enc.set_precision("fp32")
    out = enc.forward_u8(u8)
    assert out.shape == (1, 256, 64, 64)
    assert (out - ref).abs().max().item() <= 3e-5 * max(1.0, ref.abs().max().item()), (out - ref).abs().max().item()
    out2 = enc(S.preprocess(u8.permute(0, 3, 1, 2).float()))
    assert torch.equal(out, out2)


@pytest.mark.parametrize("Tk", [7, 3])
def test_split16_t2i_attention_is_the_projections_and_the_attention(host_sam, Tk):
    """msam_split16_t2i_attention (k / v projections folded into the token side, one pass over the per-prompt image stream with an online
    softmax, MFMA products on fp16 pairs) against the unfused strict steps (k | v projection, sattn_long) and an fp64 statement."""
    from micro_sam_amd import _lib as L
    from micro_sam_amd import strict
    lib, _, _ = host_sam
    g = torch.Generator().manual_seed(30 + Tk)
    B = 2
    keys = torch.randn(B * 4096, 256, generator=g) * torch.exp(0.3 * torch.randn(B * 4096, 1, generator=g))
    pos = torch.randn(4096, 256, generator=g)
    q = torch.randn(B * Tk, 128, generator=g) * 1.5
    wk, bk = torch.randn(128, 256, generator=g) / 16, torch.randn(128, generator=g) * 0.1
    wv, bv = torch.randn(128, 256, generator=g) / 16, torch.randn(128, generator=g) * 0.1
    out = torch.full((B * Tk, 128), float("nan"))
    ws = torch.zeros(B * 131072, dtype=torch.uint8)
    p = L.ST2IParams()
    p.keys, p.key_batch_stride, p.pos, p.q, p.ldq = keys.data_ptr(), 4096 * 256, pos.data_ptr(), q.data_ptr(), 128
    p.wk, p.wv, p.bv, p.denom, p.out, p.ldo, p.B, p.Tk = wk.data_ptr(), wv.data_ptr(), bv.data_ptr(), 4.0, out.data_ptr(), 128, B, Tk
    p.workspace, p.workspace_bytes = ws.data_ptr(), ws.numel()
    assert lib.msam_split16_t2i_attention(C.byref(p), None) == 0, lib.msam_last_error()
    assert torch.isfinite(out).all()
    # fp64 statement of upstream Attention.forward (q already projected)
    kd = ((keys.double().reshape(B, 4096, 256) + pos.double()) @ wk.double().T + bk.double()).reshape(B, 4096, 8, 16).transpose(1, 2)
    vd = (keys.double().reshape(B, 4096, 256) @ wv.double().T + bv.double()).reshape(B, 4096, 8, 16).transpose(1, 2)
    qd = q.double().reshape(B, Tk, 8, 16).transpose(1, 2)
    ref = (torch.softmax(qd @ kd.transpose(-1, -2) / 4.0, dim=-1) @ vd).transpose(1, 2).reshape(B * Tk, 128)
    # the unfused strict steps (fp32 products)
    k = strict.gemm(keys, wk, bk, a2=pos, a2_rows=4096)
    v = strict.gemm(keys, wv, bv)
    want = strict.attention(q, k, v, B, 8, Tk, 4096, 16, 4.0)
    strict.forget_scales()
    with strict.split_mode(True):                       # the unfused steps with split16 products: the error class the fused kernel belongs to
        want16 = strict.attention(q, strict.gemm(keys, wk, bk, a2=pos, a2_rows=4096), strict.gemm(keys, wv, bv), B, 8, Tk, 4096, 16, 4.0)
    err, err_unfused = (out.double() - ref).abs().max().item(), (want.double() - ref).abs().max().item()
    err16 = (want16.double() - ref).abs().max().item()
    print(f"\nt2i Tk={Tk}: fused split16 vs fp64 {err:.2e}, unfused fp32 steps vs fp64 {err_unfused:.2e}, unfused split16 steps {err16:.2e}")
    # scores of +-20 with these operands, some rows with one probability of 0.99: the fused kernel sums the softmax denominator with
    # compensation (without it the 2047 tiny exponentials a lane adds to a sum near 1 lost 1e-5 of their mass: 5e-5 on the output)
    assert err <= 2e-5 and err <= 2 * max(err_unfused, err16) and (out - want).abs().max().item() <= 3e-5
    p.Tk = 9
    assert lib.msam_split16_t2i_attention(C.byref(p), None) == 1
