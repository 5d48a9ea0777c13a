"""Kernel-level parity (through the C ABI) against plain torch fp32 on the same bf16-rounded operands.
Tolerances: fp32 outputs differ from torch only by accumulation order (abs 1e-3 on O(10) values); bf16 outputs by one
bf16 rounding (rel 1e-2)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from micro_sam_amd import ops
    return ops, torch.device("cuda")


def _bf(x):
    return x.to(torch.bfloat16)


def _ddt():
    """16-bit type of the decoder kernels in this library build (fp16 by default, bf16 with MSAM_DEC_F16 = 0)."""
    from micro_sam_amd import _lib
    return _lib.decoder_dtype()


def _d(x):
    return x.to(_ddt())


def _close(got, ref, atol, rtol):
    got, ref = got.float(), ref.float()
    return bool(torch.isfinite(got).all()) and bool(((got - ref).abs() <= atol + rtol * ref.abs()).all())


@pytest.mark.parametrize("glds", [0, 1])
@pytest.mark.parametrize("shape", [(128, 128, 64), (300, 128, 128), (4100, 256, 2304), (448, 2048, 256)])
def test_gemm_plain(env, shape, glds):
    ops, dev = env
    M, N, K = shape
    g = torch.Generator().manual_seed(0)
    a = _bf(torch.randn(M, K, generator=g)).to(dev)
    w = _bf(torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    ref = a.float() @ w.float().t() + bias
    assert _close(ops.gemm(a, w, bias, use_glds=glds), ref, 1e-3, 1e-4)


@pytest.mark.parametrize("glds", [0, 1])
def test_gemm_epilogues(env, glds):
    ops, dev = env
    g = torch.Generator().manual_seed(1)
    M, N, K = 8192, 256, 128
    a = _bf(torch.randn(M, K, generator=g)).to(dev)
    w = _bf(torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    table = torch.randn(4096, 128, generator=g).to(dev)
    resid_f = torch.randn(M, N, generator=g).to(dev)
    resid_b = _bf(torch.randn(4096, N, generator=g)).to(dev)
    base = a.float() @ w.float().t() + bias
    ref = base.clone(); ref[:, :128] += table.repeat(2, 1)
    assert _close(ops.gemm(a, w, bias, table=table, table_cols=128, use_glds=glds), ref, 1e-3, 1e-4)
    assert _close(ops.gemm(a, w, bias, resid=resid_f, use_glds=glds), base + resid_f, 1e-3, 1e-4)
    assert _close(ops.gemm(a, w, bias, resid=resid_b, resid_rows=4096, use_glds=glds),
                  base + resid_b.float().repeat(2, 1), 1e-3, 1e-4)
    assert _close(ops.gemm(a, w, bias, act=ops.ACT_GELU, out_dtype=torch.bfloat16, use_glds=glds), F.gelu(base), 2e-2, 1e-2)
    assert _close(ops.gemm(a, w, bias, act=ops.ACT_RELU, out_dtype=torch.bfloat16, use_glds=glds), F.relu(base), 2e-2, 1e-2)
    x = resid_f.clone()
    ops.gemm(a, w, bias, resid=x, out=x, use_glds=glds)          # in-place residual stream update
    assert _close(x, base + resid_f, 1e-3, 1e-4)


@pytest.mark.parametrize("glds", [0, 1])
def test_gemm_layout_epilogues(env, glds):
    ops, dev = env
    g = torch.Generator().manual_seed(2)
    B, heads, D = 2, 12, 768
    a = _bf(torch.randn(B * 4096, D, generator=g)).to(dev)
    w = _bf(torch.randn(3 * D, D, generator=g) / math.sqrt(D)).to(dev)
    bias = torch.randn(3 * D, generator=g).to(dev)
    ref = (a.float() @ w.float().t() + bias).reshape(B, 4096, 3, heads, 64).permute(2, 0, 3, 1, 4)
    for t, r in zip(ops.gemm_qkv(a, w, bias, B, heads, use_glds=glds), ref):
        assert _close(t, r, 3e-2, 1e-2)
    a = _bf(torch.randn(2 * 4096, 256, generator=g)).to(dev)
    w = _bf(torch.randn(256, 256, generator=g) / 16).to(dev)
    bias = torch.randn(256, generator=g).to(dev)
    table = torch.randn(4096, 128, generator=g).to(dev)
    ref = a.float() @ w.float().t() + bias
    ref[:, :128] += table.repeat(2, 1)
    k, vT = ops.gemm_kv(a, w, bias, table, 4096, use_glds=glds)
    assert _close(k, ref[:, :128], 3e-2, 1e-2)
    assert _close(vT, ref[:, 128:].reshape(2, 4096, 128).permute(0, 2, 1), 3e-2, 1e-2)


@pytest.fixture(params=[-1, 3, 4])
def staging256(request):
    """Operand staging / tiling variant of the large-shape GEMM (msam_gemm256_set_staging): -1 = the default, 3 = one 8-wave
    256 x 256 workgroup per CU (ping-pong of its two halves), 4 = two 4-wave 256 x 128 workgroups per CU (LDS-DMA ring, epilogue
    from the accumulators)."""
    from micro_sam_amd import _lib
    assert _lib.load().msam_gemm256_set_staging(request.param) == 0
    yield request.param
    _lib.load().msam_gemm256_set_staging(-1)


def test_gemm_256_tile_kernel(env, staging256):
    """Shapes with >= 256 tiles of 256 x 256 take gemm256_kernel / gemm2w_kernel (32x32x16 MFMA, transposed product): ragged M,
    every epilogue they support (bias, GELU -> bf16, fp32 residual in place, QKV head split), K with 1, 2, 3 and more k-tiles."""
    ops, dev = env
    g = torch.Generator().manual_seed(19)
    for K in (64, 128, 320):                              # 2 / 4 / 10 k-tiles of 32: prologue-only, ring wrap-around
        M, N = 16384 + 37, 1024
        a = _bf(torch.randn(M, K, generator=g)).to(dev)
        w = _bf(torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev)
        assert _close(ops.gemm(a, w), a.float() @ w.float().t(), 1e-3, 1e-4), K
    g = torch.Generator().manual_seed(9)
    M, N, K = 16384 + 100, 1024, 192                     # 65 x 4 = 260 tiles, last row tile ragged
    a = _bf(torch.randn(M, K, generator=g)).to(dev)
    w = _bf(torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    base = a.float() @ w.float().t() + bias
    assert _close(ops.gemm(a, w, bias), base, 1e-3, 1e-4)
    assert _close(ops.gemm(a, w, bias, act=ops.ACT_GELU, out_dtype=torch.bfloat16), F.gelu(base), 2e-2, 1e-2)
    x = torch.randn(M, N, generator=g).to(dev)
    ref = base + x
    ops.gemm(a, w, bias, resid=x, out=x)
    assert _close(x, ref, 1e-3, 1e-4)
    B, heads, D = 6, 4, 256                               # 96 x 3 = 288 tiles
    a = _bf(torch.randn(B * 4096, D, generator=g)).to(dev)
    w = _bf(torch.randn(3 * D, D, generator=g) / math.sqrt(D)).to(dev)
    bias = torch.randn(3 * D, generator=g).to(dev)
    ref = (a.float() @ w.float().t() + bias).reshape(B, 4096, 3, heads, 64).permute(2, 0, 3, 1, 4)
    for t, r in zip(ops.gemm_qkv(a, w, bias, B, heads), ref):
        assert _close(t, r, 3e-2, 1e-2)


def test_gemm_argument_errors(env):
    ops, dev = env
    a = torch.zeros(128, 64, dtype=torch.bfloat16, device=dev)
    w = torch.zeros(100, 64, dtype=torch.bfloat16, device=dev)      # N % 128 != 0
    with pytest.raises(ValueError):
        ops.gemm(a, w)


@pytest.mark.parametrize("dim", [64, 96, 256, 768, 1024, 1280])
def test_layernorm(env, dim):
    ops, dev = env
    g = torch.Generator().manual_seed(3)
    rows = 4096 if dim == 64 else 1001
    x = (torch.randn(rows, dim, generator=g) * 3 + 1).to(dev)
    w = torch.randn(dim, generator=g).to(dev); b = torch.randn(dim, generator=g).to(dev)
    ref = F.layer_norm(x, (dim,), w, b, eps=1e-6)
    assert _close(ops.layernorm(x, w, b, 1e-6), ref, 2e-5, 1e-5)
    assert _close(ops.layernorm(x, w, b, 1e-6, out_dtype=torch.bfloat16, gelu=True), F.gelu(ref), 2e-2, 1e-2)


def test_layernorm_nchw(env):
    ops, dev = env
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2 * 4096, 256, generator=g).to(dev)
    w = torch.randn(256, generator=g).to(dev); b = torch.randn(256, generator=g).to(dev)
    ref = F.layer_norm(x, (256,), w, b, eps=1e-6).reshape(2, 4096, 256).permute(0, 2, 1)
    assert _close(ops.layernorm(x, w, b, 1e-6, nchw_hw=4096), ref, 2e-5, 1e-5)


def test_patch_gather_and_im2col_are_exact(env):
    ops, dev = env
    from oracle import sam_ref as S
    g = torch.Generator().manual_seed(5)
    img = torch.randn(2, 3, 1024, 1024, generator=g).to(dev)
    ref = F.unfold(img, kernel_size=16, stride=16).permute(0, 2, 1).reshape(2 * 4096, 768)
    assert bool((ops.patchify(img).float() == _bf(ref).float()).all())
    u8 = torch.randint(0, 256, (2, 700, 1024, 3), generator=g, dtype=torch.uint8).to(dev)
    pre = S.preprocess(u8.permute(0, 3, 1, 2))                                    # Sam.preprocess restatement
    ref = F.unfold(pre, kernel_size=16, stride=16).permute(0, 2, 1).reshape(2 * 4096, 768)
    assert bool((ops.patchify_u8(u8).float() == _bf(ref).float()).all())
    x = _bf(torch.randn(2, 64, 64, 256, generator=g)).to(dev)
    ref = F.unfold(x.float().permute(0, 3, 1, 2), kernel_size=3, padding=1)
    ref = ref.reshape(2, 256, 9, 4096).permute(0, 3, 2, 1).reshape(2 * 4096, 9 * 256)
    assert bool((ops.im2col3x3(x).float() == ref).all())


@pytest.mark.parametrize("window", [True, False])
def test_vit_attention_vs_oracle(env, window):
    """QKV GEMM + attention kernel vs the oracle's attention (bf16 rounding points) with an identity out-projection."""
    ops, dev = env
    from oracle import sam_ref as S
    g = torch.Generator().manual_seed(6)
    B, heads, D = 1, 12, 768
    Sz = 14 if window else 64
    x = _bf(torch.randn(B, 64, 64, D, generator=g)).to(dev)
    qkv_w = (torch.randn(3 * D, D, generator=g) * 1.3 / math.sqrt(D)).to(dev)
    qkv_b = (torch.randn(3 * D, generator=g) * 0.3).to(dev)
    rel_h = (torch.randn(2 * Sz - 1, 64, generator=g) * 0.08).to(dev)
    rel_w = (torch.randn(2 * Sz - 1, 64, generator=g) * 0.08).to(dev)
    q, k, v = ops.gemm_qkv(x.reshape(-1, D), _bf(qkv_w), qkv_b, B, heads)
    if window:
        out = ops.window_attention(q, k, v, _bf(rel_h), _bf(rel_w), qkv_b)
    else:
        out = ops.global_attention(q, k, v, _bf(rel_h), _bf(rel_w))
    sd = {"a.qkv.weight": qkv_w, "a.qkv.bias": qkv_b, "a.rel_pos_h": rel_h, "a.rel_pos_w": rel_w,
          "a.proj.weight": torch.eye(D, device=dev), "a.proj.bias": torch.zeros(D, device=dev)}
    p = S.Prec("bf16")
    y = x.float()
    if window:
        yw, pad_hw = S._window_partition(y, 14)
        ref = S._window_unpartition(S._attention_relpos(sd, "a.", yw, heads, p), 14, pad_hw, (64, 64))
    else:
        ref = S._attention_relpos(sd, "a.", y, heads, p)
    assert _close(out, ref.reshape(-1, D), 2e-2, 2e-2)


@pytest.mark.parametrize("window", [True, False])
def test_vit_attention_head_dim_80(env, window):
    """vit_h geometry: 80-channel heads stored zero-padded to 96 channels (qkv rows / rel-pos columns), softmax scale of
    the true head_dim; the padded output channels are exactly zero and the rest matches the oracle's 80-channel attention."""
    import torch.nn.functional as F
    ops, dev = env
    from oracle import sam_ref as S
    g = torch.Generator().manual_seed(16)
    B, heads, hd, hs = 1, 4, 80, 96
    D = heads * hd
    Sz = 14 if window else 64
    x = _bf(torch.randn(B, 64, 64, D, generator=g)).to(dev)
    qkv_w = (torch.randn(3 * D, D, generator=g) * 1.3 / math.sqrt(D)).to(dev)
    qkv_b = (torch.randn(3 * D, generator=g) * 0.3).to(dev)
    rel_h = (torch.randn(2 * Sz - 1, hd, generator=g) * 0.08).to(dev)
    rel_w = (torch.randn(2 * Sz - 1, hd, generator=g) * 0.08).to(dev)
    w_pad = F.pad(qkv_w.reshape(3, heads, hd, D), (0, 0, 0, hs - hd)).reshape(3 * heads * hs, D)
    b_pad = F.pad(qkv_b.reshape(3, heads, hd), (0, hs - hd)).reshape(-1).contiguous()
    q, k, v = ops.gemm_qkv(x.reshape(-1, D), _bf(w_pad), b_pad, B, heads)
    assert q.shape == (B, heads, 4096, hs) and bool((q[..., hd:] == 0).all()) and bool((v[..., hd:] == 0).all())
    rh, rw = _bf(F.pad(rel_h, (0, hs - hd))), _bf(F.pad(rel_w, (0, hs - hd)))
    if window:
        out = ops.window_attention(q, k, v, rh, rw, b_pad, scale=hd ** -0.5)
    else:
        out = ops.global_attention(q, k, v, rh, rw, scale=hd ** -0.5)
    out = out.reshape(-1, heads, hs)
    assert bool((out[..., hd:] == 0).all())
    sd = {"a.qkv.weight": qkv_w, "a.qkv.bias": qkv_b, "a.rel_pos_h": rel_h, "a.rel_pos_w": rel_w,
          "a.proj.weight": torch.eye(D, device=dev), "a.proj.bias": torch.zeros(D, device=dev)}
    p = S.Prec("bf16")
    y = x.float()
    if window:
        yw, pad_hw = S._window_partition(y, 14)
        ref = S._window_unpartition(S._attention_relpos(sd, "a.", yw, heads, p), 14, pad_hw, (64, 64))
    else:
        ref = S._attention_relpos(sd, "a.", y, heads, p)
    assert _close(out[..., :hd].reshape(-1, D), ref.reshape(-1, D), 2e-2, 2e-2)


@pytest.mark.parametrize("K", [128, 256])
def test_gemm_fused_layernorm(env, K):
    """Row-complete 64x256 GEMM with LayerNorm(256) and LayerNorm(64 groups)+GELU epilogues (decoder norm4 / up-scaling)."""
    ops, dev = env
    g = torch.Generator().manual_seed(9)
    M = 4096 + 40                                   # M tail (not a multiple of 64)
    a = _bf(torch.randn(M, K, generator=g)).to(dev)
    w = _bf(torch.randn(256, K, generator=g) / math.sqrt(K)).to(dev)
    bias = torch.randn(256, generator=g).to(dev)
    resid = _bf(torch.randn(M, 256, generator=g)).to(dev)
    lw = (torch.randn(256, generator=g) * 0.2 + 1).to(dev); lb = torch.randn(256, generator=g).to(dev)
    pre = a.float() @ w.float().t() + bias + resid.float()
    ref = F.layer_norm(pre, (256,), lw, lb, eps=1e-5)
    out = ops.gemm(a, w, bias, resid=resid, out_dtype=torch.bfloat16, ln_mode=1, ln_w=lw, ln_b=lb, ln_eps=1e-5)
    assert _close(out, ref, 2e-2, 1e-2)
    out32 = ops.gemm(a, w, bias, resid=resid, out_dtype=torch.float32, ln_mode=1, ln_w=lw, ln_b=lb, ln_eps=1e-5)
    assert _close(out32, ref, 2e-4, 1e-4)
    x = resid.clone()                               # in-place stream update (layer 1 of the two-way transformer)
    ops.gemm(a, w, bias, resid=x, out=x, ln_mode=1, ln_w=lw, ln_b=lb, ln_eps=1e-5)
    assert _close(x, ref, 2e-2, 1e-2)
    pre2 = a.float() @ w.float().t() + bias
    ref2 = F.gelu(F.layer_norm(pre2.reshape(M, 4, 64), (64,), lw[:64], lb[:64], eps=1e-6)).reshape(M, 256)
    out2 = ops.gemm(a, w, bias, out_dtype=torch.bfloat16, ln_mode=2, ln_w=lw[:64].contiguous(), ln_b=lb[:64].contiguous(),
                    ln_eps=1e-6)
    assert _close(out2, ref2, 2e-2, 1e-2)


@pytest.mark.parametrize("N,K", [(256, 256), (256, 128), (128, 256), (128, 128)])
def test_wsgemm_epilogues(env, N, K):
    """Weights-stationary decoder GEMM: plain + table, residual + LayerNorm(256), LayerNorm(64)+GELU, K|V^T split."""
    ops, dev = env
    g = torch.Generator().manual_seed(10 + N + K)
    P, T = 3, 4096
    M = P * T
    a = _d(torch.randn(M, K, generator=g)).to(dev)
    w = _d(torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    table = torch.randn(T, 128, generator=g).to(dev)
    base = a.float() @ w.float().t() + bias
    ref = base.clone(); ref[:, :128] += table.repeat(P, 1)
    assert _close(ops.wsgemm(a, w, bias, table=table, table_cols=128), ref, 3e-2, 1e-2)
    if N == 256:
        lw = (torch.randn(256, generator=g) * 0.2 + 1).to(dev); lb = torch.randn(256, generator=g).to(dev)
        resid = _d(torch.randn(T, 256, generator=g)).to(dev)
        ref1 = F.layer_norm(base + resid.float().repeat(P, 1), (256,), lw, lb, eps=1e-5)
        assert _close(ops.wsgemm(a, w, bias, resid=resid, resid_rows=T, ln_mode=1, ln_w=lw, ln_b=lb), ref1, 2e-2, 1e-2)
        x = _d(torch.randn(M, 256, generator=g)).to(dev)
        ref1b = F.layer_norm(base + x.float(), (256,), lw, lb, eps=1e-5)
        ops.wsgemm(a, w, bias, resid=x, ln_mode=1, ln_w=lw, ln_b=lb, out=x)                 # in-place stream update
        assert _close(x, ref1b, 2e-2, 1e-2)
        ref2 = F.gelu(F.layer_norm(base.reshape(M, 4, 64), (64,), lw[:64], lb[:64], eps=1e-6)).reshape(M, 256)
        out2 = ops.wsgemm(a, w, bias, ln_mode=2, ln_w=lw[:64].contiguous(), ln_b=lb[:64].contiguous(), ln_eps=1e-6)
        assert _close(out2, ref2, 2e-2, 1e-2)
        k, vT = ops.wsgemm(a, w, bias, table=table, table_cols=128, kv_split_tokens=T)
        assert _close(k, ref[:, :128], 3e-2, 1e-2)
        assert _close(vT, ref[:, 128:].reshape(P, T, 128).permute(0, 2, 1), 3e-2, 1e-2)
    else:
        lw = (torch.randn(64, generator=g) * 0.2 + 1).to(dev); lb = torch.randn(64, generator=g).to(dev)
        ref2 = F.gelu(F.layer_norm(base.reshape(M, 2, 64), (64,), lw, lb, eps=1e-6)).reshape(M, 128)
        assert _close(ops.wsgemm(a, w, bias, ln_mode=2, ln_w=lw, ln_b=lb, ln_eps=1e-6), ref2, 2e-2, 1e-2)
        hm = ops.wsgemm(a, w, bias, head_major_tokens=T)                        # [P][8 heads][T][16]
        assert _close(hm.reshape(P, 8, T, 16), base.reshape(P, T, 8, 16).permute(0, 2, 1, 3), 3e-2, 1e-2)
        _, vT = ops.wsgemm(a, w, bias, kv_split_tokens=T)                      # N = 128: all columns transposed
        assert _close(vT, base.reshape(P, T, 128).permute(0, 2, 1), 3e-2, 1e-2)


@pytest.mark.parametrize("layer0", [True, False])
@pytest.mark.parametrize("Nt", [7, 12])
def test_decoder_image_layer_fused(env, layer0, Nt):
    """Fused q-proj + image->token attention + out-proj + LayerNorm vs plain torch on the same bf16-rounded operands."""
    ops, dev = env
    g = torch.Generator().manual_seed(20 + Nt + int(layer0))
    P, T = 2, 4096
    R = P * T
    x = _d(torch.randn(T if layer0 else R, 256, generator=g)).to(dev)
    ktok = _d(torch.randn(P * Nt, 128, generator=g)).to(dev)
    vtok = _d(torch.randn(P * Nt, 128, generator=g)).to(dev)
    wo = _d(torch.randn(256, 128, generator=g) / math.sqrt(128)).to(dev)
    bo = torch.randn(256, generator=g).to(dev)
    lw = (torch.randn(256, generator=g) * 0.2 + 1).to(dev); lb = torch.randn(256, generator=g).to(dev)
    wq = _d(torch.randn(128, 256, generator=g) / 16).to(dev)
    bq = torch.randn(128, generator=g).to(dev)
    peq = torch.randn(T, 128, generator=g).to(dev)
    if layer0:
        q_sh = _d(torch.randn(T, 128, generator=g)).to(dev)
        q = q_sh.float().repeat(P, 1)
        xres = x.float().repeat(P, 1)
        out = ops.decoder_image_layer(x, ktok, vtok, wo, bo, lw, lb, Nt, q_shared=q_sh, rows=R)
    else:
        q = _d(x.float() @ wq.float().t() + bq + peq.repeat(P, 1)).float()      # the kernel keeps q in bf16
        xres = x.float()
        out = ops.decoder_image_layer(x, ktok, vtok, wo, bo, lw, lb, Nt, wq=wq, bq=bq, peq=peq)
    qh = q.reshape(P, T, 8, 16).permute(0, 2, 1, 3)
    kh = ktok.float().reshape(P, Nt, 8, 16).permute(0, 2, 1, 3)
    vh = vtok.float().reshape(P, Nt, 8, 16).permute(0, 2, 1, 3)
    s = (qh @ kh.transpose(-1, -2)) / 4.0
    e = torch.exp(s - s.amax(-1, keepdim=True))
    attn = (_d(e).float() @ vh) / e.sum(-1, keepdim=True)                      # un-normalised P in bf16, fp32 row sum
    attn = _d(attn.permute(0, 2, 1, 3).reshape(R, 128)).float()
    ref = F.layer_norm(xres + attn @ wo.float().t() + bo, (256,), lw, lb, eps=1e-5)
    assert _close(out, ref, 3e-2, 2e-2)
    if not layer0:                                                               # in-place update of the stream
        x2 = x.clone()
        ops.decoder_image_layer(x2, ktok, vtok, wo, bo, lw, lb, Nt, wq=wq, bq=bq, peq=peq, out=x2)
        assert _close(x2, ref, 3e-2, 2e-2)


@pytest.mark.parametrize("P,Nt,shared", [(3, 7, False), (2, 8, False), (40, 5, True), (600, 7, True)])
def test_t2i_fold_attention(env, P, Nt, shared):
    """Folded token->image attention (one pass over the stream) vs the unfolded fp32 formulation of the reference:
    softmax((keys + pe) Wk^T + bk) . q / 4) (keys Wv^T + bv), all key-split counts (P = 3 -> 16 splits ... 600 -> 1)."""
    ops, dev = env
    g = torch.Generator().manual_seed(77 + P)
    T = 4096
    Pk = 1 if shared else P
    keys = _d(torch.randn(Pk, T, 256, generator=g)).to(dev)
    pe = torch.randn(T, 256, generator=g).to(dev)
    wk = _d(torch.randn(128, 256, generator=g) / 16).to(dev)
    wv = _d(torch.randn(128, 256, generator=g) / 16).to(dev)
    bk = torch.randn(128, generator=g).to(dev); bv = torch.randn(128, generator=g).to(dev)
    qtok = _d(torch.randn(P, Nt, 128, generator=g) * 1.5).to(dev)
    tab = pe @ wk.float().t() + bk
    tabk = tab.to(_ddt())
    out = ops.t2i_fold_attention(keys, qtok, wk, tabk, wv, bv, kv_shared=shared)
    kf = keys.float().expand(P, T, 256)
    K = kf @ wk.float().t() + tabk.float()                                      # [P,T,128]
    V = kf @ wv.float().t() + bv
    qh = qtok.float().reshape(P, Nt, 8, 16).permute(0, 2, 1, 3)
    kh = K.reshape(P, T, 8, 16).permute(0, 2, 1, 3)
    vh = V.reshape(P, T, 8, 16).permute(0, 2, 1, 3)
    a = torch.softmax((qh @ kh.transpose(-1, -2)) / 4.0, dim=-1)
    ref = (a @ vh).permute(0, 2, 1, 3).reshape(P, Nt, 128)
    # scores reach |s| ~ 30 with these operands: with a bf16 decoder the rounding of the folded query gives ~1e-2 relative score error
    assert _close(out, ref, 6e-2, 3e-2)
    # the LDS-DMA staging variant (tuning hook, off by default: measured 2.3x slower, profiles/r01_experiments.md) computes
    # the same arithmetic
    from micro_sam_amd import _lib
    _lib.load().msam_fold_attn_set_dma(1)
    try:
        out_dma = ops.t2i_fold_attention(keys, qtok, wk, tabk, wv, bv, kv_shared=shared)
    finally:
        _lib.load().msam_fold_attn_set_dma(0)
    assert torch.equal(out_dma, out)


@pytest.mark.parametrize("variant", [1, 0])
@pytest.mark.parametrize("P,Nt,shared", [(3, 7, False), (2, 8, True), (70, 5, True), (520, 7, True), (5, 1, False)])
def test_i2t_fold_layer(env, P, Nt, shared, variant):
    """Folded image->token attention + out_proj + residual + LayerNorm vs the unfolded fp32 formulation, for the token-owner
    kernel (variant 1, csrc/decfold_tok.hip: the default) and the 4-wave tile kernel (variant 0, csrc/decfold.hip)."""
    ops, dev = env
    from micro_sam_amd import _lib
    _lib.load().msam_tune_set(b"i2t_variant", variant)
    try:
        _i2t_fold_layer_case(ops, dev, P, Nt, shared)
    finally:
        _lib.load().msam_tune_set(b"i2t_variant", 1)


def _i2t_fold_layer_case(ops, dev, P, Nt, shared):
    g = torch.Generator().manual_seed(91 + P)
    T = 4096
    Px = 1 if shared else P
    x = _d(torch.randn(Px, T, 256, generator=g)).to(dev)
    pe = torch.randn(T, 256, generator=g).to(dev)
    wq = _d(torch.randn(128, 256, generator=g) / 16).to(dev); bq = torch.randn(128, generator=g).to(dev)
    wo = _d(torch.randn(256, 128, generator=g) / math.sqrt(128)).to(dev); bo = torch.randn(256, generator=g).to(dev)
    lw = (torch.randn(256, generator=g) * 0.2 + 1).to(dev); lb = torch.randn(256, generator=g).to(dev)
    ktok = _d(torch.randn(P, Nt, 128, generator=g)).to(dev)
    vtok = _d(torch.randn(P, Nt, 128, generator=g)).to(dev)
    tabq = (pe @ wq.float().t() + bq).to(_ddt())
    out = ops.i2t_fold_layer(x, ktok, vtok, wq, tabq, wo, bo, lw, lb, x_shared=shared)
    n = min(P, 4)                                                               # reference on the first / last prompts
    for sl in (slice(0, n), slice(P - n, P)):
        xf = x.float().expand(P, T, 256)[sl]
        q = xf @ wq.float().t() + tabq.float()
        qh = q.reshape(-1, T, 8, 16).permute(0, 2, 1, 3)
        kh = ktok[sl].float().reshape(-1, Nt, 8, 16).permute(0, 2, 1, 3)
        vh = vtok[sl].float().reshape(-1, Nt, 8, 16).permute(0, 2, 1, 3)
        a = torch.softmax((qh @ kh.transpose(-1, -2)) / 4.0, dim=-1)
        attn = (a @ vh).permute(0, 2, 1, 3).reshape(-1, T, 128)
        ref = F.layer_norm(xf + attn @ wo.float().t() + bo, (256,), lw, lb, eps=1e-5)
        assert _close(out[sl], ref, 5e-2, 3e-2)
    if not shared:                                                              # in-place update of the stream
        x2 = x.clone()
        ops.i2t_fold_layer(x2, ktok, vtok, wq, tabq, wo, bo, lw, lb, out=x2)
        assert torch.equal(x2, out)


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4, 5, 6, 7, 8, 9])
@pytest.mark.parametrize("P,Nt", [(130, 7), (129, 8), (3, 5)])
def test_chained_layer0_forms(env, P, Nt, variant):
    """All builds of the chained kernels (msam_tune_set "chain_variant": 0 = 8 waves with 4-fragment groups; 1 / 2 = 4 waves with
    rings of 8 / 16 fragments; 3 = 8 waves on the ring code; 4 / 5 = ring depth 3 / compact attention accumulators; 6 / 7 / 8 = tile
    loads one phase ahead with ring depth 2 / 3 / 3 + compact accumulators)."""
    from micro_sam_amd import _lib
    _lib.load().msam_tune_set(b"chain_variant", variant)
    try:
        _chained_layer0_case(env, P, Nt)
    finally:
        _lib.load().msam_tune_set(b"chain_variant", 9)


def _chained_layer0_case(env, P, Nt):
    """The chained kernels of csrc/decfold_tok.hip (layer-0 image->token block recomputed tile by tile from the shared source and
    fed into the layer-1 token->image attention / the layer-1 image->token block) against the stage-by-stage kernels and the fp32
    formulation."""
    ops, dev = env
    g = torch.Generator().manual_seed(300 + P)
    T = 4096
    src = _d(torch.randn(T, 256, generator=g)).to(dev)
    pe = torch.randn(T, 256, generator=g).to(dev)

    def layer():
        wq = _d(torch.randn(128, 256, generator=g) / 16).to(dev); bq = torch.randn(128, generator=g).to(dev)
        wo = _d(torch.randn(256, 128, generator=g) / math.sqrt(128)).to(dev); bo = torch.randn(256, generator=g).to(dev)
        lw = (torch.randn(256, generator=g) * 0.2 + 1).to(dev); lb = torch.randn(256, generator=g).to(dev)
        ktok = _d(torch.randn(P, Nt, 128, generator=g)).to(dev); vtok = _d(torch.randn(P, Nt, 128, generator=g)).to(dev)
        tabq = (pe @ wq.float().t() + bq).to(_ddt())
        return dict(wq=wq, wo=wo, bo=bo, lw=lw, lb=lb, ktok=ktok, vtok=vtok, tabq=tabq)

    L0, L1 = layer(), layer()
    q0 = (src.float() @ L0["wq"].float().t() + L0["tabq"].float()).to(_ddt())
    # stage by stage: layer-0 block on the shared source, then layer 1 on its output
    keys1 = ops.i2t_fold_layer(src[None], L0["ktok"], L0["vtok"], L0["wq"], L0["tabq"], L0["wo"], L0["bo"], L0["lw"], L0["lb"],
                               x_shared=True)
    keys2 = ops.i2t_fold_layer(keys1, L1["ktok"], L1["vtok"], L1["wq"], L1["tabq"], L1["wo"], L1["bo"], L1["lw"], L1["lb"])
    op0 = ops.i2t_fold_operands(L0["ktok"], L0["vtok"], L0["wq"], L0["wo"], L0["bo"], with_kfold=False)
    op1 = ops.i2t_fold_operands(L1["ktok"], L1["vtok"], L1["wq"], L1["wo"], L1["bo"])
    wk = _d(torch.randn(128, 256, generator=g) / 16).to(dev); bk = torch.randn(128, generator=g).to(dev)
    tabk = (pe @ wk.float().t() + bk).to(_ddt())
    tables = ops.chain_prepare_tables(src, q0, tabk, L1["tabq"])
    out2_blocked = ops.i2t01_fused(tables, op0, L0["lw"], L0["lb"], op1, L1["lw"], L1["lb"], P, Nt)
    out2 = ops.from_blocked(out2_blocked)
    # layer 0 takes its scores from the 16-bit q0 instead of the folded K' product: same arithmetic, other rounding points
    sel = [0, P // 2, P - 1]
    d2 = (out2[sel].float() - keys2[sel].float()).abs()
    assert float(d2.mean()) < 4e-3 and _close(out2[sel], keys2[sel], 6e-2, 3e-2)
    # fp32 formulation of both layers for a few prompts
    def block(xf, Lw, sl):
        q = xf @ Lw["wq"].float().t() + Lw["tabq"].float()
        qh = q.reshape(-1, T, 8, 16).permute(0, 2, 1, 3)
        kh = Lw["ktok"][sl].float().reshape(-1, Nt, 8, 16).permute(0, 2, 1, 3)
        vh = Lw["vtok"][sl].float().reshape(-1, Nt, 8, 16).permute(0, 2, 1, 3)
        a = torch.softmax((qh @ kh.transpose(-1, -2)) / 4.0, dim=-1)
        attn = (a @ vh).permute(0, 2, 1, 3).reshape(-1, T, 128)
        return F.layer_norm(xf + attn @ Lw["wo"].float().t() + Lw["bo"], (256,), Lw["lw"], Lw["lb"], eps=1e-5)
    ref1 = block(src.float()[None].expand(len(sel), T, 256), L0, sel)
    ref2 = block(ref1, L1, sel)
    assert _close(out2[sel], ref2, 8e-2, 4e-2)
    # token -> image attention of layer 1 on the (never written) layer-0 output
    wv = _d(torch.randn(128, 256, generator=g) / 16).to(dev); bv = torch.randn(128, generator=g).to(dev)
    qtok = _d(torch.randn(P, Nt, 128, generator=g) * 1.5).to(dev)
    att_stage = ops.t2i_fold_attention(keys1, qtok, wk, tabk, wv, bv)
    att = ops.i2t0_t2i_fused(tables, op0, L0["lw"], L0["lb"], qtok, wk, wv, bv)
    K = ref1 @ wk.float().t() + tabk.float()
    V = ref1 @ wv.float().t() + bv
    qh = qtok[sel].float().reshape(-1, Nt, 8, 16).permute(0, 2, 1, 3)
    kh = K.reshape(-1, T, 8, 16).permute(0, 2, 1, 3); vh = V.reshape(-1, T, 8, 16).permute(0, 2, 1, 3)
    a = torch.softmax((qh @ kh.transpose(-1, -2)) / 4.0, dim=-1)
    ref = (a @ vh).permute(0, 2, 1, 3).reshape(-1, Nt, 128)
    assert _close(att[sel], ref, 6e-2, 3e-2)
    assert _close(att, att_stage, 6e-2, 3e-2)
    # second form: norm4 folded into the operands, value projection before the LayerNorm
    t2 = ops.chain_prepare_tables2(src, wv, bv, wk, L0["lw"], L0["lb"], L0["wo"], L0["bo"])
    t2c, _ = ops.chain_prepare_tables2_cached(src, wv, bv, wk, L0["lw"], L0["lb"], L0["wo"], L0["bo"])     # the decoder's two-step path
    n_const, n_tabv = 128 * 128 * 4 + 4 * 512, 4096 * 128 * 2
    assert torch.equal(t2c[:n_const + n_tabv], t2[:n_const + n_tabv])
    mf = ops.t2i_fold_values(L0["vtok"], t2)
    op0b, mfb = ops.i2t_fold_operands_values(L0["ktok"], L0["vtok"], L0["wq"], L0["wo"], L0["bo"], t2)
    assert torch.equal(mfb, mf)
    o0, o0b = op0.view(P, -1), op0b.view(P, -1)                                    # (the K' part is not written without K fold)
    assert torch.equal(o0b[:, :8192], o0[:, :8192]) and torch.equal(o0b[:, 40960:], o0[:, 40960:])
    att2 = ops.i2t0_t2i_fused_v2(tables, t2, op0, mf, L0["lw"], qtok, wk)
    assert _close(att2[sel], ref, 6e-2, 3e-2)
    assert _close(att2, att_stage, 6e-2, 3e-2)
    # the consumers of the blocked stream: same bits as on the row-major copy of it
    assert torch.equal(ops.t2i_fold_attention(out2_blocked, qtok, wk, tabk, wv, bv, blocked=True),
                       ops.t2i_fold_attention(out2, qtok, wk, tabk, wv, bv))
    ct1 = _d(torch.randn(256, 256, generator=g) / 16).to(dev); cb1 = torch.randn(256, generator=g).to(dev)
    ulw = (torch.randn(64, generator=g) * 0.2 + 1).to(dev); ulb = (torch.randn(64, generator=g) * 0.3).to(dev)
    w2 = _d(torch.randn(128, 64, generator=g) / 8).to(dev); cb2 = torch.randn(32, generator=g).to(dev)
    hyper = torch.randn(P, 4, 128, generator=g).to(dev)
    assert torch.equal(ops.upscale_fused(out2_blocked, ct1, cb1, ulw, ulb, w2, cb2, hyper, 1, 3, blocked=True),
                       ops.upscale_fused(out2, ct1, cb1, ulw, ulb, w2, cb2, hyper, 1, 3))


@pytest.mark.parametrize("gelu16", [1, 0])
@pytest.mark.parametrize("P,mask0,nmask", [(2, 1, 3), (300, 1, 3)])
def test_upscale_fused_centred_weights(env, P, mask0, nmask, gelu16):
    """Round 6: the CEN instantiation of up_fused_kernel (centred first-layer weights, no LayerNorm2d mean: what the decoder runs) against the same
    torch formulation on the PLAIN weights - LayerNorm2d makes the two the same function - and next to the general kernel on the centred weights."""
    from micro_sam_amd import _lib
    _lib.load().msam_tune_set(b"up_gelu16", gelu16)
    try:
        _upscale_fused_case(env, P, mask0, nmask, centred=True)
    finally:
        _lib.load().msam_tune_set(b"up_gelu16", 1)


@pytest.mark.parametrize("gelu16", [1, 0])
@pytest.mark.parametrize("P,mask0,nmask", [(2, 1, 3), (3, 0, 1), (300, 1, 3)])
def test_upscale_fused(env, P, mask0, nmask, gelu16):
    """Fused ConvT + LayerNorm2d + GELU + ConvT + GELU + hyper product vs torch (conv_transpose2d on the same bf16 operands), with
    the GELUs in packed fp16 arithmetic (default of the fp16 decoder build) and in packed fp32."""
    from micro_sam_amd import _lib
    _lib.load().msam_tune_set(b"up_gelu16", gelu16)
    try:
        _upscale_fused_case(env, P, mask0, nmask)
    finally:
        _lib.load().msam_tune_set(b"up_gelu16", 1)


def _upscale_fused_case(env, P, mask0, nmask, centred=False):
    ops, dev = env
    g = torch.Generator().manual_seed(5 + P)
    keys = _d(torch.randn(P, 4096, 256, generator=g)).to(dev)
    ct1 = _d(torch.randn(256, 64, 2, 2, generator=g) / 16).to(dev); cb1 = torch.randn(64, generator=g).to(dev)
    lw = (torch.randn(64, generator=g) * 0.2 + 1).to(dev); lb = (torch.randn(64, generator=g) * 0.3).to(dev)
    ct2 = _d(torch.randn(64, 32, 2, 2, generator=g) / 8).to(dev); cb2 = torch.randn(32, generator=g).to(dev)
    hyper = torch.randn(P, 4, 128, generator=g).to(dev)
    w1 = ct1.permute(2, 3, 1, 0).reshape(256, 256).contiguous().to(_ddt())
    w2 = ct2.permute(2, 3, 1, 0).reshape(128, 64).contiguous().to(_ddt())
    if centred:
        ct1f = ct1.float().permute(2, 3, 1, 0).reshape(256, 256)
        w1c, b1c = ops.upscale_centre_weights(ct1f, cb1.repeat(4))
        out = ops.upscale_fused(keys, w1c, b1c, lw, lb, w2, cb2, hyper, mask0, nmask, centred=True)
        general = ops.upscale_fused(keys, w1c, b1c, lw, lb, w2, cb2, hyper, mask0, nmask)          # the general kernel computes the (zero) mean itself
        assert (out - general).abs().max().item() <= 6e-3 * general.abs().max().item()
    else:
        out = ops.upscale_fused(keys, w1, cb1.repeat(4).contiguous(), lw, lb, w2, cb2, hyper, mask0, nmask)
    sel = list(range(min(P, 2))) + ([P - 1] if P > 2 else [])
    src = keys[sel].float().transpose(1, 2).reshape(len(sel), 256, 64, 64)
    up = F.conv_transpose2d(src, ct1.float(), cb1, stride=2)
    mu = up.mean(1, keepdim=True); var = ((up - mu) ** 2).mean(1, keepdim=True)
    up = (up - mu) / torch.sqrt(var + 1e-6) * lw.view(1, -1, 1, 1) + lb.view(1, -1, 1, 1)
    up = _d(F.gelu(up)).float()
    up = F.gelu(F.conv_transpose2d(up, ct2.float(), cb2, stride=2))                   # [n,32,256,256]
    ref = torch.einsum("nmc,nchw->nmhw", hyper[sel][:, mask0:mask0 + nmask, :32], up)
    # stage 1 is rounded to bf16 in both; a rounding-boundary flip of one of the 64 stage-2 inputs moves an output by a few
    # 1e-3 of the output scale, so the maximum is bounded loosely and the mean error tightly
    scale = ref.abs().max().item()
    err = (out[sel] - ref).abs()
    assert err.max().item() <= 6e-3 * scale and err.mean().item() <= 3e-4 * scale


# ---------------------------------------------------------------------------------------------- fp8 (BASELINE config 5)

def test_fp8_row_quant_and_layernorm(env):
    """Row quantisation to OCP e4m3: scale = amax / 448, values round-to-nearest-even like torch's float8_e4m3fn cast."""
    ops, dev = env
    g = torch.Generator().manual_seed(31)
    x = _bf(torch.randn(1000, 3072, generator=g) * torch.rand(1000, 1, generator=g) * 5).to(dev)
    x[7] = 0                                                                     # all-zero row: scale 1, zeros out
    q, sc = ops.quant_rows_fp8(x)
    amax = x.float().abs().amax(1)
    ref_sc = torch.where(amax > 0, amax / 448.0, torch.ones_like(amax))
    assert torch.allclose(sc, ref_sc, rtol=1e-6, atol=0)
    ref_q = (x.float().cpu() * (1.0 / ref_sc.cpu())[:, None]).to(torch.float8_e4m3fn)            # cast on the host
    assert (q.cpu().float() != ref_q.float()).float().mean().item() <= 1e-4           # (the reciprocal differs from a division by 1 ulp)
    assert (q.float() * sc[:, None] - x.float()).abs().max().item() <= x.float().abs().max().item() * 0.07
    assert float(q[7].float().abs().max()) == 0.0 and float(sc[7]) == 1.0
    for D in (768, 1280):
        xf = (torch.randn(513, D, generator=g) * 3 + 0.5).to(dev)
        w = (torch.randn(D, generator=g) * 0.2 + 1).to(dev); b = (torch.randn(D, generator=g) * 0.1).to(dev)
        q, sc = ops.layernorm_fp8(xf, w, b)
        y = F.layer_norm(xf, (D,), w, b, eps=1e-6)
        ya = y.abs().amax(1)
        assert torch.allclose(sc, ya / 448.0, rtol=1e-4)
        d = (q.float() * sc[:, None] - y).abs()
        assert (d <= 0.0625 * y.abs() + ya[:, None] * 2.0 ** -9 + 1e-6).all()   # e4m3: 3 mantissa bits, subnormal step 2^-9


@pytest.mark.parametrize("M,N,K,mode", [(4096 + 24, 2304, 768, "plain"), (8192, 768, 3072, "resid"), (4096, 3072, 768, "gelu")])
def test_gemm_fp8_vs_torch(env, M, N, K, mode):
    """fp8 x fp8 GEMM on the MX MFMA (unit block scales) with row / column scales in the epilogue: exact fp8 operands, so the
    reference is a plain fp32 matmul of the dequantised operands (only the accumulation order differs)."""
    ops, dev = env
    g = torch.Generator().manual_seed(40 + N)
    a = _bf(torch.randn(M, K, generator=g)).to(dev)
    a8, a_sc = ops.quant_rows_fp8(a)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K) * (torch.rand(N, 1, generator=g) + 0.5)).to(dev)   # asymmetric rows
    w8, w_sc = ops.quant_weight_fp8(w.cpu())                                               # host-side cast
    w8, w_sc = w8.to(dev), w_sc.to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    ref = (a8.float() @ w8.float().t()) * a_sc[:, None] * w_sc[None, :] + bias
    if mode == "plain":
        out = ops.gemm_fp8(a8, a_sc, w8, w_sc, bias)
        assert _close(out, ref, 2e-3, 2e-3)
        # and the quantised product tracks the unquantised one to fp8 accuracy (sanity of the scales)
        full = a.float() @ w.t() + bias
        assert (out - full).abs().mean().item() <= 0.05 * full.abs().mean().item()
    elif mode == "resid":
        x = torch.randn(M, N, generator=g).to(dev)
        keep = x.clone()
        ops.gemm_fp8(a8, a_sc, w8, w_sc, bias, resid=x, out=x)                              # in place, like the encoder
        assert _close(x, ref + keep, 2e-3, 2e-3)
    else:
        out = ops.gemm_fp8(a8, a_sc, w8, w_sc, bias, act=ops.ACT_GELU, out_dtype=torch.bfloat16)
        assert _close(out, F.gelu(ref), 2e-2, 1e-2)


def test_gemm_split_k_matches_the_plain_product():
    """msam_gemm_t.split_k: a small output with a very long contraction (fine-tuning's weight gradients dW = dY^T X) cut into slices
    that accumulate with fp32 atomics == the same product on one workgroup chain (up to the summation order)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from micro_sam_amd import ops
    g = torch.Generator().manual_seed(12)
    for M, N, K, split in ((128, 256, 16384, 16), (200, 128, 8192, 8), (128, 128, 4096, 64)):
        a = (torch.randn(M, K, generator=g) * 0.5).to(torch.bfloat16).cuda()
        w = (torch.randn(N, K, generator=g) * 0.5).to(torch.bfloat16).cuda()
        ref = a.float() @ w.float().t()
        plain = ops.gemm(a, w, None, out_dtype=torch.float32)
        got = ops.gemm(a, w, None, out_dtype=torch.float32, split_k=split)
        scale = ref.abs().max().item()
        assert (plain - ref).abs().max().item() <= 2e-3 * scale
        assert (got - ref).abs().max().item() <= 2e-3 * scale and (got - plain).abs().max().item() <= 1e-4 * scale
    with pytest.raises((RuntimeError, ValueError)):
        ops.gemm(a, w, torch.zeros(N, device="cuda"), out_dtype=torch.float32, split_k=4)       # no bias in split-K mode
