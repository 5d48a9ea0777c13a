"""gemm_kernel (csrc/gemm.hip: the 128 x 128 x 64 MFMA tile kernel behind msam_gemm_bf16 - patch embedding, neck, the decoder's token
side, every product of the training path) executed on the CPU behind tests/hip_host_shim.py: operand staging with the chunk swizzle
(register staging and the LDS-DMA form), the 16x16x32 MFMA in its register layout, the epilogues (bias, row table, fp32 / 16-bit residual,
GELU / ReLU, fp32 / bf16 output, q / k / v head split, split-K accumulation) on a ragged M - against fp64 products of the same bf16
operands.  Device run of the same kernel: tests/test_gpu_kernels.py."""
import ctypes
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from hip_host_shim import build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

ENTRY = r"""
extern "C" void emu_gemm(int glds, const u16* A, long lda, const u16* W, long ldw, int M, int N, int K, const float* bias, const float* table,
                         int table_rows, int table_cols, long table_ld, const void* resid, int resid_dtype, int resid_rows, long ldr, int act,
                         void* out, int out_dtype, long ldc, int out_mode, u16* q, u16* k, u16* v, int heads, int head_dim, int tokens,
                         int split_k) {
    Epi e{};
    e.bias = bias; e.table = table; e.table_rows = table_rows; e.table_cols = table_cols; e.table_ld = table_ld;
    e.resid = resid; e.resid_dtype = resid_dtype; e.resid_rows = resid_rows; e.ldr = ldr; e.act = act;
    e.out = out; e.out_dtype = out_dtype; e.ldc = ldc; e.out_mode = out_mode; e.q = q; e.k = k; e.v = v;
    e.heads = heads; e.head_dim = head_dim; e.tokens = tokens; e.row_scale = nullptr; e.col_scale = nullptr;
    e.splitk_len = split_k > 1 ? K / split_k : 0;
    const int tiles = ((M + 127) / 128) * (N / 128);
    if (glds) launch_grid(tiles, split_k > 1 ? split_k : 1, [=] { gemm_kernel<true, false>(A, lda, W, ldw, M, N, K, e); });
    else launch_grid(tiles, split_k > 1 ? split_k : 1, [=] { gemm_kernel<false, false>(A, lda, W, ldw, M, N, K, e); });
}
extern "C" void emu_gemm_ln(const u16* A, long lda, const u16* W, long ldw, int M, int K, const float* bias, void* out, int out_dtype,
                            const float* ln_w, const float* ln_b, float eps, int mode, const float* add, u16* out_a, u16* out_b) {
    Epi e{};
    e.bias = bias; e.out = out; e.out_dtype = out_dtype; e.ldc = 256;
    EpiLN ln{ln_w, ln_b, eps, mode, add, out_a, out_b, MSAM_BF16};
    launch_grid((M + 63) / 64, 1, [=] { gemm_ln_kernel<false>(A, lda, W, ldw, M, K, e, ln); });
}
// the two-workgroups-per-CU kernel (staging 4): persistent grid of `grid` workgroups over the 256 x 128 tiles
extern "C" void emu_gemm2w(int grid, const u16* A, long lda, const u16* W, long ldw, int M, int N, int K, const float* bias,
                           const void* resid, int resid_dtype, long ldr, int act, void* out, int out_dtype, long ldc, int out_mode, u16* q,
                           u16* k, u16* v, int heads, int head_dim, int tokens) {
    Epi e{};
    e.bias = bias; e.resid = resid; e.resid_dtype = resid_dtype; e.ldr = ldr; e.act = act; e.out = out; e.out_dtype = out_dtype; e.ldc = ldc;
    e.out_mode = out_mode; e.q = q; e.k = k; e.v = v; e.heads = heads; e.head_dim = head_dim; e.tokens = tokens;
    launch_grid(grid, 1, [=] { gemm2w_kernel<false>(A, lda, W, ldw, M, N, K, e); });
}
// the 256 x 256 tile kernel: staging 0..3 (bf16), fp8 = staging 1 with e4m3 operands and row / column scales
extern "C" void emu_gemm256(int staging, int fp8, int direct_epi, const u16* A, long lda, const u16* W, long ldw, int M, int N, int K,
                            const float* bias, const void* resid, int resid_dtype, long ldr, int act, void* out, int out_dtype, long ldc,
                            int out_mode, u16* q, u16* k, u16* v, int heads, int head_dim, int tokens, const float* row_scale,
                            const float* col_scale) {
    Epi e{};
    e.bias = bias; e.resid = resid; e.resid_dtype = resid_dtype; e.ldr = ldr; e.act = act; e.out = out; e.out_dtype = out_dtype; e.ldc = ldc;
    e.out_mode = out_mode; e.q = q; e.k = k; e.v = v; e.heads = heads; e.head_dim = head_dim; e.tokens = tokens;
    e.row_scale = row_scale; e.col_scale = col_scale; e.direct_epi = direct_epi;
    const int tiles = ((M + 255) / 256) * (N / 256);
    if (fp8) { launch_grid(tiles, 1, [=] { gemm256_kernel<1, true>(A, lda, W, ldw, M, N, K, e); }, 512); return; }
    switch (staging) {
        case 0: launch_grid(tiles, 1, [=] { gemm256_kernel<0>(A, lda, W, ldw, M, N, K, e); }, 512); break;
        case 1: launch_grid(tiles, 1, [=] { gemm256_kernel<1>(A, lda, W, ldw, M, N, K, e); }, 512); break;
        case 2: launch_grid(tiles, 1, [=] { gemm256_kernel<2>(A, lda, W, ldw, M, N, K, e); }, 512); break;
        default: launch_grid(tiles, 1, [=] { gemm256_kernel<3>(A, lda, W, ldw, M, N, K, e); }, 512); break;
    }
}
"""


@pytest.fixture(scope="module")
def emu(tmp_path_factory):
    common = open(os.path.join(ROOT, "micro_sam_amd", "csrc", "common.h")).read()
    a0 = common.index("MSAM_DEVINL float gelu_erf(float x)")
    a = common.index("MSAM_DEVINL float relu1(float x)")
    gelu = common[a0:common.index("// two values at once", a0)] + common[a:common.index("// round-to-nearest-even fp32 -> packed fp16", a)]
    w0 = common.index("MSAM_DEVINL float wave_sum_xor16(float v)")
    gelu += common[w0:common.index("MSAM_DEVINL float wave_max64(float v)", w0)]
    text = open(os.path.join(ROOT, "micro_sam_amd", "csrc", "gemm.hip")).read()
    start = text.index("constexpr int BM = 128, BN = 128, BK = 64;")
    end = text.index("// Two-workgroups-per-CU variant of the large-shape kernel")
    end = text.rindex("// ----", start, end)
    header = '#include "%s"\n' % os.path.join(ROOT, "include", "msam_hip.h")
    body = text[start:end]
    # clang-only pieces of the 256 x 256 kernel: its MFMA helpers (the shim's emulation has the same names and layouts) and the
    # dynamic-LDS declaration (a static 160 KB array here)
    a = body.index("typedef float f32x16_t __attribute__((ext_vector_type(16)));")
    b = body.index("// Epilogue straight from the accumulators of the 128 x 64 wave tile")
    body = body[:a] + body[b:]
    body = body.replace("extern __shared__ __attribute__((aligned(16))) uint4 dyn[];", "static uint4 dyn[10240];")
    # the two-workgroups-per-CU kernel: its LDS-DMA instruction is inline assembly and its LDS addresses are address-space casts -
    # here the descriptor is (base, bytes), the DMA an immediate copy into the static LDS image at the same byte offsets, the counted
    # waits nothing (the copy has landed); what runs is the kernel's own ring protocol, swizzle, fragment reads and epilogue
    a2 = text.index("constexpr int GW_BM = 256, GW_BN = 128, GW_BK = 32;")
    b2 = text.index("// Row-complete variant for N == 256 with a fused LayerNorm epilogue", a2)
    b2 = text.rindex("// ----", a2, b2)
    gw = text[a2:b2]
    c = gw.index("typedef unsigned int u32x4_t_gw")
    d = gw.index("template <bool F16>", c)
    gw = gw[:c] + """struct u32x4_t_gw { const char* base; long bytes; };
static uint4 dyn[10240];
static inline void gw_dma16(const u32x4_t_gw& r, int voff, int soff, unsigned lds_addr) {
    uint4 v{0, 0, 0, 0};
    const long o = (long)voff + soff;
    if (o >= 0 && o + 16 <= r.bytes) std::memcpy(&v, r.base + o, 16);
    std::memcpy((char*)dyn + lds_addr + (threadIdx.x & 63) * 16, &v, 16);
}
static inline u32x4_t_gw gw_rsrc(const void* base, long bytes) { return u32x4_t_gw{(const char*)base, bytes}; }
""" + gw[d:]
    gw = gw.replace("extern __shared__ __attribute__((aligned(16))) uint4 dyn[];", "")
    gw = gw.replace("const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)dyn;", "const unsigned lds0 = 0;")
    gw = gw.replace('asm volatile("s_waitcnt vmcnt(6)" ::: "memory")', "(void)0").replace('asm volatile("s_waitcnt vmcnt(0)" ::: "memory")', "(void)0")
    assert "asm volatile" not in gw and "address_space" not in gw
    body = body.replace("static uint4 dyn[10240];", "")            # one LDS image for both kernels (declared in front of gemm2w)
    body = header + gelu + body.replace("uint4 dyn_placeholder;", "")
    # gemm256_kernel refers to dyn before gemm2w's declaration: declare it first
    body = body.replace(header, header + "static uint4 dyn[10240];\n", 1)
    gw = gw.replace("static uint4 dyn[10240];\n", "")
    # the row-complete 64 x 256 kernel with the fused LayerNorm epilogues (decoder token side / up-scaling stage 1)
    a3 = text.index("struct EpiLN {")
    b3 = text.index("thread_local char g_err[512]", a3)
    ln = text[a3:b3].replace("extern __shared__ __attribute__((aligned(16))) uint4 dyn_lds[];", "uint4* dyn_lds = dyn;")
    body = body + gw + ln
    assert "gemm_body" in body and "gemm_kernel" in body and "gemm256_kernel" in body and "extern __shared__" not in body
    lib = build(str(tmp_path_factory.mktemp("emu_gemm")), "gemm", body, ENTRY)
    vp, i, l = ctypes.c_void_p, ctypes.c_int, ctypes.c_long
    lib.emu_gemm.argtypes = [i, vp, l, vp, l, i, i, i, vp, vp, i, i, l, vp, i, i, l, i, vp, i, l, i, vp, vp, vp, i, i, i, i]
    lib.emu_gemm_ln.argtypes = [vp, l, vp, l, i, i, vp, vp, i, vp, vp, ctypes.c_float, i, vp, vp, vp]
    lib.emu_gemm2w.argtypes = [i, vp, l, vp, l, i, i, i, vp, vp, i, l, i, vp, i, l, i, vp, vp, vp, i, i, i]
    lib.emu_gemm256.argtypes = [i, i, i, vp, l, vp, l, i, i, i, vp, vp, i, l, i, vp, i, l, i, vp, vp, vp, i, i, i, vp, vp]
    return lib


MSAM_F32, MSAM_BF16 = 1, 2          # include/msam_hip.h
ACT_GELU, ACT_RELU = 1, 2


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _bf(t):
    return t.to(torch.bfloat16)


def _bits(t):
    return _bf(t).view(torch.int16).numpy().view(np.uint16).copy()


def _from_bits(a):
    return torch.from_numpy(a.view(np.int16)).view(torch.bfloat16).double()


def test_header_constants():
    text = open(os.path.join(ROOT, "include", "msam_hip.h")).read()
    for name, val in (("MSAM_F32", MSAM_F32), ("MSAM_BF16", MSAM_BF16), ("MSAM_ACT_GELU", ACT_GELU), ("MSAM_ACT_RELU", ACT_RELU)):
        assert f"#define {name} {val}" in text or f"{name} = {val}" in text or f"{name}  {val}" in text, name


@pytest.mark.parametrize("glds", [0, 1])
def test_gemm_kernel_source_on_the_cpu(emu, glds):
    g = torch.Generator().manual_seed(21 + glds)
    M, N, K = 200, 256, 192                                                       # 2 x 2 tiles, ragged last row tile, 3 k-tiles
    a, w = _bf(torch.randn(M, K, generator=g)), _bf(torch.randn(N, K, generator=g) / K ** 0.5)
    bias = torch.randn(N, generator=g)
    prod = a.double() @ w.double().t()
    A, W, B = _bits(a), _bits(w), bias.numpy().astype(np.float32).copy()

    def run(**kw):
        d = dict(bias=None, table=None, table_rows=0, table_cols=0, table_ld=0, resid=None, resid_dtype=0, resid_rows=0, ldr=0, act=0,
                 out=None, out_dtype=MSAM_F32, ldc=N, out_mode=0, q=None, k=None, v=None, heads=0, head_dim=0, tokens=0, split_k=0)
        d.update(kw)
        emu.emu_gemm(glds, _ptr(A), K, _ptr(W), K, M, N, K, _ptr(d["bias"]), _ptr(d["table"]), d["table_rows"], d["table_cols"], d["table_ld"],
                     _ptr(d["resid"]), d["resid_dtype"], d["resid_rows"], d["ldr"], d["act"], _ptr(d["out"]), d["out_dtype"], d["ldc"],
                     d["out_mode"], _ptr(d["q"]), _ptr(d["k"]), _ptr(d["v"]), d["heads"], d["head_dim"], d["tokens"], d["split_k"])

    # bias, fp32 output
    out = np.full((M, N), np.nan, np.float32)
    run(bias=B, out=out)
    ref = prod + bias.double()
    assert np.isfinite(out).all() and np.abs(out - ref.numpy()).max() <= 2e-5 * ref.abs().max().item()
    # GELU, bf16 output
    out16 = np.zeros((M, N), np.uint16)
    run(bias=B, act=ACT_GELU, out=out16, out_dtype=MSAM_BF16)
    want = F.gelu(ref)
    assert (_from_bits(out16) - want).abs().max().item() <= 1.2e-2 * want.abs().max().item()
    # row table on the first 128 columns (rows wrap at 7) + fp32 residual with row wrap + ReLU
    table = torch.randn(7, 128, generator=g)
    resid = torch.randn(50, N, generator=g)
    T_, R_ = table.numpy().astype(np.float32).copy(), resid.numpy().astype(np.float32).copy()
    out = np.full((M, N), np.nan, np.float32)
    run(table=T_, table_rows=7, table_cols=128, table_ld=128, resid=R_, resid_dtype=MSAM_F32, resid_rows=50, ldr=N, act=ACT_RELU, out=out)
    rows = torch.arange(M)
    want = prod + resid.double()[rows % 50]
    want[:, :128] += table.double()[rows % 7]
    want = torch.relu(want)
    assert np.abs(out - want.numpy()).max() <= 2e-5 * want.abs().max().item()
    # 16-bit residual, in place on an fp32 output is not offered: bf16 residual into fp32 output
    r16 = _bf(torch.randn(M, N, generator=g))
    R16 = _bits(r16)
    out = np.full((M, N), np.nan, np.float32)
    run(resid=R16, resid_dtype=MSAM_BF16, ldr=N, out=out)
    assert np.abs(out - (prod + r16.double()).numpy()).max() <= 2e-5 * prod.abs().max().item()


def test_gemm_kernel_qkv_split_and_split_k_on_the_cpu(emu):
    g = torch.Generator().manual_seed(5)
    # q / k / v head split: rows = (b, token), columns = (which, head, d)
    Bn, tokens, heads, hd, K = 2, 64, 2, 64, 128
    M, N = Bn * tokens, 3 * heads * hd
    a, w = _bf(torch.randn(M, K, generator=g)), _bf(torch.randn(N, K, generator=g) / K ** 0.5)
    bias = torch.randn(N, generator=g)
    A, W, B = _bits(a), _bits(w), bias.numpy().astype(np.float32).copy()
    q, k, v = (np.zeros((Bn, heads, tokens, hd), np.uint16) for _ in range(3))
    emu.emu_gemm(0, _ptr(A), K, _ptr(W), K, M, N, K, _ptr(B), None, 0, 0, 0, None, 0, 0, 0, 0, None, MSAM_BF16, N, 1, _ptr(q), _ptr(k), _ptr(v),
                 heads, hd, tokens, 0)
    ref = (a.double() @ w.double().t() + bias.double()).reshape(Bn, tokens, 3, heads, hd).permute(2, 0, 3, 1, 4)
    for got, want in zip((q, k, v), ref):
        assert (_from_bits(got) - want).abs().max().item() <= 1.2e-2 * want.abs().max().item()
    # split-K: the 4 slices of the contraction store their fp32 tiles to THEIR part of a workspace [4][M][N] (round 6: no atomics - the launcher
    # adds the parts in slice order, train.hip msam_det_reduce, so that weight gradients are the same bits on every run)
    M, N, K = 128, 128, 1024
    a, w = _bf(torch.randn(M, K, generator=g)), _bf(torch.randn(N, K, generator=g) / K ** 0.5)
    A, W = _bits(a), _bits(w)
    parts = np.full((4, M, N), np.nan, np.float32)
    emu.emu_gemm(0, _ptr(A), K, _ptr(W), K, M, N, K, None, None, 0, 0, 0, None, 0, 0, 0, 0, _ptr(parts), MSAM_F32, N, 0, None, None, None, 0, 0, 0, 4)
    out = ((parts[0] + parts[1]) + parts[2]) + parts[3]
    ref = a.double() @ w.double().t()
    assert np.isfinite(parts).all() and np.abs(out - ref.numpy()).max() <= 3e-5 * ref.abs().max().item()
    for sl in range(4):                                   # every part is the product over its own quarter of K
        want = a[:, sl * 256:(sl + 1) * 256].double() @ w[:, sl * 256:(sl + 1) * 256].double().t()
        assert np.abs(parts[sl] - want.numpy()).max() <= 3e-5 * ref.abs().max().item()


@pytest.mark.parametrize("staging", [3, 0, 1, 2])
def test_gemm256_kernel_source_on_the_cpu(emu, staging):
    """The encoder's 256 x 256 tile kernel (8 waves, 32x32x16 MFMA, transposed product): every operand staging form incl. the
    ping-pong schedule of its two wave groups (staging 3: the groups run one barrier slot apart) on a ragged M, with the LDS-transposed
    epilogue and (staging 3) the epilogue straight from the accumulators: bias + GELU -> bf16, fp32 residual in place."""
    g = torch.Generator().manual_seed(40 + staging)
    M, N, K = 300, 256, 192                                                       # two row tiles (second ragged), 3 k-tiles of 64
    a, w = _bf(torch.randn(M, K, generator=g)), _bf(torch.randn(N, K, generator=g) / K ** 0.5)
    bias = torch.randn(N, generator=g)
    A, W, B = _bits(a), _bits(w), bias.numpy().astype(np.float32).copy()
    ref = a.double() @ w.double().t() + bias.double()
    for direct in ((0, 1) if staging == 3 else (0,)):
        out16 = np.zeros((M, N), np.uint16)
        emu.emu_gemm256(staging, 0, direct, _ptr(A), K, _ptr(W), K, M, N, K, _ptr(B), None, 0, 0, ACT_GELU, _ptr(out16), MSAM_BF16, N, 0,
                        None, None, None, 0, 0, 0, None, None)
        want = F.gelu(ref)
        assert (_from_bits(out16) - want).abs().max().item() <= 1.2e-2 * want.abs().max().item(), direct
        x = torch.randn(M, N, generator=g)
        X = x.numpy().astype(np.float32).copy()
        emu.emu_gemm256(staging, 0, direct, _ptr(A), K, _ptr(W), K, M, N, K, _ptr(B), _ptr(X), MSAM_F32, N, 0, _ptr(X), MSAM_F32, N, 0,
                        None, None, None, 0, 0, 0, None, None)
        assert np.abs(X - (ref + x.double()).numpy()).max() <= 2e-5 * ref.abs().max().item(), direct


def test_gemm256_kernel_qkv_split_and_fp8_on_the_cpu(emu):
    g = torch.Generator().manual_seed(50)
    Bn, tokens, heads, hd, K = 1, 256, 4, 64, 128
    M, N = Bn * tokens, 3 * heads * hd                                            # 256 x 768: three column tiles
    a, w = _bf(torch.randn(M, K, generator=g)), _bf(torch.randn(N, K, generator=g) / K ** 0.5)
    bias = torch.randn(N, generator=g)
    A, W, B = _bits(a), _bits(w), bias.numpy().astype(np.float32).copy()
    q, k, v = (np.zeros((Bn, heads, tokens, hd), np.uint16) for _ in range(3))
    emu.emu_gemm256(3, 0, 0, _ptr(A), K, _ptr(W), K, M, N, K, _ptr(B), None, 0, 0, 0, None, MSAM_BF16, N, 1, _ptr(q), _ptr(k), _ptr(v),
                    heads, hd, tokens, None, None)
    ref = (a.double() @ w.double().t() + bias.double()).reshape(Bn, tokens, 3, heads, hd).permute(2, 0, 3, 1, 4)
    for got, want in zip((q, k, v), ref):
        assert (_from_bits(got) - want).abs().max().item() <= 1.2e-2 * want.abs().max().item()
    # fp8 e4m3 operands (k-tiles of 128 bytes), per-row / per-column scales applied in the epilogue
    M, N, K = 256, 256, 256
    a8 = (torch.randn(M, K, generator=g) * 0.5).to(torch.float8_e4m3fn)
    w8 = (torch.randn(N, K, generator=g) * 0.5).to(torch.float8_e4m3fn)
    rs, cs = torch.rand(M, generator=g) + 0.5, torch.rand(N, generator=g) + 0.5
    A8, W8 = a8.view(torch.uint8).numpy().copy(), w8.view(torch.uint8).numpy().copy()
    RS, CS = rs.numpy().astype(np.float32).copy(), cs.numpy().astype(np.float32).copy()
    out = np.full((M, N), np.nan, np.float32)
    emu.emu_gemm256(1, 1, 0, _ptr(A8), K, _ptr(W8), K, M, N, K, None, None, 0, 0, 0, _ptr(out), MSAM_F32, N, 0, None, None, None, 0, 0, 0,
                    _ptr(RS), _ptr(CS))
    want = (a8.double() @ w8.double().t()) * rs.double()[:, None] * cs.double()[None, :]
    assert np.abs(out - want.numpy()).max() <= 2e-5 * want.abs().max().item()


@pytest.mark.parametrize("K", [64, 96, 320])
def test_gemm2w_kernel_source_on_the_cpu(emu, K):
    """The two-workgroups-per-CU kernel (256 x 128 tiles, LDS-DMA ring of three stages, persistent over tiles): 2 / 3 / 10 k-tiles of 32
    (prologue only, one ring wrap, several), a ragged M, fewer workgroups than tiles (each walks several tiles: the ring is re-used
    across tiles), epilogue from the accumulators with bias + GELU -> bf16 and with the fp32 residual in place."""
    g = torch.Generator().manual_seed(60 + K)
    M, N = 700, 256                                                                # 3 x 2 tiles, last row tile ragged
    a, w = _bf(torch.randn(M, K, generator=g)), _bf(torch.randn(N, K, generator=g) / K ** 0.5)
    bias = torch.randn(N, generator=g)
    A, W, B = _bits(a), _bits(w), bias.numpy().astype(np.float32).copy()
    ref = a.double() @ w.double().t() + bias.double()
    out16 = np.zeros((M, N), np.uint16)
    emu.emu_gemm2w(4, _ptr(A), K, _ptr(W), K, M, N, K, _ptr(B), None, 0, 0, ACT_GELU, _ptr(out16), MSAM_BF16, N, 0, None, None, None, 0, 0, 0)
    want = F.gelu(ref)
    assert (_from_bits(out16) - want).abs().max().item() <= 1.2e-2 * want.abs().max().item()
    x = torch.randn(M, N, generator=g)
    X = x.numpy().astype(np.float32).copy()
    emu.emu_gemm2w(6, _ptr(A), K, _ptr(W), K, M, N, K, _ptr(B), _ptr(X), MSAM_F32, N, 0, _ptr(X), MSAM_F32, N, 0, None, None, None, 0, 0, 0)
    assert np.abs(X - (ref + x.double()).numpy()).max() <= 2e-5 * ref.abs().max().item()


@pytest.mark.parametrize("mode", [1, 2])
def test_gemm_ln_kernel_source_on_the_cpu(emu, mode):
    """Row-complete product (N = 256) with the LayerNorm in the epilogue: mode 1 = LayerNorm over the row (+ the 16-bit operand copies
    out_a = round16(v + add), out_b = round16(v) for the next products), mode 2 = LayerNorm over each 64-column group + GELU."""
    g = torch.Generator().manual_seed(70 + mode)
    M, K = 150, 128                                                               # ragged last 64-row tile
    a, w = _bf(torch.randn(M, K, generator=g)), _bf(torch.randn(256, K, generator=g) / K ** 0.5)
    bias = torch.randn(256, generator=g)
    n = 256 if mode == 1 else 64
    lw, lb = torch.randn(n, generator=g) * 0.3 + 1, torch.randn(n, generator=g) * 0.3
    add = torch.randn(M, 256, generator=g)
    A, W, B = _bits(a), _bits(w), bias.numpy().astype(np.float32).copy()
    LW, LB, ADD = (t.numpy().astype(np.float32).copy() for t in (lw, lb, add))
    out = np.full((M, 256), np.nan, np.float32)
    oa, ob = np.zeros((M, 256), np.uint16), np.zeros((M, 256), np.uint16)
    emu.emu_gemm_ln(_ptr(A), K, _ptr(W), K, M, K, _ptr(B), _ptr(out), MSAM_F32, _ptr(LW), _ptr(LB), 1e-5, mode,
                    _ptr(ADD) if mode == 1 else None, _ptr(oa) if mode == 1 else None, _ptr(ob) if mode == 1 else None)
    y = a.double() @ w.double().t() + bias.double()
    if mode == 1:
        want = F.layer_norm(y, (256,), lw.double(), lb.double(), 1e-5)
    else:
        want = F.gelu(F.layer_norm(y.reshape(M, 4, 64), (64,), lw.double(), lb.double(), 1e-5)).reshape(M, 256)
    assert np.isfinite(out).all() and np.abs(out - want.numpy()).max() <= 2e-4 * want.abs().max().item()
    if mode == 1:
        assert (_from_bits(ob) - want).abs().max().item() <= 1e-2 * want.abs().max().item()
        assert (_from_bits(oa) - (want + add.double())).abs().max().item() <= 1e-2 * (want + add.double()).abs().max().item()
