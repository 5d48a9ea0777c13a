"""The WHOLE library compiled for the host (tests/hip_host_shim.build_library: every csrc/*.hip file behind a generated common.h and a
minimal HIP runtime; kernels on host threads, one per lane) and driven through its C ABI (include/msam_hip.h) exactly as
micro_sam_amd/_lib.py drives the GPU build: argument checks, derived launch parameters, kernels, epilogues.  Integer / byte stages are
compared bit for bit with the oracle, Pillow or numpy; floating-point products with fp64.  TEST INFRASTRUCTURE: the product never loads
this build (there is no CPU fallback), and the GPU suite runs the same comparisons on the device."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from hip_host_shim import build_library

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F32, BF16 = 1, 2                      # include/msam_hip.h
vp = C.c_void_p


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    return build_library(str(tmp_path_factory.mktemp("host_lib")), ROOT)


def _p(a):
    return None if a is None else a.ctypes.data_as(vp)


def _err(lib):
    lib.msam_last_error.restype = C.c_char_p
    return lib.msam_last_error().decode()


def test_every_prototype_of_the_header_is_exported(lib):
    names = sorted(set(re.findall(r"\b(msam_[a-z0-9_]+)\s*\(", open(os.path.join(ROOT, "include", "msam_hip.h")).read())))
    assert len(names) > 80 and not [n for n in names if not hasattr(lib, n)]


def _blobs(rng, n, H, W):
    yy, xx = np.mgrid[0:H, 0:W]
    masks = np.zeros((n, H, W), bool)
    for i in range(n):
        cy, cx = rng.uniform(0, H), rng.uniform(0, W)
        ry, rx = rng.uniform(3, H / 5), rng.uniform(3, W / 5)
        masks[i] = ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 < 1
        if i % 5 == 4:                                       # a second piece: one mask, two components
            masks[i] |= ((yy - (cy + 2.5 * ry) % H) / (ry / 2)) ** 2 + ((xx - cx) / (rx / 2)) ** 2 < 1
    return masks


def _pack(masks):
    n, H, W = masks.shape
    pad = (-H) % 32
    m = np.pad(masks, ((0, 0), (0, pad), (0, 0))).reshape(n, (H + pad) // 32, 32, W).astype(np.uint32)
    return (m << np.arange(32, dtype=np.uint32)[None, None, :, None]).sum(2).astype(np.uint32)


@pytest.mark.parametrize("H,W,n,seed", [(96, 128, 60, 0), (530, 150, 40, 1), (150, 530, 30, 2)])
def test_amg_generate_labels_is_the_oracles_generate(lib, H, W, n, seed):
    """generate(output_mode="instance_segmentation") as ONE call (filters, box NMS, paint, connected components in the reference's
    block-major numbering, size filter, consecutive relabel: 15 kernels) against oracle/pipeline_ref.amg_generate on the same
    candidates - identical label images, ids included; the other cases span two 512-blocks vertically / horizontally."""
    from oracle import amg_ref as A
    from oracle import pipeline_ref as PR
    rng = np.random.default_rng(seed)
    masks = _blobs(rng, n, H, W)
    masks[3] = False                                          # an empty mask
    iou = rng.uniform(0.8, 1.0, n).astype(np.float32)
    stab = rng.uniform(0.9, 1.0, n).astype(np.float32)
    mt = torch.from_numpy(masks)
    boxes = A.batched_mask_to_box(mt).numpy().astype(np.int32)
    area = masks.reshape(n, -1).sum(1).astype(np.int32)
    bits = _pack(masks)
    lib.msam_amg_generate_workspace_bytes.restype = C.c_int64
    need = lib.msam_amg_generate_workspace_bytes(n, H, W)
    assert need > 0
    ws = np.zeros(need, np.uint8)
    labels = np.full((H, W), -7, np.int32)
    flag = np.full(1, -1, np.int32)
    crop = (C.c_int32 * 4)(0, 0, W, H)
    rc = lib.msam_amg_generate_labels(_p(iou), _p(stab), _p(boxes), _p(area), _p(bits), n, H, W, crop, C.c_float(0.88), C.c_float(0.95),
                                      C.c_float(0.7), 0, 1, _p(labels), _p(flag), _p(ws), C.c_int64(need), None)
    assert rc == 0, _err(lib)
    assert flag[0] == 0
    data = A.MaskData(iou_preds=torch.from_numpy(iou), stability_score=torch.from_numpy(stab), boxes=torch.from_numpy(boxes).long(),
                      rles=A.mask_to_rle(mt), points=torch.zeros(n, 2))
    state = {"crop_list": [data], "crop_boxes": [[0, 0, W, H]], "original_size": (H, W)}
    seg = PR.amg_generate(state, with_background=True)
    assert np.array_equal(labels.astype(np.int64), np.asarray(seg).astype(np.int64))
    assert labels.max() >= 3


def test_resample_u8_is_pillow(lib):
    """ResizeLongestSide.apply_image: Pillow's fixed-point BILINEAR resize, horizontal pass then vertical pass, bit for bit."""
    from PIL import Image

    from micro_sam_amd.transforms import pil_bilinear_tables
    rng = np.random.default_rng(2)
    for (H, W, nh, nw) in [(70, 90, 113, 145), (120, 64, 45, 24), (33, 200, 33, 97)]:
        img = rng.integers(0, 256, (2, H, W, 3), dtype=np.uint8)
        x, h, w = img, H, W
        if nw != w:
            b, c = pil_bilinear_tables(w, nw)
            b, c = np.ascontiguousarray(b, np.int32), np.ascontiguousarray(c, np.int32)
            out = np.zeros((2, h, nw, 3), np.uint8)
            assert lib.msam_resample_u8(_p(x), 2, h, w, 3, 1, nw, _p(b), _p(c), int(c.shape[1]), _p(out), None) == 0, _err(lib)
            x, w = out, nw
        if nh != h:
            b, c = pil_bilinear_tables(h, nh)
            b, c = np.ascontiguousarray(b, np.int32), np.ascontiguousarray(c, np.int32)
            out = np.zeros((2, nh, w, 3), np.uint8)
            assert lib.msam_resample_u8(_p(x), 2, h, w, 3, 0, nh, _p(b), _p(c), int(c.shape[1]), _p(out), None) == 0, _err(lib)
            x = out
        for k in range(2):
            assert np.array_equal(x[k], np.array(Image.fromarray(img[k]).resize((nw, nh), Image.BILINEAR)))


def test_to_image_is_the_host_formula(lib):
    from micro_sam_amd import util
    rng = np.random.default_rng(3)
    for arr, dt in ((rng.integers(3, 200, (40, 50), dtype=np.uint8), 5), ((rng.standard_normal((31, 47, 2)) * 100).astype(np.float32), F32)):
        text = open(os.path.join(ROOT, "include", "msam_hip.h")).read()
        u8 = int(re.search(r"#define MSAM_U8 (\d+)", text).group(1))
        a = np.ascontiguousarray(arr if arr.ndim == 3 else arr[..., None])
        H, W, Cc = a.shape
        out = np.zeros((H, W, 3), np.uint8)
        ws = np.zeros(8, np.int32)
        assert lib.msam_to_image(_p(a), u8 if arr.dtype == np.uint8 else F32, H, W, Cc, _p(out), _p(ws), None) == 0, _err(lib)
        assert np.array_equal(out, util._to_image(arr))


def test_slice_overlaps_is_the_contingency_table(lib):
    rng = np.random.default_rng(4)
    Z, H, W = 4, 40, 60
    vol = np.zeros((Z, H, W), np.int32)
    nxt = 1
    for z in range(Z):
        for _ in range(6):
            y, x, h, w = rng.integers(0, H - 8), rng.integers(0, W - 8), rng.integers(4, 16), rng.integers(4, 16)
            vol[z, y:y + h, x:x + w] = nxt
            nxt += 1
    cap, max_edges = 1 << 12, 1 << 10
    keys, counts = np.zeros(cap, np.uint64), np.zeros(cap, np.int32)
    edges, n = np.zeros((max_edges, 3), np.int32), np.zeros(2, np.int32)
    assert lib.msam_slice_overlaps(_p(vol), Z, H, W, _p(keys), _p(counts), cap, _p(edges), max_edges, _p(n), None) == 0, _err(lib)
    assert n[1] == 0
    got = {(int(a), int(b)): int(c) for a, b, c in edges[: n[0]]}
    want = {}
    for z in range(Z - 1):
        a, b = vol[z].ravel(), vol[z + 1].ravel()
        for u, v in zip(a[a > 0], b[a > 0]):
            want[(int(u), int(v))] = want.get((int(u), int(v)), 0) + 1
    assert got == want


def test_cast_transpose_and_layernorm_through_the_abi(lib):
    rng = np.random.default_rng(5)
    M, K = 130, 68
    x = (rng.standard_normal((M, K)) * 3).astype(np.float32)
    o16, oT, cs = np.zeros((M, K), np.uint16), np.zeros((K, M), np.uint16), np.zeros(K, np.float32)
    assert lib.msam_cast_transpose(_p(x), F32, C.c_int64(M), K, C.c_int64(K), _p(o16), _p(oT), _p(cs), None) == 0, _err(lib)
    want = torch.from_numpy(x).to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)
    assert np.array_equal(o16, want) and np.array_equal(oT, want.T) and np.abs(cs - x.sum(0)).max() <= 1e-4 * np.abs(x).sum(0).max()
    assert lib.msam_cast_transpose(_p(x), F32, C.c_int64(M), 66, C.c_int64(K), _p(o16), None, None, None) == 1      # K % 4 != 0: refused
    rows, dim = 37, 256
    xs = (rng.standard_normal((rows, dim)) * 2 + 0.5).astype(np.float32)
    w, b = (rng.standard_normal(dim) * 0.2 + 1).astype(np.float32), rng.standard_normal(dim).astype(np.float32)
    out = np.zeros((rows, dim), np.float32)
    assert lib.msam_layernorm(_p(xs), _p(w), _p(b), C.c_float(1e-6), C.c_int64(rows), dim, _p(out), F32, 0, 0, None) == 0, _err(lib)
    ref = torch.nn.functional.layer_norm(torch.from_numpy(xs).double(), (dim,), torch.from_numpy(w).double(), torch.from_numpy(b).double(), 1e-6)
    assert np.abs(out - ref.numpy()).max() <= 2e-5


def test_gemm_through_the_abi(lib):
    """msam_gemm_bf16 with the msam_gemm_t of micro_sam_amd/_lib.py: the dispatcher's checks and its choice of kernel (here the
    128 x 128 tile kernel, the row-complete LayerNorm kernel and the split-K launch) in front of the kernels."""
    from micro_sam_amd import _lib as L
    g = torch.Generator().manual_seed(8)
    M, N, K = 150, 256, 128
    a = torch.randn(M, K, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(N, generator=g)
    A_, W_ = (t.view(torch.int16).numpy().view(np.uint16).copy() for t in (a, w))
    B_ = bias.numpy().copy()
    y = a.double() @ w.double().t() + bias.double()

    def params(**kw):
        p = L.GemmParams()
        base = dict(A=A_.ctypes.data, W=W_.ctypes.data, M=M, N=N, K=K, lda=K, ldw=K, ldc=N, bias=B_.ctypes.data, out_dtype=F32)
        base.update(kw)
        for k, v in base.items():
            setattr(p, k, v)
        return p
    out = np.full((M, N), np.nan, np.float32)
    assert lib.msam_gemm_bf16(C.byref(params(out=out.ctypes.data)), None) == 0, _err(lib)
    assert np.abs(out - y.numpy()).max() <= 2e-5 * y.abs().max().item()
    lw, lb = (torch.randn(256, generator=g) * 0.3 + 1), torch.randn(256, generator=g) * 0.3
    LW, LB = lw.numpy().copy(), lb.numpy().copy()
    out = np.full((M, N), np.nan, np.float32)
    assert lib.msam_gemm_bf16(C.byref(params(out=out.ctypes.data, ln_mode=1, ln_w=LW.ctypes.data, ln_b=LB.ctypes.data, ln_eps=1e-5)), None) == 0, _err(lib)
    ref = torch.nn.functional.layer_norm(y, (256,), lw.double(), lb.double(), 1e-5)
    assert np.abs(out - ref.numpy()).max() <= 2e-4 * ref.abs().max().item()
    assert lib.msam_gemm_bf16(C.byref(params(out=out.ctypes.data, N=200)), None) == 1 and "128" in _err(lib)        # N % 128 != 0
    # split-K (training: dW): 128 x 128 output over K = 512 in 4 slices, no bias
    K2 = 512
    a2, w2 = torch.randn(128, K2, generator=g).to(torch.bfloat16), (torch.randn(128, K2, generator=g) / K2 ** 0.5).to(torch.bfloat16)
    A2, W2 = (t.view(torch.int16).numpy().view(np.uint16).copy() for t in (a2, w2))
    out2 = np.full((128, 128), np.nan, np.float32)
    p = params(A=A2.ctypes.data, W=W2.ctypes.data, M=128, N=128, K=K2, lda=K2, ldw=K2, ldc=128, bias=None, out=out2.ctypes.data, split_k=4)
    assert lib.msam_gemm_bf16(C.byref(p), None) == 0, _err(lib)
    ref2 = a2.double() @ w2.double().t()
    assert np.abs(out2 - ref2.numpy()).max() <= 3e-5 * ref2.abs().max().item()


@pytest.mark.parametrize("case", ["single", "all_rejected", "ragged_size", "no_background", "nested"])
def test_amg_generate_labels_edge_cases(lib, case):
    """One candidate; every candidate rejected by the thresholds; an image whose sides are not multiples of 32; with_background=False
    (the largest component is NOT set to zero); nested / overlapping masks painted by descending area - each against the oracle."""
    from oracle import amg_ref as A
    from oracle import pipeline_ref as PR
    rng = np.random.default_rng({"single": 1, "all_rejected": 2, "ragged_size": 3, "no_background": 4, "nested": 5}[case])
    H, W, n = (75, 101, 20) if case == "ragged_size" else (64, 96, 1 if case == "single" else 24)
    masks = _blobs(rng, n, H, W)
    if case == "nested":
        yy, xx = np.mgrid[0:H, 0:W]
        for i in range(6):                                     # concentric discs: smaller ones end up on top
            masks[i] = (yy - 32) ** 2 + (xx - 48) ** 2 < (30 - 4 * i) ** 2
    iou = rng.uniform(0.9, 1.0, n).astype(np.float32)
    stab = rng.uniform(0.96, 1.0, n).astype(np.float32)
    if case == "all_rejected":
        iou[:] = 0.5
    mt = torch.from_numpy(masks)
    boxes = A.batched_mask_to_box(mt).numpy().astype(np.int32)
    area = masks.reshape(n, -1).sum(1).astype(np.int32)
    bits = _pack(masks)
    lib.msam_amg_generate_workspace_bytes.restype = C.c_int64
    need = lib.msam_amg_generate_workspace_bytes(n, H, W)
    ws = np.zeros(need, np.uint8)
    labels = np.full((H, W), -7, np.int32)
    flag = np.full(1, -1, np.int32)
    crop = (C.c_int32 * 4)(0, 0, W, H)
    wb = 0 if case == "no_background" else 1
    rc = lib.msam_amg_generate_labels(_p(iou), _p(stab), _p(boxes), _p(area), _p(bits), n, H, W, crop, C.c_float(0.88), C.c_float(0.95),
                                      C.c_float(0.7), 0, wb, _p(labels), _p(flag), _p(ws), C.c_int64(need), None)
    assert rc == 0 and flag[0] == 0, _err(lib)
    data = A.MaskData(iou_preds=torch.from_numpy(iou), stability_score=torch.from_numpy(stab), boxes=torch.from_numpy(boxes).long(),
                      rles=A.mask_to_rle(mt), points=torch.zeros(n, 2))
    state = {"crop_list": [data], "crop_boxes": [[0, 0, W, H]], "original_size": (H, W)}
    seg = np.asarray(PR.amg_generate(state, with_background=bool(wb)))
    assert np.array_equal(labels.astype(np.int64), seg.astype(np.int64)), case
