"""tools/uf_lab.py's candidate variants of up_fused_kernel (text patches of csrc/upfused.hip, measured on the GPU by that tool) on the CPU:
every patch still applies to the shipped source, and the variants that claim to compute the same function do - `exact` ones bit for bit,
`close` ones within the decoder's tolerance - when the patched file is compiled for the host (tests/hip_host_shim.py) and driven through
its own C entry point.  The timing-only variants (results wrong by construction) are only checked to apply."""
import ctypes
import importlib.util
import os

import numpy as np
import pytest
import torch

from hip_host_shim import ERROR_STUBS, FILE_PATCHES, build_file

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_spec = importlib.util.spec_from_file_location("uf_lab", os.path.join(ROOT, "tools", "uf_lab.py"))
lab = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(lab)


def test_every_variant_applies_to_the_shipped_source():
    base = lab.variant_source("base")
    assert base == open(lab.SRC).read()
    for name, v in lab.V.items():
        src = lab.variant_source(name)                     # asserts that each patched text occurs exactly once
        assert (src == base) == (not v["patches"]), name
        assert v["kind"] in ("base", "exact", "close", "timing")


def _host_lib(tmpdir, name):
    src = lab.variant_source(name)
    for start, end, text in FILE_PATCHES["upfused.hip"]:
        a = src.index(start)
        src = src[:a] + text + src[src.index(end, a):]
    lib = build_file(str(tmpdir), ROOT, "upfused.hip", source=src, extra=ERROR_STUBS)
    vp, i32 = ctypes.c_void_p, ctypes.c_int32
    lib.msam_upscale_fused_layout.restype = i32
    lib.msam_upscale_fused_layout.argtypes = [vp, i32, i32, vp, vp, vp, vp, ctypes.c_float, vp, vp, vp, i32, i32, i32, vp, vp]
    return lib


def _bits(t):
    return t.to(torch.float16).contiguous().view(torch.int16).numpy().view(np.uint16).copy()


def _run(lib, arrs, P, nmask):
    out = np.full((P, nmask, 256, 256), np.nan, np.float32)
    p = [a.ctypes.data_as(ctypes.c_void_p) for a in arrs]
    rc = lib.msam_upscale_fused_layout(p[0], 1, P, p[1], p[2], p[3], p[4], 1e-6, p[5], p[6], p[7], 128, 1, nmask, out.ctypes.data_as(ctypes.c_void_p), None)
    assert rc == 0
    return out


def _inputs(P, seed):
    g = torch.Generator().manual_seed(seed)
    keys = torch.randn(P, 4096, 256, generator=g)
    keys[:, :, :64] += 1.5                                   # channel means away from zero: the one-pass variance has something to cancel
    return [_bits(keys), _bits(torch.randn(256, 256, generator=g) / 16), (torch.randn(256, generator=g) * 0.5 + 1.0).numpy().copy(),
            (torch.randn(64, generator=g) * 0.2 + 1).numpy().copy(), (torch.randn(64, generator=g) * 0.3).numpy().copy(),
            _bits(torch.randn(128, 64, generator=g) / 8), torch.randn(32, generator=g).numpy().copy(), torch.randn(P, 4, 128, generator=g).numpy().copy()]


@pytest.fixture(scope="module")
def case(tmp_path_factory):
    os.environ["MSAM_EMU_CUS"] = "2"                         # one prompt as four quarter-prompt items over four workgroups
    arrs, P = _inputs(1, 11), 1
    base = _run(_host_lib(tmp_path_factory.mktemp("uf_base"), "base"), arrs, P, 3)
    assert np.isfinite(base).all()
    yield arrs, P, base
    os.environ.pop("MSAM_EMU_CUS", None)


@pytest.mark.parametrize("name", [n for n, v in lab.V.items() if v["kind"] in ("exact", "close") and v.get("host_checked", True)])
def test_candidate_variants_compute_the_shipped_function(case, tmp_path, name):
    arrs, P, base = case
    out = _run(_host_lib(tmp_path, name), arrs, P, 3)
    if lab.V[name]["kind"] == "exact":
        assert np.array_equal(out, base), float(np.abs(out - base).max())
    else:
        scale = float(np.abs(base).max())
        err = np.abs(out - base)
        assert err.max() <= 2e-3 * scale and err.mean() <= 1e-4 * scale, (err.max() / scale, err.mean() / scale)


def test_pipelined_loop_across_prompts_of_one_workgroup(tmp_path):
    """Three whole prompts over two workgroups: the first one runs prompts 0 and 2 back to back (512 tiles: the prefetched stage 1 crosses the
    prompt boundary, the hyper weights change under it), one and two masks - against each prompt launched alone."""
    os.environ["MSAM_EMU_CUS"] = "1"
    try:
        arrs, P = _inputs(3, 12), 3
        lib = _host_lib(tmp_path / "b", "base")
        for nmask in (1, 2):
            together = _run(lib, arrs, P, nmask)
            for p in range(P):
                alone = _run(lib, [np.ascontiguousarray(arrs[0][p:p + 1])] + arrs[1:7] + [np.ascontiguousarray(arrs[7][p:p + 1])], 1, nmask)
                assert np.array_equal(together[p:p + 1], alone), p
    finally:
        os.environ.pop("MSAM_EMU_CUS", None)


def test_the_host_build_notices_a_missing_barrier(tmp_path):
    """Negative control for the checks above: the waves of a workgroup are concurrent host threads, so the kernel WITHOUT its per-tile
    barrier (staging buffers and output patches reused while other waves still read them) does not reproduce the shipped kernel's output."""
    os.environ["MSAM_EMU_CUS"] = "1"
    try:
        arrs, P = _inputs(2, 5), 2
        base = _run(_host_lib(tmp_path / "b", "base"), arrs, P, 3)
        out = _run(_host_lib(tmp_path / "n", "T_no_barrier"), arrs, P, 3)
        assert (out != base).mean() > 0.05
    finally:
        os.environ.pop("MSAM_EMU_CUS", None)


def test_fp16_low_res_output_is_the_rounded_fp32_output(tmp_path):
    """msam_upscale_fused_out(low_res_dtype = MSAM_F16) (round 4: the AMG path's hand-over to msam_postprocess_masks16) writes exactly the
    fp16 rounding of what the fp32 form writes."""
    os.environ["MSAM_EMU_CUS"] = "2"
    try:
        arrs, P = _inputs(1, 21), 1
        lib = _host_lib(tmp_path, "base")
        vp, i32 = ctypes.c_void_p, ctypes.c_int32
        lib.msam_upscale_fused_out.restype = i32
        lib.msam_upscale_fused_out.argtypes = [vp, i32, i32, vp, vp, vp, vp, ctypes.c_float, vp, vp, vp, i32, i32, i32, vp, i32, vp]
        ref = _run(lib, arrs, P, 3)
        out16 = np.zeros((P, 3, 256, 256), np.float16)
        p = [a.ctypes.data_as(vp) for a in arrs]
        assert lib.msam_upscale_fused_out(p[0], 1, P, p[1], p[2], p[3], p[4], 1e-6, p[5], p[6], p[7], 128, 1, 3, out16.ctypes.data_as(vp), 4, None) == 0
        assert np.array_equal(out16.view(np.uint16), ref.astype(np.float16).view(np.uint16))
        assert lib.msam_upscale_fused_out(p[0], 1, P, p[1], p[2], p[3], p[4], 1e-6, p[5], p[6], p[7], 128, 1, 3, out16.ctypes.data_as(vp), 2, None) == 1
    finally:
        os.environ.pop("MSAM_EMU_CUS", None)
