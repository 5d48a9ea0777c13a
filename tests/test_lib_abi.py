"""CPU-side checks of the C-ABI library: it builds, loads and exports every symbol include/msam_hip.h declares."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_builds_and_exports_header_symbols():
    from micro_sam_amd import _lib, build
    path = build.build(verbose=False)
    assert os.path.exists(path)
    lib = _lib.load()
    assert lib.msam_abi_version() == 1
    header = open(os.path.join(ROOT, "include", "msam_hip.h")).read()
    declared = set(re.findall(r"\b(msam_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in msam_hip.h but not exported"
    for name in _lib.exported_symbols():
        assert name in declared, f"{name} bound in _lib.py but not declared in msam_hip.h"


def test_argument_errors_raise_without_gpu():
    """Validation happens before any launch, so these run on the CPU-only container."""
    import ctypes as C
    from micro_sam_amd import _lib
    lib = _lib.load()
    p = _lib.GemmParams()
    assert lib.msam_gemm_bf16(C.byref(p), None) == 1
    with pytest.raises(ValueError):
        _lib.check(1, "gemm")
    assert b"null operand" in lib.msam_last_error()


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "micro_sam_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "import oracle" not in src and "from oracle" not in src, fn


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from micro_sam_amd import util
    from micro_sam_amd.synthetic import synthetic_state_dict
    with pytest.raises(RuntimeError):
        util.get_sam_model("vit_b", state_dict={})
