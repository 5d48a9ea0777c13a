"""Test helper: run the PRODUCT's Python layer against the host-compiled library (tests/hip_host_shim.build_library) - the library
handle, require_gpu and the stream accessors of micro_sam_amd._lib and torch.cuda.current_stream are patched inside a context manager
and restored afterwards.  TEST INFRASTRUCTURE: the product itself has no CPU path."""
import contextlib
import os

import torch

from hip_host_shim import build_library

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@contextlib.contextmanager
def product_on_host(tmpdir: str, emu_cus: int = 4):
    from micro_sam_amd import _lib
    os.environ["MSAM_EMU_CUS"] = str(emu_cus)
    host = build_library(tmpdir, ROOT)
    for name, (res, args) in _lib._PROTOS.items():
        fn = getattr(host, name)
        fn.restype, fn.argtypes = res, args
    saved = (_lib._lib, _lib.require_gpu, _lib.stream_ptr, _lib.ptr, torch.cuda.current_stream)

    class _Stream:
        cuda_stream = 0
    _lib._lib = host
    _lib.require_gpu = lambda device=None: torch.device("cpu") if device is None else torch.device(device)
    _lib.stream_ptr = lambda: None
    _lib.ptr = lambda t: None if t is None else t.data_ptr()
    torch.cuda.current_stream = lambda *a, **k: _Stream()
    try:
        yield host
    finally:
        _lib._lib, _lib.require_gpu, _lib.stream_ptr, _lib.ptr, torch.cuda.current_stream = saved
        os.environ.pop("MSAM_EMU_CUS", None)
