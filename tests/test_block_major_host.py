"""csrc/common.h bm_key / bm_pix (the block-major order that numbers connected components the way the reference's
elf.parallel.label(block_shape=(512, 512)) does, micro_sam/util.py:1834-1838): the two device functions are plain integer arithmetic, so
their SOURCE TEXT is compiled as host C++ here and compared with the oracle's key map - a bijection, inverse of each other, equal to
oracle.amg_ref.block_major_keys on aligned and ragged shapes."""
import ctypes
import os
import re
import subprocess
import tempfile

import numpy as np

from oracle import amg_ref as A

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build():
    src = open(os.path.join(ROOT, "micro_sam_amd", "csrc", "common.h")).read()
    fns = re.findall(r"MSAM_DEVINL int bm_(?:key|pix)\(.*?\n}\n", src, re.S)
    assert len(fns) == 2
    code = "#include <algorithm>\nusing std::min;\n#define MSAM_DEVINL static inline\n" + "".join(fns) + (
        'extern "C" void keys(int H, int W, int* k, int* p) { for (int i = 0; i < H * W; ++i) { k[i] = bm_key(i, H, W); '
        "p[i] = bm_pix(i, H, W); } }\n")
    d = tempfile.mkdtemp()
    with open(os.path.join(d, "bm.cpp"), "w") as f:
        f.write(code)
    so = os.path.join(d, "bm.so")
    subprocess.check_call(["g++", "-O1", "-shared", "-fPIC", os.path.join(d, "bm.cpp"), "-o", so])
    return ctypes.CDLL(so)


def test_block_major_key_functions():
    lib = _build()
    for h, w in [(1024, 1024), (512, 512), (300, 200), (700, 1100), (1536, 1536), (513, 1025), (2048, 520), (33, 17)]:
        n = h * w
        k = np.zeros(n, dtype=np.int32); p = np.zeros(n, dtype=np.int32)
        lib.keys(h, w, k.ctypes.data_as(ctypes.c_void_p), p.ctypes.data_as(ctypes.c_void_p))
        ref = A.block_major_keys(h, w).reshape(-1)
        assert np.array_equal(k, ref), (h, w)
        assert np.array_equal(p[k], np.arange(n)) and np.array_equal(k[p], np.arange(n)), (h, w)
        if h <= 512 and w <= 512:
            assert np.array_equal(k, np.arange(n))
