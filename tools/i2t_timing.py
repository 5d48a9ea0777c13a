"""GPU-box experiment: where does a fold_i2t tile iteration spend its cycles?  Runs msam_i2t_fold_layer (P = 1024 prompts, the
AMG shape) through the instrumented instantiation (csrc/decfold.hip, I2T_STAMP) and prints the mean shader-clock distance
between the phase stamps of workgroup 0 / wave 0 over tiles 8..63, for layer-0 (shared source) and layer-1 (per-prompt stream).
    python tools/i2t_timing.py"""
import ctypes as C
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from micro_sam_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda", 0)
lib = _lib.load()
g = torch.Generator().manual_seed(1)
P, Nt, T = 1024, 7, 4096
bf = lambda t: t.to(torch.bfloat16)
ktok = bf(torch.randn(P, Nt, 128, generator=g)).to(dev)
vtok = bf(torch.randn(P, Nt, 128, generator=g)).to(dev)
wq = bf(torch.randn(128, 256, generator=g) / 16).to(dev)
tabq = bf(torch.randn(T, 128, generator=g)).to(dev)
wo = bf(torch.randn(256, 128, generator=g) / math.sqrt(128)).to(dev)
bo = torch.randn(256, generator=g).to(dev)
lw = (torch.randn(256, generator=g) * 0.2 + 1).to(dev)
lb = torch.randn(256, generator=g).to(dev)
NAMES = ["loads issued + item setup + scores (16 ds_read, 18 MFMA)", "softmax + P^T to LDS", "barrier A", "O^T (24+8 MFMA)",
         "LN partial sums + shuffles + red", "barrier B", "normalise + in-place write", "stage next tile (ds_write)", "barrier C",
         "copy-out (ds_read + buffer_store)"]
for shared in (True, False):
    x = bf(torch.randn(1 if shared else P, T, 256, generator=g)).to(dev)
    out = torch.empty((P, T, 256), dtype=torch.bfloat16, device=dev)
    for timing in (0, 1):
        lib.msam_debug_i2t_timing(timing, None)
        ops.i2t_fold_layer(x, ktok, vtok, wq, tabq, wo, bo, lw, lb, x_shared=shared, out=out)       # warm
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        ops.i2t_fold_layer(x, ktok, vtok, wq, tabq, wo, bo, lw, lb, x_shared=shared, out=out)
        b.record()
        torch.cuda.synchronize()
        print(f"x_shared={shared} timing={timing}: {a.elapsed_time(b):.3f} ms", flush=True)
    buf = (C.c_uint64 * (64 * 12))()
    lib.msam_debug_i2t_timing(0, buf)
    st = np.frombuffer(buf, dtype=np.uint64).reshape(64, 12).astype(np.int64)
    d = np.diff(st[8:, :11], axis=1)                          # phases inside an iteration
    it = np.diff(st[8:, 0])                                   # start-to-start
    print(f"  iteration start-to-start: mean {it.mean():.0f} cycles (min {it.min()}, max {it.max()})")
    for k, name in enumerate(NAMES):
        print(f"  {name:62s} {d[:, k].mean():8.0f}  (min {d[:, k].min():6d} max {d[:, k].max():6d})")
