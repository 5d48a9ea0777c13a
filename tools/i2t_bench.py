"""GPU-box A/B of the image->token layer kernels at the AMG shape (P = 1024 prompts, Nt = 7): the token-owner kernel
(csrc/decfold_tok.hip, i2t_variant 1) against the 4-wave tile kernel (csrc/decfold.hip, i2t_variant 0), layer-0 form (shared
source) and layer-1 form (per-prompt stream, in place):  python tools/i2t_bench.py"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from micro_sam_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda", 0)
lib = _lib.load()
dt = _lib.decoder_dtype()
g = torch.Generator().manual_seed(1)
P, Nt, T = 1024, 7, 4096
d = lambda t: t.to(dt)
ktok = d(torch.randn(P, Nt, 128, generator=g)).to(dev)
vtok = d(torch.randn(P, Nt, 128, generator=g)).to(dev)
wq = d(torch.randn(128, 256, generator=g) / 16).to(dev)
tabq = d(torch.randn(T, 128, generator=g)).to(dev)
wo = d(torch.randn(256, 128, generator=g) / math.sqrt(128)).to(dev)
bo = torch.randn(256, generator=g).to(dev)
lw = (torch.randn(256, generator=g) * 0.2 + 1).to(dev)
lb = torch.randn(256, generator=g).to(dev)


def run(x, shared, out, reps=5):
    ops.i2t_fold_layer(x, ktok, vtok, wq, tabq, wo, bo, lw, lb, x_shared=shared, out=out)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        ops.i2t_fold_layer(x, ktok, vtok, wq, tabq, wo, bo, lw, lb, x_shared=shared, out=out)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


for shared in (True, False):
    x = d(torch.randn(1 if shared else P, T, 256, generator=g)).to(dev)
    outs = {}
    for variant, wg in ((0, 2), (1, 2), (1, 1)):
        lib.msam_tune_set(b"i2t_variant", variant)
        lib.msam_tune_set(b"i2t_wg_per_cu", wg)
        out = torch.empty((P, T, 256), dtype=dt, device=dev)
        ms = run(x, shared, out)
        outs[(variant, wg)] = out
        moved = (1 if shared else P) * T * 256 * 2 + P * T * 256 * 2
        print(f"x_shared={shared} variant={variant} wg_per_cu={wg}: {ms:.3f} ms per launch (incl. fold kernels), "
              f"{moved / ms / 1e6:.0f} GB/s of stream traffic", flush=True)
    dd = (outs[(0, 2)].float() - outs[(1, 2)].float()).abs()
    print(f"  variant 1 vs 0: max |d| {dd.max().item():.4f}, mean |d| {dd.mean().item():.6f}; "
          f"wg 1 == wg 2: {torch.equal(outs[(1, 1)], outs[(1, 2)])}", flush=True)
lib.msam_tune_set(b"i2t_variant", 1)
lib.msam_tune_set(b"i2t_wg_per_cu", 2)
