"""GPU-box micro-benchmark of msam_upscale_fused at the AMG shape (P = 1024 prompts, 3 masks): python tools/upfused_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from micro_sam_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(5)
P = 1024
bf = lambda t: t.to(_lib.decoder_dtype())
keys = bf(torch.randn(P, 4096, 256, generator=g)).to(dev)
w1 = bf(torch.randn(256, 256, generator=g) / 16).to(dev); b1 = torch.randn(256, generator=g).to(dev)
lw = (torch.randn(64, generator=g) * 0.2 + 1).to(dev); lb = (torch.randn(64, generator=g) * 0.3).to(dev)
w2 = bf(torch.randn(128, 64, generator=g) / 8).to(dev); b2 = torch.randn(32, generator=g).to(dev)
hyper = torch.randn(P, 4, 128, generator=g).to(dev)
for rep in range(6):
    _lib.load().msam_tune_set(b"up_gelu16", rep & 1)
    for _ in range(2):
        ops.upscale_fused(keys, w1, b1, lw, lb, w2, b2, hyper, 1, 3)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        out = ops.upscale_fused(keys, w1, b1, lw, lb, w2, b2, hyper, 1, 3)
    b.record()
    torch.cuda.synchronize()
    print(f"up_fused gelu16={rep & 1} P={P}: {a.elapsed_time(b) / 5:.3f} ms  checksum {float(out.float().abs().mean()):.6f}", flush=True)
