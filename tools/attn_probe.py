"""GPU-box timing of the image encoder's attention kernels at the bench shapes (16 tiles, vit_b: 12 heads x 64; vit_h: 16 heads x 80 in 96).
    python tools/attn_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from micro_sam_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)


def timeit(fn, n=20, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


for (B, heads, hd, hds, name) in [(16, 12, 64, 64, "vit_b"), (16, 16, 80, 96, "vit_h (head_dim 80 stored in 96)")]:
    q, k, v = (torch.randn(B, heads, 4096, hds, generator=g).to(torch.bfloat16).to(dev) for _ in range(3))
    if hds != hd:
        for t in (q, k, v):
            t[..., hd:] = 0
    rel_h = (torch.randn(127, hds, generator=g) * 0.1).to(torch.bfloat16).to(dev)
    rel_w = (torch.randn(127, hds, generator=g) * 0.1).to(torch.bfloat16).to(dev)
    ms = timeit(lambda: ops.global_attention(q, k, v, rel_h, rel_w, hd ** -0.5))
    flops = 4.0 * B * heads * 4096 * 4096 * hd
    print(f"global attention {name}: {ms:.3f} ms per {B}-tile launch, {flops / ms / 1e9:.0f} TFLOP/s (algorithmic, head_dim {hd})", flush=True)
    rh = (torch.randn(27, hds, generator=g) * 0.1).to(torch.bfloat16).to(dev)
    rw = (torch.randn(27, hds, generator=g) * 0.1).to(torch.bfloat16).to(dev)
    bias = torch.zeros(3 * heads * hds, device=dev)
    ms = timeit(lambda: ops.window_attention(q, k, v, rh, rw, bias, hd ** -0.5))
    print(f"window attention {name}: {ms:.3f} ms per {B}-tile launch, {4.0 * B * 25 * heads * 196 * 196 * hd / ms / 1e9:.0f} TFLOP/s (25 windows of 196 real tokens)",
          flush=True)
