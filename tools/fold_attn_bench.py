"""GPU-box A/B of fold_attn_kernel staging (registers, 2 workgroups per CU vs LDS-DMA, 3 per CU) at the AMG shape (P = 1024,
7 tokens per prompt): outputs must agree bit for bit (same arithmetic, different staging).  python tools/fold_attn_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from micro_sam_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda", 0)
lib = _lib.load()
g = torch.Generator().manual_seed(77)
P, Nt, T = 1024, 7, 4096
bf = lambda t: t.to(torch.bfloat16)
keys = bf(torch.randn(P, T, 256, generator=g)).to(dev)
wk = bf(torch.randn(128, 256, generator=g) / 16).to(dev)
wv = bf(torch.randn(128, 256, generator=g) / 16).to(dev)
bv = torch.randn(128, generator=g).to(dev)
qtok = bf(torch.randn(P, Nt, 128, generator=g) * 1.5).to(dev)
tabk = bf(torch.randn(T, 128, generator=g)).to(dev)
ref = None
for rep in range(6):
    dma = rep & 1
    lib.msam_fold_attn_set_dma(dma)
    for _ in range(2):
        out = ops.t2i_fold_attention(keys, qtok, wk, tabk, wv, bv, kv_shared=False)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        out = ops.t2i_fold_attention(keys, qtok, wk, tabk, wv, bv, kv_shared=False)
    b.record()
    torch.cuda.synchronize()
    if ref is None:
        ref = out.clone()
    ms = a.elapsed_time(b) / 5
    print(f"fold_attn dma={dma}: {ms:.3f} ms (incl. fold_q)  {2 * 1024 ** 3 / ms / 1e6:.0f} GB/s  same={bool(torch.equal(out, ref))} "
          f"finite={bool(torch.isfinite(out.float()).all())}", flush=True)
lib.msam_fold_attn_set_dma(0)
