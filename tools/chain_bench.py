"""GPU-box A/B of the chained layer-0 kernels (csrc/decfold_tok.hip) against the stage-by-stage kernels at the AMG shape
(P = 1024 prompts, Nt = 7):  python tools/chain_bench.py"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from micro_sam_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda", 0)
dt = _lib.decoder_dtype()
g = torch.Generator().manual_seed(1)
P, Nt, T = 1024, 7, 4096
d = lambda t: t.to(dt)
src = d(torch.randn(T, 256, generator=g)).to(dev)


def layer():
    return dict(wq=d(torch.randn(128, 256, generator=g) / 16).to(dev), wo=d(torch.randn(256, 128, generator=g) / math.sqrt(128)).to(dev),
                bo=torch.randn(256, generator=g).to(dev), lw=(torch.randn(256, generator=g) * 0.2 + 1).to(dev),
                lb=torch.randn(256, generator=g).to(dev), ktok=d(torch.randn(P, Nt, 128, generator=g)).to(dev),
                vtok=d(torch.randn(P, Nt, 128, generator=g)).to(dev), tabq=d(torch.randn(T, 128, generator=g)).to(dev))


L0, L1 = layer(), layer()
q0 = (src.float() @ L0["wq"].float().t() + L0["tabq"].float()).to(dt)
wk = d(torch.randn(128, 256, generator=g) / 16).to(dev); wv = d(torch.randn(128, 256, generator=g) / 16).to(dev)
bv = torch.randn(128, generator=g).to(dev); tabk = d(torch.randn(T, 128, generator=g)).to(dev)
qtok = d(torch.randn(P, Nt, 128, generator=g) * 1.5).to(dev)
tables = ops.chain_prepare_tables(src, q0, tabk, L1["tabq"])
keys = torch.empty((P, T, 256), dtype=dt, device=dev)


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def staged():
    ops.i2t_fold_layer(src[None], L0["ktok"], L0["vtok"], L0["wq"], L0["tabq"], L0["wo"], L0["bo"], L0["lw"], L0["lb"], x_shared=True, out=keys)
    att = ops.t2i_fold_attention(keys, qtok, wk, tabk, wv, bv)
    ops.i2t_fold_layer(keys, L1["ktok"], L1["vtok"], L1["wq"], L1["tabq"], L1["wo"], L1["bo"], L1["lw"], L1["lb"], out=keys)
    return att


def chained():
    op0 = ops.i2t_fold_operands(L0["ktok"], L0["vtok"], L0["wq"], L0["wo"], L0["bo"], with_kfold=False)
    att = ops.i2t0_t2i_fused(tables, op0, L0["lw"], L0["lb"], qtok, wk, wv, bv)
    op1 = ops.i2t_fold_operands(L1["ktok"], L1["vtok"], L1["wq"], L1["wo"], L1["bo"])
    ops.i2t01_fused(tables, op0, L0["lw"], L0["lb"], op1, L1["lw"], L1["lb"], P, Nt)
    return att


op0 = ops.i2t_fold_operands(L0["ktok"], L0["vtok"], L0["wq"], L0["wo"], L0["bo"], with_kfold=False)
op1 = ops.i2t_fold_operands(L1["ktok"], L1["vtok"], L1["wq"], L1["wo"], L1["bo"])
print(f"stage by stage (i2t layer 0 + t2i + i2t layer 1): {timeit(staged):.3f} ms", flush=True)
lib = _lib.load()
for variant in (0, 6):
    lib.msam_tune_set(b"chain_variant", variant)
    print(f"chain_variant {variant}: chained (operands + i2t0_t2i + operands + i2t01): {timeit(chained):.3f} ms", flush=True)
    print(f"  i2t0_t2i alone: {timeit(lambda: ops.i2t0_t2i_fused(tables, op0, L0['lw'], L0['lb'], qtok, wk, wv, bv)):.3f} ms", flush=True)
    print(f"  i2t01 alone:    {timeit(lambda: ops.i2t01_fused(tables, op0, L0['lw'], L0['lb'], op1, L1['lw'], L1['lb'], P, Nt)):.3f} ms", flush=True)
    a1, a2 = staged(), chained()
    print(f"  attention outputs vs stage by stage: max |d| {(a1.float() - a2.float()).abs().max().item():.4f}", flush=True)
lib.msam_tune_set(b"chain_variant", 9)
t2 = ops.chain_prepare_tables2(src, wv, bv, wk, L0["lw"], L0["lb"], L0["wo"], L0["bo"])
mf = ops.t2i_fold_values(L0["vtok"], t2)
print(f"second form: i2t0_t2i_v2 (incl. its fold kernel): {timeit(lambda: ops.i2t0_t2i_fused_v2(tables, t2, op0, mf, L0['lw'], qtok, wk)):.3f} ms; "
      f"tables2 {timeit(lambda: ops.chain_prepare_tables2(src, wv, bv, wk, L0['lw'], L0['lb'], L0['wo'], L0['bo'])):.3f} ms; "
      f"fold_values {timeit(lambda: ops.t2i_fold_values(L0['vtok'], t2)):.3f} ms", flush=True)
a2 = ops.i2t0_t2i_fused_v2(tables, t2, op0, mf, L0["lw"], qtok, wk)
print(f"  v2 attention outputs vs stage by stage: max |d| {(staged().float() - a2.float()).abs().max().item():.4f}", flush=True)
print(f"fold operands (with K'): {timeit(lambda: ops.i2t_fold_operands(L1['ktok'], L1['vtok'], L1['wq'], L1['wo'], L1['bo'])):.3f} ms", flush=True)
