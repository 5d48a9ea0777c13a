"""GPU-box experiment: what do the phases of the chained layer-0 -> layer-1 attention kernel (i2t0_t2i, csrc/decfold_tok.hip) cost
in place?  Times the ablation builds (msam_tune_set "chain_variant" 100 + mask; results of those builds are meaningless) at the
AMG shape (P = 1024 prompts, Nt = 7):  python tools/chain_ablation.py"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from micro_sam_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda", 0)
dt = _lib.decoder_dtype()
lib = _lib.load()
g = torch.Generator().manual_seed(1)
P, Nt, T = 1024, 7, 4096
d = lambda t: t.to(dt)
src = d(torch.randn(T, 256, generator=g)).to(dev)
L0 = dict(wq=d(torch.randn(128, 256, generator=g) / 16).to(dev), wo=d(torch.randn(256, 128, generator=g) / math.sqrt(128)).to(dev),
          bo=torch.randn(256, generator=g).to(dev), lw=(torch.randn(256, generator=g) * 0.2 + 1).to(dev),
          lb=torch.randn(256, generator=g).to(dev), ktok=d(torch.randn(P, Nt, 128, generator=g)).to(dev),
          vtok=d(torch.randn(P, Nt, 128, generator=g)).to(dev), tabq=d(torch.randn(T, 128, generator=g)).to(dev))
q0 = (src.float() @ L0["wq"].float().t() + L0["tabq"].float()).to(dt)
wk = d(torch.randn(128, 256, generator=g) / 16).to(dev); wv = d(torch.randn(128, 256, generator=g) / 16).to(dev)
bv = torch.randn(128, generator=g).to(dev); tabk = d(torch.randn(T, 128, generator=g)).to(dev)
qtok = d(torch.randn(P, Nt, 128, generator=g) * 1.5).to(dev)
op0 = ops.i2t_fold_operands(L0["ktok"], L0["vtok"], L0["wq"], L0["wo"], L0["bo"], with_kfold=False)
tables = ops.chain_prepare_tables(src, q0, tabk, L0["tabq"])


def timeit(reps=6):
    f = lambda: ops.i2t0_t2i_fused(tables, op0, L0["lw"], L0["lb"], qtok, wk, wv, bv)
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


NAMES = {3: "full (8 waves, ring of 4)", 101: "no layer-0 score MFMAs", 102: "no layer-0 V'^T MFMAs", 104: "no LayerNorm",
         108: "no attention score MFMAs", 116: "no V-projection MFMAs", 132: "no LDS operand reads", 164: "no softmax exps",
         127: "no MFMA phase at all, LayerNorm kept (1+2+8+16)", 131: "no MFMA phase, no LayerNorm", 227: "everything off"}
for v, name in NAMES.items():
    lib.msam_tune_set(b"chain_variant", v)
    print(f"chain_variant {v:3d}  {name:52s} {timeit():.3f} ms", flush=True)
# shared-table loads confined to the first 32 tiles (0.5 MB of the 4 MB of src / q0 / tabk): cache-miss share of the skeleton
for v in (3, 227, 132):
    for tm in (255, 31, 7):
        lib.msam_tune_set(b"chain_variant", v); lib.msam_tune_set(b"chain_tmask", tm)
        print(f"chain_variant {v:3d} tile mask {tm:3d}: {timeit():.3f} ms", flush=True)
lib.msam_tune_set(b"chain_tmask", 255)
lib.msam_tune_set(b"chain_variant", 9)
