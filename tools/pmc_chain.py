"""Short workload for rocprofv3 --pmc passes over the decoder stream kernels at the AMG shape (P = 1024, Nt = 7): two launches each
of the chained kernels, the token-owner layer kernel (both forms) and the stage-by-stage kernels they replace.
    rocprofv3 --kernel-trace --pmc <counters> --output-format csv -d gpurun_out/pmc_x -- python tools/pmc_chain.py"""
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
op0 = ops.i2t_fold_operands(L0["ktok"], L0["vtok"], L0["wq"], L0["wo"], L0["bo"], with_kfold=False)
op1 = ops.i2t_fold_operands(L1["ktok"], L1["vtok"], L1["wq"], L1["wo"], L1["bo"])
for _ in range(2):
    ops.i2t0_t2i_fused(tables, op0, L0["lw"], L0["lb"], qtok, wk, wv, bv)
    ops.i2t01_fused(tables, op0, L0["lw"], L0["lb"], op1, L1["lw"], L1["lb"], P, Nt)
    ops.i2t_fold_layer(src[None], L0["ktok"], L0["vtok"], L0["wq"], L0["tabq"], L0["wo"], L0["bo"], L0["lw"], L0["lb"], x_shared=True, out=keys)
    ops.t2i_fold_attention(keys, qtok, wk, tabk, wv, bv)
    ops.i2t_fold_layer(keys, L1["ktok"], L1["vtok"], L1["wq"], L1["tabq"], L1["wo"], L1["bo"], L1["lw"], L1["lb"], out=keys)
torch.cuda.synchronize()
print("done")
