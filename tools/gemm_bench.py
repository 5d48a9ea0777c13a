"""GPU-box micro-benchmark of the encoder's GEMM shapes (vit_b, batch of 8 tiles: M = 65536) across the operand staging
variants of the 256 x 256 tile kernel; each variant is checked against the first one (bit-identical accumulation order).
    python tools/gemm_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from micro_sam_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda", 0)
lib = _lib.load()
g = torch.Generator().manual_seed(3)
SHAPES = [(65536, 2304, 768, "qkv"), (65536, 768, 768, "proj"), (65536, 3072, 768, "lin1"), (65536, 768, 3072, "lin2"),
          (65536, 3072, 1024, "vit_l lin1 (N=3072 stand-in)")]


def timeit(fn, n=10, warm=3):
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


for (M, N, K, name) in SHAPES:
    a = (torch.rand(M, K, generator=g) * 2 - 1).to(torch.bfloat16).to(dev)
    w = (torch.rand(N, K, generator=g) * 2 - 1).to(torch.bfloat16).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    ref = None
    line = []
    for st in (3, 4, 3, 4, 4):
        assert lib.msam_gemm256_set_staging(st) == 0
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        ms = timeit(lambda: ops.gemm(a, w, bias, out=out, act=ops.ACT_GELU))
        if ref is None:
            ref = out.clone()
        same = bool(torch.equal(out, ref))
        line.append(f"st{st}: {ms:.3f} ms {2 * M * N * K / ms / 1e9:7.1f} TF {'ok' if same else 'DIFF'}")
    print(f"{name:8s} {M}x{N}x{K}  " + " | ".join(line), flush=True)
lib.msam_gemm256_set_staging(-1)
# fp8 (MX MFMA, unit block scales) on the same shapes
for (M, N, K, name) in SHAPES:
    a = (torch.rand(M, K, generator=g) * 2 - 1).to(torch.bfloat16).to(dev)
    a8, a_sc = ops.quant_rows_fp8(a)
    w8, w_sc = ops.quant_weight_fp8(torch.rand(N, K, generator=g) * 2 - 1)
    w8, w_sc = w8.to(dev), w_sc.to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    ms = timeit(lambda: ops.gemm_fp8(a8, a_sc, w8, w_sc, bias, out=out, act=ops.ACT_GELU))
    print(f"{name:8s} {M}x{N}x{K}  fp8: {ms:.3f} ms {2 * M * N * K / ms / 1e9:7.1f} TF", flush=True)

# where the time of the 256 x 256 kernel goes (timing experiments, WRONG results): full / no global stores / k-loop only / epilogue only;
# real epilogues of the encoder: qkv head-split bf16 store, proj + lin2 with the fp32 residual read and written in place, lin1 GELU
for ST in (3, 4):
  print(f"\ngemm_dbg experiments, staging {ST} (ms): full | no stores (1) | k-loop only (2) | prologue + epilogue only (4)", flush=True)
  for (M, N, K, name) in SHAPES[:4]:
      a = (torch.rand(M, K, generator=g) * 2 - 1).to(torch.bfloat16).to(dev)
      w = (torch.rand(N, K, generator=g) * 2 - 1).to(torch.bfloat16).to(dev)
      bias = torch.randn(N, generator=g).to(dev)
      resid = name in ("proj", "lin2")
      out = torch.zeros(M, N, dtype=torch.float32 if resid else torch.bfloat16, device=dev)
      line = []
      for dbg in (0, 1, 2, 4, 0):
          lib.msam_tune_set(b"gemm_dbg", dbg)
          assert lib.msam_gemm256_set_staging(ST) == 0
          if resid:
              ms = timeit(lambda: ops.gemm(a, w, bias, out=out, resid=out))
          else:
              ms = timeit(lambda: ops.gemm(a, w, bias, out=out, act=ops.ACT_GELU if name == "lin1" else 0))
          line.append(f"{ms:.3f}")
      lib.msam_tune_set(b"gemm_dbg", 0)
      print(f"{name:6s} {M}x{N}x{K} resid={int(resid)}: " + " | ".join(line) + f"   ({2 * M * N * K / float(line[0]) / 1e9:.0f} TF full)", flush=True)

lib.msam_gemm256_set_staging(-1)
