"""(needs a library built with `python -m micro_sam_amd.build --experiments`: the production build ignores these knobs)
GPU-box probes of the large-shape GEMM kernels behind msam_gemm_bf16 (round 3; results: profiles/r03_experiments.md section 7).

    python tools/gemm_probe.py power      # k-loop / full kernel on random, zero and small-integer operands: what bounds the k-loop
    python tools/gemm_probe.py operands   # gemm_dbg 8 / 16: every operand k-tile read from k = 0 (cache hits)
    python tools/gemm_probe.py st4        # the two-workgroups-per-CU kernel (staging 4): times, start delays, timeline of the CUs
    python tools/gemm_probe.py dephase    # gemm256_kernel with a class-dependent start delay of the first dispatch wave ("g3_delay")
    python tools/gemm_probe.py epilogue   # gemm256_kernel: LDS-transposed epilogue vs stores straight from the accumulators ("g3_epi")

Timing experiments through msam_tune_set("gemm_dbg", bits): 1 = no global stores, 2 = no epilogue, 4 = no k-loop, 8 = operands always
from k-tile 0, 16 = only the weights from k-tile 0 (WRONG results by construction)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from micro_sam_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda", 0)
lib = _lib.load()
g = torch.Generator().manual_seed(3)
ENC = [(65536, 2304, 768, 0, "qkv"), (65536, 768, 768, 1, "proj"), (65536, 3072, 768, 0, "lin1"), (65536, 768, 3072, 1, "lin2")]


def check(rc):
    assert rc == 0, rc


def timeit(fn, n=30, warm=30):
    """(the clocks need some ten launches to settle: short warm-ups read 10 - 15 % high)"""
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


def operands(M, N, K, data="random"):
    if data == "random":
        a = (torch.rand(M, K, generator=g) * 2 - 1).to(torch.bfloat16).to(dev)
        w = (torch.rand(N, K, generator=g) * 2 - 1).to(torch.bfloat16).to(dev)
    elif data == "zeros":
        a = torch.zeros(M, K, dtype=torch.bfloat16, device=dev)
        w = torch.zeros(N, K, dtype=torch.bfloat16, device=dev)
    else:
        a = torch.randint(-1, 2, (M, K), generator=g).to(torch.bfloat16).to(dev)
        w = torch.randint(-1, 2, (N, K), generator=g).to(torch.bfloat16).to(dev)
    return a, w


def encoder_launch(M, N, K, resid):
    a, w = operands(M, N, K)
    bias = torch.randn(N, generator=g).to(dev)
    out = torch.zeros(M, N, dtype=torch.float32 if resid else torch.bfloat16, device=dev)
    if resid:
        return lambda: ops.gemm(a, w, bias, out=out, resid=out)
    return lambda: ops.gemm(a, w, bias, out=out, act=ops.ACT_GELU if N == 3072 else 0)


def power():
    for (M, N, K) in [(65536, 2304, 768), (65536, 768, 3072), (8192, 8192, 8192)]:
        for data in ("random", "zeros", "small-int"):
            a, w = operands(M, N, K, data)
            out = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
            line = []
            for st in (3, 4):
                check(lib.msam_gemm256_set_staging(st))
                for dbg in (0, 2):
                    check(lib.msam_tune_set(b"gemm_dbg", dbg))
                    ms = timeit(lambda: ops.gemm(a, w, None, out=out), 40, 40)
                    line.append(f"st{st} {'k-loop' if dbg else 'full  '}: {ms:.3f} ms {2 * M * N * K / ms / 1e9:6.0f} TF")
            check(lib.msam_tune_set(b"gemm_dbg", 0))
            print(f"{M}x{N}x{K} {data:9s} " + " | ".join(line), flush=True)


def operand_hits():
    for (M, N, K) in [(65536, 2304, 768), (65536, 768, 3072), (4096, 4096, 4096)]:
        a, w = operands(M, N, K)
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        for st in (3, 4):
            check(lib.msam_gemm256_set_staging(st))
            line = []
            for dbg in (0, 2, 2 | 8, 2 | 16):
                check(lib.msam_tune_set(b"gemm_dbg", dbg))
                ms = timeit(lambda: ops.gemm(a, w, None, out=out))
                line.append(f"dbg {dbg:2d}: {ms:.3f} ms {2 * M * N * K / ms / 1e9:7.1f} TF")
            check(lib.msam_tune_set(b"gemm_dbg", 0))
            print(f"{M}x{N}x{K} st{st}  " + " | ".join(line), flush=True)


def timeline(M, N, K, delay, cls):
    run = encoder_launch(M, N, K, 0)
    check(lib.msam_gemm256_set_staging(4))
    check(lib.msam_tune_set(b"gw_delay", delay))
    check(lib.msam_tune_set(b"gw_class", cls))
    ms = timeit(run)
    tr = torch.zeros(512 * 64, dtype=torch.int64, device=dev)
    check(lib.msam_gemm_set_trace(tr.data_ptr()))
    run()
    torch.cuda.synchronize()
    check(lib.msam_gemm_set_trace(None))
    t = tr.cpu().numpy().reshape(512, 64)
    hw, xcc = t[:, 0], t[:, 1] & 15
    cu = (xcc << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15)      # XCC, SE, SH, CU
    slot = hw & 15
    st = t[:, 2:].astype(np.float64)
    t0 = st[st > 0].min()
    n_st = (st > 0).sum(1)
    print(f"--- {M}x{N}x{K} staging 4, delay {delay} class rule {cls}: {ms:.3f} ms; distinct CUs {len(set(cu.tolist()))}, workgroups per CU "
          f"{np.bincount(np.unique(cu, return_counts=True)[1]).tolist()}, wave slots {np.bincount(slot).tolist()}")
    for c in list(dict.fromkeys(cu.tolist()))[:2]:              # per workgroup: tile start, end of its k-loop, ... exit (us)
        for b in np.nonzero(cu == c)[0]:
            print(f"  CU {c:#x} wg {b:3d} slot {slot[b]} :", " ".join(f"{(x - t0) / 100:.1f}" for x in st[b, :n_st[b]]))
    tot_e = tot_ov = 0.0
    for c in set(cu.tolist()):
        bs = np.nonzero(cu == c)[0]
        if len(bs) != 2:
            continue
        iv = []
        for b in bs:
            s_ = st[b, :n_st[b]]
            tiles = (n_st[b] - 1) // 2
            iv.append(([(s_[2 * i], s_[2 * i + 1]) for i in range(tiles)], [(s_[2 * i + 1], s_[2 * i + 2]) for i in range(tiles)]))
        for me, other in ((0, 1), (1, 0)):
            for (e0, e1) in iv[me][1]:
                tot_e += e1 - e0
                tot_ov += sum(max(0.0, min(e1, k1) - max(e0, k0)) for (k0, k1) in iv[other][0])
    print(f"  epilogue time under the co-resident workgroup's k-loop: {tot_ov / max(tot_e, 1):.2f} of {tot_e / 100 / 512:.1f} us per workgroup")


def st4():
    for (M, N, K, resid, name) in ENC:
        run = encoder_launch(M, N, K, resid)
        line = []
        for st in (3, 4, 3, 4):
            check(lib.msam_gemm256_set_staging(st))
            line.append(f"st{st} {timeit(run):.3f}")
        check(lib.msam_tune_set(b"gemm_dbg", 2))
        line.append(f"st4 k-loop only {timeit(run):.3f}")
        check(lib.msam_tune_set(b"gemm_dbg", 0))
        print(f"{name:5s} {M}x{N}x{K}  " + " | ".join(line), flush=True)
    for (delay, cls) in ((0, 0), (6, 0), (6, 2)):
        timeline(65536, 2304, 768, delay, cls)
    check(lib.msam_tune_set(b"gw_delay", -1))
    check(lib.msam_tune_set(b"gw_class", 0))


def dephase():
    check(lib.msam_gemm256_set_staging(3))
    for (M, N, K, resid, name) in ENC:
        run = encoder_launch(M, N, K, resid)
        line = []
        for rep in range(2):
            for d in (0, 1, 2, 4, 8):
                check(lib.msam_tune_set(b"g3_delay", d))
                line.append(f"d{d} {timeit(run, 30, 30 if not line else 5):.3f}")
        check(lib.msam_tune_set(b"g3_delay", 0))
        print(f"{name:5s} {M}x{N}x{K}  " + " | ".join(line), flush=True)


def epilogue():
    check(lib.msam_gemm256_set_staging(3))
    for (M, N, K, resid, name) in ENC:
        run = encoder_launch(M, N, K, resid)
        line = []
        for rep in range(3):
            for epi in (0, 1):
                check(lib.msam_tune_set(b"g3_epi", epi))
                line.append(f"epi{epi} {timeit(run, 30, 30 if not line else 5):.3f}")
        check(lib.msam_tune_set(b"g3_epi", 0))
        print(f"{name:5s} {M}x{N}x{K}  " + " | ".join(line), flush=True)


if __name__ == "__main__":
    for what in (sys.argv[1:] or ["power"]):
        print(f"=== {what}", flush=True)
        {"power": power, "operands": operand_hits, "st4": st4, "dephase": dephase, "epilogue": epilogue}[what]()
    check(lib.msam_gemm256_set_staging(-1))
