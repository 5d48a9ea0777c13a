"""GPU-box probe: what does HBM give a write-only, a read-only and a copy stream of 2 GiB (torch fill / sum / copy kernels)?
Context for the stream kernels of the decoder (the layer-1 output stream is 2 GiB written per 1024-prompt pass).
    python tools/hbm_probe.py"""
import torch

dev = torch.device("cuda", 0)
n = 1 << 30                                   # 2 GiB of fp16
a = torch.empty(n, dtype=torch.float16, device=dev)
b = torch.empty(n, dtype=torch.float16, device=dev)


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


gib = 2.0
for name, fn, moved in (("write only (fill)", lambda: a.fill_(1.0), gib), ("write only (zero_)", lambda: a.zero_(), gib),
                        ("read only (sum)", lambda: a.view(torch.int32).sum(), gib), ("copy (read + write)", lambda: b.copy_(a), 2 * gib)):
    ms = timeit(fn)
    print(f"{name:22s} {ms:.3f} ms  {moved * 1.073741824 / ms:.2f} TB/s", flush=True)
