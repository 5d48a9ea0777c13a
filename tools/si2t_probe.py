"""Timing experiments of si2t_kernel (csrc/strict.hip; msam_tune_set("si2t_dbg", bits) removes phases - WRONG results, timing only):
python tools/si2t_probe.py -> one JSON line (us per launch of 128 prompts, 7 tokens)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    from micro_sam_amd import _lib, strict
    dev = torch.device("cuda")
    B, Tk = 128, 7
    g = torch.Generator(device="cpu").manual_seed(0)
    r = lambda *s: torch.randn(*s, generator=g).to(dev)
    keys, pos = r(B * 4096, 256), r(4096, 256)
    wq, wo = (r(128, 256) / 16, r(128) * 0.1), (r(256, 128) / 11, r(256) * 0.1)
    tok_k, tok_v = r(B * Tk, 128), r(B * Tk, 128)
    norm = (torch.rand(256, generator=g).to(dev) + 0.5, r(256) * 0.2, 1e-5)
    out = torch.empty_like(keys)
    lib = _lib.load()
    rec = {}

    def launch_us():
        for _ in range(3):
            strict.i2t_block(keys, False, pos, wq, tok_k, tok_v, wo, norm, B, Tk, out=out)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            strict.i2t_block(keys, False, pos, wq, tok_k, tok_v, wo, norm, B, Tk, out=out)
        b.record()
        torch.cuda.synchronize()
        return round(a.elapsed_time(b) * 100.0, 1)
    for us in (0, 5, 10, 15, 20, 25, 30, 40, 50, 0):
        _lib.check(lib.msam_tune_set(b"si2t_late_us", us), "tune")
        rec[f"late_{us}us" + ("_again" if f"late_{us}us" in rec else "")] = launch_us()
    _lib.check(lib.msam_tune_set(b"si2t_late_us", 0), "tune")
    for name, bits in (("full", 0), ("no_proj1", 1), ("no_attention", 2), ("no_proj2", 4), ("no_residual_ln", 8), ("no_proj1_attention", 3),
                       ("only_proj1", 14), ("only_attention", 13), ("only_proj2", 11), ("only_epilogue", 7), ("nothing", 15), ("full_again", 0)):
        _lib.check(lib.msam_tune_set(b"si2t_dbg", bits), "tune")
        for _ in range(3):
            strict.i2t_block(keys, False, pos, wq, tok_k, tok_v, wo, norm, B, Tk, out=out)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            strict.i2t_block(keys, False, pos, wq, tok_k, tok_v, wo, norm, B, Tk, out=out)
        b.record()
        torch.cuda.synchronize()
        rec[name] = round(a.elapsed_time(b) * 100.0, 1)
    _lib.check(lib.msam_tune_set(b"si2t_dbg", 0), "tune")
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
