#!/usr/bin/env python3
"""Static instruction mix of a kernel's loops, from the gfx950 assembly hipcc emits for one source file (no GPU needed).

    python tools/isa_mix.py micro_sam_amd/csrc/upfused.hip 'up_fused_kernelILi1ELi1E' [-D MSAM_DEC_F16=1]

For every backward branch of the matching kernel it prints the instruction classes between the loop head and the branch (MFMA, plain
VALU, packed VALU, transcendental, LDS, VMEM, SALU) with the most frequent opcodes, and an estimate of the pipe time in SIMD cycles:
4 cycles per vector instruction (packed or not), 9 per transcendental, 16 (17) per 16x16x32 16-bit MFMA, 32 per 32x32x16 - calibrated on
up_fused_kernel, whose tile loop this model prices at 1749 VALU + 952 MFMA cycles against SQ_ACTIVE_INST_VALU = 1770 and
SQ_VALU_MFMA_BUSY_CYCLES = 896 per tile and wave (profiles/r03_experiments.md section 10).  In the decoder's kernels the two times ADD UP to
the measured duration (section 11), so the sum is the number to watch."""
import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith(("v_exp", "v_rsq", "v_rcp", "v_log", "v_sqrt", "v_sin", "v_cos")):
        return "trans"
    if op.startswith("v_pk_"):
        return "vpk"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_"):
        return "salu"
    return "other"


def issue_cycles(ops):
    valu = mfma = 0.0
    for op, n in ops.items():
        k = classify(op)
        if k == "mfma":
            mfma += n * (32 if "32x32" in op else 17)
        elif k == "trans":
            valu += n * 9
        elif k in ("vpk", "valu"):
            valu += n * 4
    return valu, mfma


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("source")
    ap.add_argument("kernel", help="regex over the mangled kernel names")
    ap.add_argument("-D", action="append", default=[])
    ap.add_argument("--top", type=int, default=14)
    a = ap.parse_args()
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", *[f"-D{x}" for x in a.D], "-o", out,
               a.source]
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        text = open(out).read()
    names = [n for n in re.findall(r"^(\S+):\s*; @", text, re.M) if re.search(a.kernel, n)]
    if not names:
        sys.exit("no kernel matches; kernels: " + ", ".join(re.findall(r"^(\S+):\s*; @", text, re.M)))
    for name in names:
        body = text[text.index(name + ":"):]
        body = body[:body.index(".Lfunc_end")]
        m = re.search(r"; NumVgprs: (\d+)", text[text.index(name + ":"):])
        m2 = re.search(r"; NumAgprs: (\d+)", text[text.index(name + ":"):])
        m3 = re.search(r"; ScratchSize: (\d+)", text[text.index(name + ":"):])
        print(f"== {name}: VGPRs {m.group(1) if m else '?'}, AGPRs {m2.group(1) if m2 else '?'}, scratch {m3.group(1) if m3 else '?'} B")
        lines = body.split("\n")
        labels = {mm.group(1): i for i, l in enumerate(lines) if (mm := re.match(r"^(\.LBB\d+_\d+):", l))}
        for i, l in enumerate(lines):
            mm = re.search(r"s_c?branch\S*\s+(\.LBB\d+_\d+)", l)
            if not mm or mm.group(1) not in labels or labels[mm.group(1)] >= i:
                continue
            ops = collections.Counter()
            for x in lines[labels[mm.group(1)]:i + 1]:
                x = x.strip()
                if not x or x[0] in ".;/" or x.endswith(":"):
                    continue
                ops[x.split()[0]] += 1
            cls = collections.Counter()
            for op, n in ops.items():
                cls[classify(op)] += n
            valu, mfma = issue_cycles(ops)
            print(f"-- loop {mm.group(1)} .. line {i}: {sum(ops.values())} instructions  " + "  ".join(f"{k} {v}" for k, v in sorted(cls.items())))
            print(f"   pipe-time estimate: VALU {valu:.0f} + MFMA {mfma:.0f} = {valu + mfma:.0f} cycles per trip")
            print("   " + ", ".join(f"{op} {n}" for op, n in ops.most_common(a.top)))


if __name__ == "__main__":
    main()
