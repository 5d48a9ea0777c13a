"""SQ counters of the strict mode's kernels: python tools/strict_sq_summary.py gpurun_out/strict_sq  (one rocprofv3 --pmc pass of
`python tools/strict_probe.py --modes strict --tiles 1` with SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES
SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS).  Normalisation (the one under which `up_fused_kernel`'s shares add up to 101 %,
profiles/r05_experiments.md section 1): SQ_BUSY_CYCLES is summed over the 32 shader engines (launch cycles = / 32), SQ_VALU_MFMA_BUSY_CYCLES over the
1024 SIMDs, SQ_WAVE_CYCLES and SQ_ACTIVE_INST_VALU count in units of 4 cycles summed over waves."""
import collections
import csv
import glob
import os
import sys


def main():
    f = glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True)[0]
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:40]
        if "sgemm" in n:
            n += f" x {int(r['Grid_Size']) // 256} tiles"
        vals[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
    rows = []
    for n, c in vals.items():
        m = {k: sum(v) / len(v) for k, v in c.items()}
        if m.get("SQ_BUSY_CYCLES", 0) < 2e5:
            continue
        cyc = m["SQ_BUSY_CYCLES"] / 32
        simd = cyc * 1024
        mf = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / simd
        waves = 4 * m.get("SQ_WAVE_CYCLES", 0) / simd
        va_wave = m.get("SQ_ACTIVE_INST_VALU", 0) / max(m.get("SQ_WAVE_CYCLES", 1), 1)
        rows.append((cyc * len(c["SQ_BUSY_CYCLES"]), n, len(c["SQ_BUSY_CYCLES"]), cyc / 2.4e3, mf, va_wave, waves, va_wave * waves,
                     m.get("SQ_INSTS_MFMA", 0), m.get("SQ_INSTS_VALU", 0)))
    rows.sort(reverse=True)
    print("| kernel | launches | us per launch (busy cycles / 2.4 GHz) | matrix-busy share of a SIMD's time | waves per SIMD | vector-active share of a wave's time | "
          "x waves = vector share of the SIMD | matrix + vector | MFMA instructions | vector instructions |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    for _, n, k, us, mf, va, w, vs, im, iv in rows[:18]:
        print(f"| `{n}` | {k} | {us:.0f} | {mf:.2f} | {w:.1f} | {va:.2f} | {vs:.2f} | {mf + vs:.2f} | {im:.3g} | {iv:.3g} |")


if __name__ == "__main__":
    main()
