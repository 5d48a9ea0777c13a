"""Shader clock under load, per kernel: python tools/clock_from_pmc.py <dir of one `rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace` run>
GRBM_GUI_ACTIVE counts the cycles the graphics engine was busy during a dispatch; divided by the dispatch's duration from the kernel
trace it is the clock the kernel actually ran at - what a nominal-peak roofline silently assumes to be 2.4 GHz.  (rocprofv3 may report the
counter summed over the 8 XCDs: the table prints cycles / ns as is and / 8.)"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    dur = {}
    for path in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            dur[r["Dispatch_Id"]] = (r["Kernel_Name"], float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    per = defaultdict(list)
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] != "GRBM_GUI_ACTIVE" or r["Dispatch_Id"] not in dur:
                continue
            name, ns = dur[r["Dispatch_Id"]]
            if ns > 20000:                          # >= 20 us: the counter's start / stop skew is below 1 %
                short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
                per[short].append((float(r["Counter_Value"]), ns))
    print("| kernel | dispatches | mean us | GRBM_GUI_ACTIVE / ns | / 8 |")
    print("|---|---|---|---|---|")
    for name, v in sorted(per.items(), key=lambda kv: -sum(x[1] for x in kv[1]))[:24]:
        cyc, ns = sum(x[0] for x in v), sum(x[1] for x in v)
        print(f"| `{name}` | {len(v)} | {ns / len(v) / 1e3:.1f} | {cyc / ns:.3f} | {cyc / ns / 8:.3f} |")


if __name__ == "__main__":
    main()
