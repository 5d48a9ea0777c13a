"""Per-kernel averages of rocprofv3 --pmc passes: python tools/pmc_summary.py gpurun_out/pmc_a gpurun_out/pmc_b ...
Prints a markdown table (kernel x counter, mean over dispatches; counters summed over XCDs / SEs as rocprofv3 reports them)."""
import csv
import glob
import os
import sys
from collections import defaultdict

KEEP = ("tok_layer", "up_fused", "fold_i2t", "fold_attn", "postprocess", "gemm256", "gemm_kernel", "global_attention", "window_attention",
        "t2i_shared4", "cc_hook", "nms_sweep", "fused_i2t", "i2t0_t2i", "i2t01", "i2t_tok")


def measured_sha(dirs):
    """csrc_sha16 recorded next to the passes (tools/final_measure.sh writes <gpurun_out>/final_csrc_sha.txt before them)."""
    for d in dirs:
        p = os.path.join(os.path.dirname(os.path.abspath(d.rstrip("/"))), "final_csrc_sha.txt")
        if os.path.exists(p):
            return open(p).read().strip()
    return None


def main():
    vals = defaultdict(lambda: defaultdict(list))
    sha = measured_sha(sys.argv[1:])
    # every table under profiles/ names the code it was measured on: the SQ file of round 4 carried no hash
    print(f"csrc_sha16 of the measured code: `{sha}` (tools/csrc_sha.py at measurement time; a table quoted for other code is stale)\n")
    for d in sys.argv[1:]:
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(path)):
                name = r.get("Kernel_Name") or r.get("Kernel Name") or ""
                short = next((k for k in KEEP if k in name), None)
                if short is None:
                    continue
                vals[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
    counters = sorted({c for k in vals for c in vals[k]})
    print("| kernel | n | " + " | ".join(counters) + " |")
    print("|---|---|" + "---|" * len(counters))
    for k in sorted(vals):
        n = max(len(v) for v in vals[k].values())
        row = []
        for c in counters:
            v = vals[k].get(c)
            row.append(f"{sum(v) / len(v):.4g}" if v else "")
        print(f"| {k} | {n} | " + " | ".join(row) + " |")


if __name__ == "__main__":
    main()
