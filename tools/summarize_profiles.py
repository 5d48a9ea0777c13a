"""Turn the rocprofv3 outputs of a measurement run (gpurun_out/final_*) into the committed summaries under profiles/.

Usage: python tools/summarize_profiles.py [round_tag]          (default r01)
Inputs (written on the GPU box by the commands quoted in profiles/<tag>_bench_kernel_summary.md):
  gpurun_out/final_bench.log                    python bench.py                          (bench line incl. cpu_baseline)
  gpurun_out/final_prof/*/*_kernel_stats.csv   rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline
  gpurun_out/final_fetch, final_write           rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes)
"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = next((a for a in sys.argv[1:] if not a.startswith("--")), "r01")
OUT = os.path.join(ROOT, "profiles")
GO = os.path.join(ROOT, "gpurun_out")

# algorithmic HBM bytes per launch at P = 1024 prompts: stream = 1024 * 4096 * 256 * 2 B = 2 GiB
GiB = 1 << 30
# key (the name bench.py's families use) -> (description, algorithmic bytes per launch); PATTERN maps a key to the substring of the
# rocprof kernel name it stands for (the shipped chained kernels are i2t01_ring_kernel / i2t0_t2i_v2_kernel)
PATTERN = {"i2t01_kernel": "i2t01", "i2t0_t2i_kernel": "i2t0_t2i", "gemm256_kernel": "gemm256_kernel"}
ALGO = {
    "gemm256_kernel": ("A [65536, K] and W read, C written (encoder projections of a 16-tile batch; mixed shapes)", None),
    "i2t01_kernel": ("2 GiB written (the layer-1 stream, blocked layout); the shared tables (5 MiB) and the per-prompt operands "
                     "(144 KiB per prompt) come from L2", 2 * GiB),
    "i2t0_t2i_kernel": ("no per-prompt stream: shared tables (5 MiB) + operands (112 KiB per prompt) from L2, [P,7,128] written",
                        1024 * (112 * 1024 + 7 * 128 * 2)),
    "fold_i2t_kernel": ("layer-1 launches (the larger half): 2 GiB read + 2 GiB written in place; layer 0 writes 2 GiB only", 4 * GiB),
    "fold_attn_kernel": ("2 GiB read (value projection fused: only [P,7,128] bf16 written)", 2 * GiB),
    "up_fused_kernel": ("2 GiB read + 0.75 GiB fp32 low-res logits written", 2 * GiB + 3 * 1024 * 65536 * 4),
    "postprocess_kernel": ("0.75 GiB fp32 low-res read + 384 MiB bit masks written", 3 * 1024 * 65536 * 4 + 3072 * 128 * 1024),
    "gemm_kernel": ("A and W read once, C written once (mixed shapes)", None),
    "global_attention_kernel": ("q, k, v read once, out written (8 tiles x 12 heads)", None),
}


def one(pattern):
    files = glob.glob(pattern)
    if not files:
        raise SystemExit(f"missing {pattern}")
    # gpurun merges new files INTO gpurun_out: results of earlier calls (even earlier rounds) stay next to them - take the newest
    return max(files, key=os.path.getmtime)


def main():
    os.makedirs(OUT, exist_ok=True)
    # every table names the code it was measured on, and a measurement of OTHER code is refused (round 4's kernel
    # summary and SQ table described the build before the last csrc commit)
    sys.path.insert(0, ROOT)
    import bench
    sha_path = os.path.join(GO, "final_csrc_sha.txt")
    measured = open(sha_path).read().strip() if os.path.exists(sha_path) else None
    if measured != bench.csrc_sha16() and "--force" not in sys.argv:
        raise SystemExit(f"gpurun_out/final_* was measured on csrc_sha16 {measured}, the working tree is {bench.csrc_sha16()}: re-run "
                         "tools/final_measure.sh on this code (or pass --force to summarise the old measurement under its own hash)")
    stamp = f"csrc_sha16 of the measured code: `{measured}`\n\n"
    line = open(os.path.join(GO, "final_bench.log")).read().strip().splitlines()[-1]
    bench = json.loads(line)
    with open(os.path.join(OUT, f"{TAG}_bench_line.json"), "w") as f:
        json.dump(bench, f, indent=1)
    stats = one(os.path.join(GO, "final_prof", "*", "*_kernel_stats.csv"))
    shutil.copy(stats, os.path.join(OUT, f"{TAG}_bench_kernel_stats.csv"))
    rows = list(csv.DictReader(open(stats)))
    total = sum(float(r["TotalDurationNs"]) for r in rows)
    prof_line = [l for l in open(os.path.join(GO, "final_prof.log")) if l.startswith("{")][-1]
    pb = json.loads(prof_line)
    # tiles = launches of a once-per-tile kernel in THIS trace (the step arithmetic missed the pipelined re-check pass and
    # made every "ms / tile" 20 % high); the step arithmetic is only the fallback
    per_tile = [int(r["Calls"]) for r in rows if "up_fused_kernel" in r["Name"]]
    tiles = per_tile[0] if per_tile else pb["config"]["tiles_per_step_per_gpu"] * (pb["steps"] + pb["warmup"] + 2)
    with open(os.path.join(OUT, f"{TAG}_bench_kernel_summary.md"), "w") as f:
        f.write(f"# {TAG}: rocprofv3 kernel summary of `python bench.py --no-cpu-baseline --no-side --lanes 1` on one MI355X\n\n")
        f.write(stamp)
        f.write("Command: `rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final_prof -- python bench.py "
                "--no-cpu-baseline --no-side --lanes 1` (no side measurements: every launch belongs to the hot path; one decode lane: kernels of different tiles do not overlap, the same order as the "
                "serial roofline pass of bench.py)\n")
        f.write(f"(bench line of this profiled run: {pb['value']} tiles/s; unprofiled run with cpu_baseline: {bench['value']} "
                f"tiles/s, `{TAG}_bench_line.json`).\n\n")
        f.write(f"GPU time {total / 1e6:.1f} ms over {tiles} tiles (= launches of the once-per-tile `up_fused_kernel` in this trace: "
                f"{pb['warmup']} warm-up + {pb['steps']} timed + the serial instrumented pass + the pipelined re-check pass, steps of "
                f"{pb['config']['tiles_per_step_per_gpu']}) = **{total / 1e6 / tiles:.2f} ms per tile**.\n\n")
        f.write("| kernel | calls | total ms | ms / tile | avg us | % |\n|---|---|---|---|---|---|\n")
        for r in rows[:24]:
            t = float(r["TotalDurationNs"])
            f.write(f"| `{r['Name'][:90]}` | {r['Calls']} | {t / 1e6:.2f} | {t / 1e6 / tiles:.3f} | {float(r['AverageNs']) / 1e3:.1f} | "
                    f"{float(r['Percentage']):.1f} |\n")
        rf = bench["roofline"]
        extras = os.path.join(GO, "final_bench_extras.json")          # round 6: the final line carries the dominant kernel only; the families are in the full tree
        if os.path.exists(extras):
            rf = json.load(open(extras)).get("roofline", rf)
        f.write("\nLive HIP-event measurement of the same kernels inside bench.py (timed region, unprofiled run):\n\n")
        f.write("| kernel family | bound | achieved | peak | frac | avg launch us |\n|---|---|---|---|---|---|\n")
        for k in [rf] + rf.get("other_kernels", []):
            f.write(f"| {k['kernel'][:70]} | {k['bound']} | {k['achieved']} {k['unit']} | {k['peak']} | {k['frac']} | {k['avg_launch_us']} |\n")
    # ---- PMC traffic
    acc = {}
    for name, key in (("final_fetch", "FETCH_SIZE"), ("final_write", "WRITE_SIZE")):
        path = one(os.path.join(GO, name, "*", "*_counter_collection.csv"))
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] != key:
                continue
            kn = r["Kernel_Name"]
            short = next((s for s in ALGO if PATTERN.get(s, s) in kn), None)
            if short is None:
                continue
            acc.setdefault(short, {}).setdefault(key, []).append(float(r["Counter_Value"]))
    table = {}
    for short, d in acc.items():
        fetch, write = d.get("FETCH_SIZE", []), d.get("WRITE_SIZE", [])
        # the large launches only (P = 1024 decoder passes / 8-tile encoder batches): top half by bytes
        def top(v):
            v = sorted(v, reverse=True)
            return v[: max(1, len(v) // 2)] if short != "gemm_kernel" else v
        ft, wt = top(fetch), top(write)
        fk, wk = sum(ft) / max(len(ft), 1), sum(wt) / max(len(wt), 1)
        # MI355X_MICROARCH.md (HBM section): counters are KiB; FETCH_SIZE reports half of the bytes of wide streaming reads
        hbm = 2 * fk * 1024 + wk * 1024
        table[short] = {"launches": len(fetch), "fetch_kib": fk, "write_kib": wk, "hbm_bytes_per_launch": hbm,
                        "algorithmic_bytes_per_launch": ALGO[short][1], "algorithmic": ALGO[short][0]}
    sha_path = os.path.join(GO, "final_csrc_sha.txt")
    table["_meta"] = {"csrc_sha16": open(sha_path).read().strip() if os.path.exists(sha_path) else None,
                      "what": "sha16 of micro_sam_amd/csrc + include/msam_hip.h at measurement time (tools/csrc_sha.py); bench.py "
                              "reports roofline.traffic from this table only while the sources still hash to it"}
    with open(os.path.join(OUT, f"{TAG}_pmc_traffic.json"), "w") as f:
        json.dump(table, f, indent=1)
    table.pop("_meta")
    with open(os.path.join(OUT, f"{TAG}_pmc_traffic.md"), "w") as f:
        f.write(f"# {TAG}: HBM traffic from PMC counters (separate rocprofv3 passes of `python bench.py --steps 1 --warmup 1 "
                "--no-cpu-baseline --no-side`)\n\n")
        f.write(stamp)
        f.write("Commands: `rocprofv3 --kernel-trace --pmc FETCH_SIZE ...` and `rocprofv3 --kernel-trace --pmc WRITE_SIZE ...` "
                "(one counter per pass, no other trace domain).\n\n")
        f.write("Units: counter values are KiB.  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports 1/2 of "
                "the bytes of wide coalesced streaming reads, so read bytes = 2 x FETCH_SIZE x 1024; WRITE_SIZE is taken as is "
                "(x 1024).  Averages over the larger half of each kernel's launches (the P = 1024 decoder passes).\n\n")
        f.write("`postprocess_kernel` reads 4-byte elements through a small LDS patch (not 16-byte streaming loads); the x2 read "
                "correction is uncalibrated for that access pattern, so its ratio is an upper bound.\n\n")
        f.write("| kernel | launches | FETCH_SIZE avg (KiB) | WRITE_SIZE avg (KiB) | HBM bytes / launch (corrected) | algorithmic bytes "
                "/ launch | ratio | algorithmic traffic |\n|---|---|---|---|---|---|---|---|\n")
        for short, r in sorted(table.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"]):
            alg = r["algorithmic_bytes_per_launch"]
            ratio = "-" if alg is None else f"{r['hbm_bytes_per_launch'] / alg:.2f}"
            f.write(f"| `{short}` | {r['launches']} | {r['fetch_kib']:.0f} | {r['write_kib']:.0f} | {r['hbm_bytes_per_launch'] / 1e9:.3f} GB | "
                    f"{'-' if alg is None else f'{alg / 1e9:.3f} GB'} | {ratio} | {r['algorithmic']} |\n")
    # ---- the strict precision mode's kernel table (tools/strict_probe.py under rocprofv3: both modes' per-tile API loop + product timings)
    sp = glob.glob(os.path.join(GO, "final_strict_prof", "*", "*_kernel_stats.csv"))
    if sp:
        rows = list(csv.DictReader(open(max(sp, key=os.path.getmtime))))
        probe = os.path.join(GO, "final_strict_probe.json")
        with open(os.path.join(OUT, f"{TAG}_strict_kernel_summary.md"), "w") as f:
            f.write(f"# {TAG}: rocprofv3 kernel summary of `python tools/strict_probe.py` (the strict precision mode next to the default one)\n\n")
            f.write(stamp)
            f.write("Command: `rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final_strict_prof -- python tools/strict_probe.py`: per mode "
                    "(default, strict) 5 passes of the per-tile API loop over 2 tiles + 5 encoder-only passes, then the f32-input MFMA product on seven shapes "
                    "of the path (7 launches each).  Kernels of `csrc/strict.hip`: `sgemm_kernel<CONV, stages, tile>`, `srelpos_mfma_kernel`, `srelpos_kernel`, `si2t_kernel`, `sattn_*`, `sln*`, `shyper_kernel`, "
                    "`spatchify / sim2col / ssrc`.\n\n")
            if os.path.exists(probe):
                f.write("Probe line of the same run: `" + open(probe).read().strip() + "`\n\n")
            f.write("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|\n")
            for r in rows[:30]:
                f.write(f"| `{r['Name'][:100]}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.2f} | {float(r['AverageNs']) / 1e3:.1f} | "
                        f"{float(r['Percentage']):.1f} |\n")
    # ---- the split16 precision mode's kernel table (tools/split16_probe.py under rocprofv3)
    sp = glob.glob(os.path.join(GO, "final_split16_prof", "*", "*_kernel_stats.csv"))
    if sp:
        rows = list(csv.DictReader(open(max(sp, key=os.path.getmtime))))
        probe = os.path.join(GO, "final_split16_probe.json")
        per_tile = [int(r["Calls"]) for r in rows if "s16_up2_kernel" in r["Name"]]
        tiles = per_tile[0] if per_tile else 12
        total = sum(float(r["TotalDurationNs"]) for r in rows)
        with open(os.path.join(OUT, f"{TAG}_split16_kernel_summary.md"), "w") as f:
            f.write(f"# {TAG}: rocprofv3 kernel summary of `python tools/split16_probe.py --modes split16 --no-products --slice-tiles 0` (the split16 precision mode)\n\n")
            f.write(stamp)
            f.write("Command: `rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final_split16_prof -- python tools/split16_probe.py --modes split16 --no-products "
                    f"--slice-tiles 0`: {tiles} tile decodes (1024 prompts each, one pass) and 22 tile encodes at batch 1 through the literal API loop.  GPU time {total / 1e6:.1f} ms.\n\n")
            if os.path.exists(probe):
                d = json.load(open(probe))
                for k in ("split16", "strict"):
                    if k in d:
                        f.write(f"`{k}` (unprofiled probe of the same code): `" + json.dumps(d[k]) + "`\n\n")
            f.write(f"| kernel | calls | total ms | ms / tile decode ({tiles}) | avg us | % |\n|---|---|---|---|---|---|\n")
            for r in rows[:24]:
                t = float(r["TotalDurationNs"])
                f.write(f"| `{r['Name'][:100]}` | {r['Calls']} | {t / 1e6:.2f} | {t / 1e6 / tiles:.2f} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.1f} |\n")
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
