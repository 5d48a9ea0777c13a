"""Per-kernel summary of a rocprofv3 --kernel-trace run (rocpd sqlite output): launches, total / average duration, share
of the GPU time and time per tile (tiles = launches of up_fused_kernel, one per tile decode).

    python tools/rocprof_db_summary.py gpurun_out/<dir>/prof/bench_results.db [> profiles/rNN_bench_kernel_summary.md]
"""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = list(cur.execute(f"select s.kernel_name, count(*), sum(d.end - d.start), min(d.end - d.start), max(d.end - d.start), "
                            f"max(s.arch_vgpr_count), max(s.accum_vgpr_count), max(s.group_segment_size) "
                            f"from {kd} d join {ks} s on d.kernel_id = s.id group by s.kernel_name order by 3 desc"))
    total = sum(r[2] for r in rows)
    tiles = max([r[1] for r in rows if "up_fused_kernel" in r[0]] or [1])

    def short(n):
        n = re.sub(r"\(anonymous namespace\)::", "", n)
        n = re.sub(r"^void ", "", n)
        return n if len(n) <= 90 else n[:87] + "..."
    print(f"rocprofv3 --kernel-trace summary: {sum(r[1] for r in rows)} launches, {total / 1e6:.1f} ms of kernel time, "
          f"{tiles} tiles (launches of up_fused_kernel), {total / 1e6 / tiles:.3f} ms of kernel time per tile\n")
    print("| kernel | launches | per tile | total ms | avg us | min us | max us | % | us / tile | VGPR+AGPR | LDS B |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    for name, n, tot, mn, mx, vg, ag, lds in rows[:45]:
        print(f"| `{short(name)}` | {n} | {n / tiles:.2f} | {tot / 1e6:.2f} | {tot / n / 1e3:.1f} | {mn / 1e3:.1f} | {mx / 1e3:.1f} | "
              f"{100.0 * tot / total:.1f} | {tot / 1e3 / tiles:.1f} | {vg}+{ag} | {lds} |")
    rest = rows[45:]
    if rest:
        print(f"| ({len(rest)} more kernels) | {sum(r[1] for r in rest)} | | {sum(r[2] for r in rest) / 1e6:.2f} | | | | "
              f"{100.0 * sum(r[2] for r in rest) / total:.1f} | {sum(r[2] for r in rest) / 1e3 / tiles:.1f} | | |")


if __name__ == "__main__":
    main()
