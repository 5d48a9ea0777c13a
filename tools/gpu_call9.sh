#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/c9
mkdir -p $O
timeout 300 python bench.py --no-cpu-baseline > $O/bench.log 2> $O/bench.err
python - <<PY
import json
d = json.loads(open("$O/bench.log").read().strip().splitlines()[-1]); print("bench", d["value"], d["ms_per_step"], d["config"]["tiles_per_step_per_gpu"], d["api_inclusive"]["value"], d["roofline"]["traffic"], d["roofline"]["traffic_source"][:60])
PY
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/train_full -- python $R/tools/train_bench.py --model vit_b --steps 1 --warmup 1 > $R/$O/train_full.log 2>&1
cd $R
python - <<PY
import csv, glob
f = glob.glob("$O/train_full/*/*_kernel_stats.csv")
if f:
    rows = list(csv.DictReader(open(f[0])))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("train (vit_b whole model) kernels total ms over 2 steps", tot / 1e6, "launches", sum(int(r["Calls"]) for r in rows))
    for r in rows[:22]:
        print(f"{r['Name'][:90]:90s} {r['Calls']:>6s} {float(r['TotalDurationNs'])/1e6:9.2f} ms {float(r['AverageNs'])/1e3:9.1f} us {r['Percentage']}")
PY
find $O/train_full -type f -size +4M -delete
