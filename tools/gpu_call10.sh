#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/c10
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_training_encoders.py tests/test_gpu_training.py -q 2>&1 | tail -8 | tee $O/tests.log
for m in vit_b vit_h; do
  for impl in gemm kernel; do
    MSAM_RELPOS_IMPL=$impl timeout 400 python tools/train_bench.py --model $m --steps 3 --warmup 1 2>&1 | tail -2 | tee $O/train_${m}_${impl}.log
  done
done
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
    for r in rows[:25]:
        print(f"{r['Name'][:90]:90s} {r['Calls']:>6s} {float(r['TotalDurationNs'])/1e6:9.2f} ms {float(r['AverageNs'])/1e3:9.1f} us {r['Percentage']}")
PY
find $O/train_full -type f -size +4M -delete
