#!/bin/bash
# Round 3, GPU call 2: the tests touched since call 1, host profiles of the API-level paths, kernel profile of a training step, RCCL via bench
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/c2
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_segment.py tests/test_gpu_rccl_smoke.py "tests/test_gpu_model.py::test_config1_vit_t_plumbing" "tests/test_gpu_model.py::test_amg_initialize_generate_vs_oracle" "tests/test_gpu_model.py::test_tiled_amg_vs_oracle" "tests/test_gpu_model.py::test_amg_crop_layers" "tests/test_gpu_model.py::test_config3_vit_l_tiled_volume_segment_slices" tests/test_gpu_parity_iou.py -m gpu -q -x > $O/tests.log 2>&1
tail -12 $O/tests.log
for w in api config3 train; do timeout 400 python tools/host_profile.py $w > $O/prof_$w.log 2>&1; head -3 $O/prof_$w.log | tail -2; done
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/train_prof -- python $R/tools/train_bench.py --model vit_b --freeze image_encoder prompt_encoder --steps 1 --warmup 1 > $R/$O/train_prof.log 2>&1
cd $R
python - <<PY
import csv, glob
f = glob.glob("$O/train_prof/*/*_kernel_stats.csv")
if f:
    rows = list(csv.DictReader(open(f[0])))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("train kernels total ms", tot / 1e6)
    for r in rows[:22]:
        print(f"{r['Name'][:80]:80s} {r['Calls']:>6s} {float(r['TotalDurationNs'])/1e6:9.2f} ms {float(r['AverageNs'])/1e3:9.1f} us {r['Percentage']}")
PY
find $O/train_prof -type f -size +4M -delete
MSAM_FORCE_DIST=1 timeout 300 python -X faulthandler bench.py --no-cpu-baseline --no-side --steps 2 --lanes 1 > $O/bench_forcedist.log 2> $O/bench_forcedist.err; echo "forcedist rc=$?"; tail -c 400 $O/bench_forcedist.log; tail -5 $O/bench_forcedist.err
timeout 300 python bench.py --no-cpu-baseline > $O/bench.log 2> $O/bench.err
python - <<PY
import json
d = json.loads(open("$O/bench.log").read().strip().splitlines()[-1])
print("bench", d["value"], d.get("api_inclusive"), d.get("pcie_inclusive"))
PY
