#!/bin/bash
# Round 3, GPU call 3: training speed-ups (row-block attention, split-K weight gradients), API-path / config-3 host fixes, vit_t plumbing
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/c4
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_training.py tests/test_gpu_training_encoders.py "tests/test_gpu_kernels.py::test_gemm_split_k_matches_the_plain_product" "tests/test_gpu_kernels.py::test_gemm_plain" "tests/test_gpu_model.py::test_config1_vit_t_plumbing" "tests/test_gpu_model.py::test_amg_initialize_generate_vs_oracle" "tests/test_gpu_model.py::test_tiled_amg_vs_oracle" "tests/test_gpu_model.py::test_amg_crop_layers" "tests/test_gpu_model.py::test_precompute_3d_batched" "tests/test_gpu_model.py::test_zarr_cache_gpu" "tests/test_gpu_model.py::test_config3_vit_l_tiled_volume_segment_slices" tests/test_gpu_segment.py -m gpu -q > $O/tests.log 2>&1
tail -8 $O/tests.log
for w in api config3 train; do timeout 400 python tools/host_profile.py $w > $O/prof_$w.log 2>&1; grep -m1 -E "api path|config3:|train step" $O/prof_$w.log; done
timeout 300 python tools/train_bench.py --model vit_b --freeze image_encoder prompt_encoder --steps 3 > $O/train_b_dec.log 2>&1; tail -1 $O/train_b_dec.log
timeout 400 python tools/train_bench.py --model vit_b --steps 2 > $O/train_b_full.log 2>&1; tail -1 $O/train_b_full.log
timeout 600 python tools/train_bench.py --model vit_h --steps 2 > $O/train_h_full.log 2>&1; tail -1 $O/train_h_full.log
MSAM_FORCE_DIST=1 timeout 300 python -X faulthandler bench.py --no-cpu-baseline --no-side --steps 2 > $O/bench_forcedist3.log 2> $O/bench_forcedist3.err; echo "forcedist lanes3 rc=$?"; tail -c 300 $O/bench_forcedist3.log; tail -8 $O/bench_forcedist3.err
timeout 300 python bench.py --no-cpu-baseline > $O/bench.log 2> $O/bench.err
timeout 400 python bench.py --workload config3 --steps 1 --warmup 1 --slices 2 > $O/config3.log 2> $O/config3.err
python - <<PY
import json
d = json.loads(open("$O/bench.log").read().strip().splitlines()[-1])
print("bench", d["value"], d.get("api_inclusive"), d.get("pcie_inclusive"))
d = json.loads(open("$O/config3.log").read().strip().splitlines()[-1])
print("config3", d["value"], d["config"]["tiles_per_second"])
PY
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
    print("train kernels total ms (2 steps incl. warm-up)", tot / 1e6)
    for r in rows[:16]:
        print(f"{r['Name'][:80]:80s} {r['Calls']:>6s} {float(r['TotalDurationNs'])/1e6:9.2f} ms {float(r['AverageNs'])/1e3:9.1f} us {r['Percentage']}")
PY
find $O/train_prof -type f -size +4M -delete
