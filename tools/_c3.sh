cd $GRAFT_REPO_ROOT
(timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "chained or i2t_fold_layer" 2>&1 | tail -25) > gpurun_out/c3_test.log 2>&1
timeout 120 python tools/chain_bench.py > gpurun_out/c3_chain.log 2>&1
MSAM_TUNE="dec_chain=0" timeout 200 python bench.py --no-cpu-baseline > gpurun_out/c3_bench_staged.log 2> gpurun_out/c3_bench_staged.err
timeout 200 python bench.py --no-cpu-baseline > gpurun_out/c3_bench_chain.log 2> gpurun_out/c3_bench_chain.err
tail -25 gpurun_out/c3_test.log; cat gpurun_out/c3_chain.log
python - <<'PY'
import json
for f in ("c3_bench_staged", "c3_bench_chain"):
    try:
        d = json.loads(open("gpurun_out/" + f + ".log").read().strip().splitlines()[-1])
        r = d["roofline"]
        print(f, d["value"], d["ms_per_step"], d.get("mask_iou_vs_ref"))
        for o in [r] + r["other_kernels"]:
            print("   ", o["kernel"][:40], o["launches"], o["seconds_per_tile"], o["avg_launch_us"])
    except Exception as e:
        print(f, "failed", e)
PY
tail -3 gpurun_out/c3_bench_chain.err
