cd $GRAFT_REPO_ROOT
(timeout 400 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "chained" 2>&1 | tail -4) > gpurun_out/c14_test.log 2>&1
timeout 100 python tools/chain_bench.py 2>&1 | grep "second form\|v2 att" > gpurun_out/c14_chain.log
(MSAM_TUNE="chain_variant=9" timeout 300 python -m pytest tests/test_gpu_parity_iou.py -x -q -s -m gpu -k "per_instance_iou and not fp16" 2>&1 | grep "mask_iou_vs_ref\|passed\|failed" | cut -c1-300) > gpurun_out/c14_parity.log 2>&1
MSAM_TUNE="chain_variant=9" timeout 200 python bench.py --no-cpu-baseline > gpurun_out/c14_bench_v9.log 2> gpurun_out/c14_bench_v9.err
MSAM_TUNE="chain_variant=0" timeout 200 python bench.py --no-cpu-baseline > gpurun_out/c14_bench_v0.log 2> gpurun_out/c14_bench_v0.err
timeout 200 python bench.py --no-cpu-baseline > gpurun_out/c14_bench_v6.log 2> gpurun_out/c14_bench_v6.err
MSAM_TUNE="chain_variant=9" timeout 200 python bench.py --no-cpu-baseline > gpurun_out/c14_bench_v9b.log 2> gpurun_out/c14_bench_v9b.err
cat gpurun_out/c14_test.log gpurun_out/c14_chain.log gpurun_out/c14_parity.log
python - <<'PY'
import json
for f in ("c14_bench_v9", "c14_bench_v0", "c14_bench_v6", "c14_bench_v9b"):
    try:
        d = json.loads(open("gpurun_out/" + f + ".log").read().strip().splitlines()[-1])
        r = d["roofline"]
        print(f, d["value"], d["ms_per_step"], d["config"].get("instances_per_tile"))
        for o in [r] + r["other_kernels"]:
            if "i2t0" in o["kernel"] or "i2t01" in o["kernel"]: print("   ", o["kernel"][:40], o["launches"], o["seconds_per_tile"], o["avg_launch_us"])
    except Exception as e:
        print(f, "failed", e)
PY
