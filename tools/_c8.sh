cd $GRAFT_REPO_ROOT
(timeout 400 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "chained or t2i_fold or upscale" 2>&1 | tail -25) > gpurun_out/c8_test.log 2>&1
timeout 200 python tools/chain_bench.py > gpurun_out/c8_chain.log 2>&1
timeout 100 python tools/chain_ablation.py 2>&1 | grep -v amdgpu.ids | head -12 > gpurun_out/c8_ablation.log
MSAM_TUNE="dec_chain=0" timeout 200 python bench.py --no-cpu-baseline > gpurun_out/c8_bench_staged.log 2> gpurun_out/c8_bench_staged.err
timeout 200 python bench.py --no-cpu-baseline > gpurun_out/c8_bench_chain.log 2> gpurun_out/c8_bench_chain.err
tail -6 gpurun_out/c8_test.log; cat gpurun_out/c8_chain.log gpurun_out/c8_ablation.log
python - <<'PY'
import json
for f in ("c8_bench_staged", "c8_bench_chain"):
    try:
        d = json.loads(open("gpurun_out/" + f + ".log").read().strip().splitlines()[-1])
        r = d["roofline"]
        print(f, d["value"], d["ms_per_step"], d["config"].get("pipelined_labels_equal_serial"), d["config"].get("instances_per_tile"))
        for o in [r] + r["other_kernels"]:
            print("   ", o["kernel"][:40], o["launches"], o["seconds_per_tile"], o["avg_launch_us"])
    except Exception as e:
        print(f, "failed", e)
PY
tail -3 gpurun_out/c8_bench_chain.err
