cd $GRAFT_REPO_ROOT
(timeout 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_parity_iou.py tests/test_gpu_kernels.py -x -q -s -m gpu -k "encoder_vs_oracle or fp16_encoder or vit_attention or gemm" 2>&1 | grep -v "^$" | tail -30) > gpurun_out/c11_test.log 2>&1
timeout 60 python tools/hbm_probe.py > gpurun_out/c11_hbm.log 2>&1
timeout 200 python bench.py --no-cpu-baseline --encoder-dtype fp16 > gpurun_out/c11_bench_fp16.log 2> gpurun_out/c11_bench_fp16.err
timeout 200 python bench.py --no-cpu-baseline > gpurun_out/c11_bench_bf16.log 2> gpurun_out/c11_bench_bf16.err
cat gpurun_out/c11_test.log | cut -c1-1200; cat gpurun_out/c11_hbm.log
python - <<'PY'
import json
for f in ("c11_bench_bf16", "c11_bench_fp16"):
    try:
        d = json.loads(open("gpurun_out/" + f + ".log").read().strip().splitlines()[-1])
        r = d["roofline"]
        print(f, d["value"], d["ms_per_step"], d["dtype"][:40], d["config"].get("instances_per_tile"))
        for o in [r] + r["other_kernels"]:
            print("   ", o["kernel"][:40], o["launches"], o["seconds_per_tile"], o["avg_launch_us"], o["frac"])
    except Exception as e:
        print(f, "failed", e)
PY
tail -3 gpurun_out/c11_bench_fp16.err
