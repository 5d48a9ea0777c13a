cd $GRAFT_REPO_ROOT
for G in 0 1; do
  (MSAM_TUNE="up_gelu16=$G" timeout 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity_iou.py -x -q -s -m gpu -k "upscale_fused or per_instance or scores_close" 2>&1 | grep -v "^$" | tail -12) > gpurun_out/c10_test_g$G.log 2>&1
done
timeout 100 python tools/upfused_bench.py > gpurun_out/c10_upfused.log 2>&1
MSAM_TUNE="up_gelu16=1" timeout 200 python bench.py --no-cpu-baseline > gpurun_out/c10_bench_g1.log 2> gpurun_out/c10_bench_g1.err
timeout 200 python bench.py --no-cpu-baseline > gpurun_out/c10_bench_g0.log 2> gpurun_out/c10_bench_g0.err
for G in 0 1; do echo "== gelu16=$G"; cat gpurun_out/c10_test_g$G.log | cut -c1-900; done
cat gpurun_out/c10_upfused.log
python - <<'PY'
import json
for f in ("c10_bench_g0", "c10_bench_g1"):
    try:
        d = json.loads(open("gpurun_out/" + f + ".log").read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d["config"].get("instances_per_tile"))
    except Exception as e:
        print(f, "failed", e)
PY
