cd $GRAFT_REPO_ROOT
(timeout 60 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity_iou.py -x -q -m gpu -k "chained or (per_instance_iou and not fp16)" 2>&1 | tail -4) > gpurun_out/c17_test.log 2>&1
timeout 45 python bench.py --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/c17_bench.log 2> gpurun_out/c17_bench.err
cat gpurun_out/c17_test.log
python - <<'PY'
import json
d = json.loads(open("gpurun_out/c17_bench.log").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["config"].get("instances_per_tile"), d["config"].get("pipelined_labels_equal_serial"))
PY
