cd $GRAFT_REPO_ROOT
(timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -30) > gpurun_out/c9_test.log 2>&1
for L in 2 4 5; do timeout 200 python bench.py --no-cpu-baseline --lanes $L > gpurun_out/c9_bench_l$L.log 2> gpurun_out/c9_bench_l$L.err; done
tail -30 gpurun_out/c9_test.log
python - <<'PY'
import json
for f in ("c9_bench_l2", "c9_bench_l4", "c9_bench_l5"):
    try:
        d = json.loads(open("gpurun_out/" + f + ".log").read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d["config"].get("pipelined_labels_equal_serial"))
    except Exception as e:
        print(f, "failed", e)
PY
