cd $GRAFT_REPO_ROOT
(timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -12) > gpurun_out/c15_test.log 2>&1
timeout 100 python tools/chain_bench.py 2>&1 | grep "second form\|v2 att" > gpurun_out/c15_chain.log
MSAM_TUNE="chain_variant=9" timeout 200 python bench.py --no-cpu-baseline > gpurun_out/c15_bench_v9.log 2> gpurun_out/c15_bench_v9.err
timeout 200 python bench.py --no-cpu-baseline > gpurun_out/c15_bench_v6.log 2> gpurun_out/c15_bench_v6.err
MSAM_TUNE="chain_variant=9" timeout 200 python bench.py --no-cpu-baseline > gpurun_out/c15_bench_v9b.log 2> gpurun_out/c15_bench_v9b.err
cat gpurun_out/c15_test.log gpurun_out/c15_chain.log
python - <<'PY'
import json
for f in ("c15_bench_v9", "c15_bench_v6", "c15_bench_v9b"):
    try:
        d = json.loads(open("gpurun_out/" + f + ".log").read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d["config"].get("instances_per_tile"))
    except Exception as e:
        print(f, "failed", e)
PY
