#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/c8
mkdir -p $O
run() { name=$1; shift; timeout 300 python bench.py --no-cpu-baseline --no-side "$@" > $O/$name.log 2> $O/$name.err; python - <<PY
import json
try:
    d = json.loads(open("$O/$name.log").read().strip().splitlines()[-1]); print("$name", d["value"], d["ms_per_step"], d["config"]["pipelined_labels_equal_serial"])
except Exception as e: print("$name", "missing", e)
PY
}
run t16_e16_l3 --tiles-per-step 16 --enc-batch 16 --lanes 3 --steps 4
run t32_e16_l3 --tiles-per-step 32 --enc-batch 16 --lanes 3 --steps 2
run t64_e16_l3 --tiles-per-step 64 --enc-batch 16 --lanes 3 --steps 2
run t16_e8_l3 --tiles-per-step 16 --enc-batch 8 --lanes 3 --steps 4
run t32_e16_l4 --tiles-per-step 32 --enc-batch 16 --lanes 4 --steps 2
run t32_e16_l2 --tiles-per-step 32 --enc-batch 16 --lanes 2 --steps 2
run t16_e16_l3_again --tiles-per-step 16 --enc-batch 16 --lanes 3 --steps 4
timeout 400 python bench.py --workload config3 --steps 2 --warmup 1 --slices 4 > $O/config3.log 2> $O/config3.err; tail -c 300 $O/config3.log
