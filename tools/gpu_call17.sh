#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/c17
mkdir -p $O
python bench.py --workload config3 --steps 1 --warmup 1 --slices 4 > gpurun_out/final_config3.log 2> gpurun_out/final_config3.err; tail -1 gpurun_out/final_config3.log | cut -c1-200
python bench.py --workload config3 --steps 2 --warmup 1 --slices 8 2>/dev/null | tail -1 | cut -c1-200
python bench.py --no-cpu-baseline > $O/bench.log 2>/dev/null; tail -1 $O/bench.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bench', d['value'], 'api', d['api_inclusive'], 'pcie', d['pcie_inclusive']['value'])" | cut -c1-600
timeout 300 python -m pytest tests/test_gpu_model.py -q -k "zarr or precompute or raw" 2>&1 | tail -2
