#!/bin/bash
# round 4, call 4: API path with cached lanes, mask-only prompts, segment_slices tests
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -k "set_image or lora or amg_initialize" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -k "segment_slices" 2>&1 | tail -3
timeout 900 python bench.py --no-cpu-baseline --no-config-sides --steps 2 > gpurun_out/r4_4_bench.log 2> gpurun_out/r4_4_bench.err; tail -c 400 gpurun_out/r4_4_bench.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r4_4_bench.log').read().strip().splitlines() if l.startswith('{')][-1])
print('value', d['value'], 'up_fused us', d['roofline'].get('avg_launch_us'))
a=d.get('api_inclusive'); print('api', a.get('value'), a.get('labels_equal_literal_loop'), json.dumps(a.get('literal_loop'))[:300]); print('pcie', d['pcie_inclusive']['value'])
PY
