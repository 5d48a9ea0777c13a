#!/bin/bash
# round 4, call: split token MLP - decoder tests, parity suites (designed + fine-tuned weights), bench line
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -k "decoder or amg_initialize or set_image" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_parity_iou.py -x -q 2>&1 | tail -3
timeout 900 python tools/trained_parity.py --steps 100 --thresholds 0.5 0.8 --ablations 2>/dev/null | tee gpurun_out/trained_parity4.log | cut -c1-1800
timeout 900 python tools/logit_error_probe.py 2>/dev/null | tee gpurun_out/probe4b.log
timeout 900 python bench.py --no-cpu-baseline --no-config-sides --steps 2 > gpurun_out/r4_5_bench.log 2> gpurun_out/r4_5_bench.err; tail -c 300 gpurun_out/r4_5_bench.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r4_5_bench.log').read().strip().splitlines() if l.startswith('{')][-1])
print('value', d['value'], 'up_fused us', d['roofline'].get('avg_launch_us'))
a=d.get('api_inclusive'); print('api', a.get('value'), a.get('labels_equal_literal_loop'), a['literal_loop']['value']); print('pcie', d['pcie_inclusive']['value'])
for k in d['roofline']['other_kernels']: print(k['kernel'][:30], k['launches'], k['avg_launch_us'], k['seconds_per_tile'])
PY
