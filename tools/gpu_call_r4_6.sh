#!/bin/bash
# round 4: fp16 low-res hand-over (up_fused -> postprocess), short first batch in the slice pipeline; postprocess / AMG / parity tests + bench
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_postprocess.py tests/test_gpu_kernels.py -x -q -k "postprocess or upscale or rle" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -k "amg or segment_slices or encoder_bits or tiled or batched_inference" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_parity_iou.py tests/test_gpu_parity_trained.py -x -q 2>&1 | tail -4
timeout 900 python bench.py --no-cpu-baseline --no-config-sides --steps 3 > gpurun_out/r4_6_bench.log 2> gpurun_out/r4_6_bench.err; tail -c 300 gpurun_out/r4_6_bench.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r4_6_bench.log').read().strip().splitlines() if l.startswith('{')][-1])
print('value', d['value'], 'up_fused us', d['roofline'].get('avg_launch_us'), 'frac', d['roofline']['frac'])
a=d.get('api_inclusive'); print('api', a.get('value'), a.get('labels_equal_literal_loop'), a['literal_loop']['value']); print('pcie', d['pcie_inclusive']['value'])
for k in d['roofline']['other_kernels']: print(k['kernel'][:30], k['launches'], k['avg_launch_us'], k['seconds_per_tile'])
PY
