#!/bin/bash
# round 4, call 3: one-pass LN up_fused, pipelined segment_slices (API path), comm-stream gather (world-1 nccl), config sides
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -k "segment_slices or precompute_3d or vit_t or amg_initialize" 2>&1 | tail -5
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "upscale" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_parity_iou.py -x -q 2>&1 | tail -3
MSAM_FORCE_DIST=1 timeout 300 python bench.py --no-cpu-baseline --no-side --steps 2 > gpurun_out/r4_3_dist.log 2> gpurun_out/r4_3_dist.err; tail -c 300 gpurun_out/r4_3_dist.err; python -c "
import json; d=json.loads(open('gpurun_out/r4_3_dist.log').read().strip().splitlines()[-1]); print('force_dist', d['value'], d['config']['parallelism'])"
timeout 1500 python bench.py --no-cpu-baseline > gpurun_out/r4_3_bench.log 2> gpurun_out/r4_3_bench.err; tail -c 600 gpurun_out/r4_3_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4_3_bench.log').read().strip().splitlines()[-1])
print('value', d['value'], 'up_fused us', d['roofline'].get('avg_launch_us'), 'frac', d['roofline']['frac'])
for k in ('pcie_inclusive','api_inclusive','fp8_side','config3_side','train_side'):
    print(k, json.dumps(d.get(k))[:700])
PY
