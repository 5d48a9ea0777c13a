#!/bin/bash
# round 4, call 2: the shipped (pipelined) up_fused_kernel: lab ablations on the new base, its GPU tests, a quick bench line
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python tools/uf_lab.py 2>&1 | tee gpurun_out/uf_lab2.log | tail -12
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "upscale or chained or t2i or i2t" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_parity_iou.py -x -q 2>&1 | tail -3
timeout 300 python bench.py --no-cpu-baseline --no-side > gpurun_out/r4_2_bench.log 2> gpurun_out/r4_2_bench.err; tail -c 1500 gpurun_out/r4_2_bench.log
