#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/c11
mkdir -p $O
timeout 400 python -m pytest tests/test_gpu_kernels.py -q -k "gemm" 2>&1 | tail -8 | tee $O/tests.log
timeout 500 python tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/gemm_bench.log
