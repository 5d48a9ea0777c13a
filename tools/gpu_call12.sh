#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/c12
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -k "gemm" 2>&1 | tail -6 | tee $O/tests.log
timeout 300 python tools/gemm_probe.py st4 2>&1 | grep -v amdgpu.ids | tee $O/gemm_probe_st4.log
