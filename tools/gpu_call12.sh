#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/c12
mkdir -p $O
timeout 300 python tools/gemm_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/gemm_probe.log
