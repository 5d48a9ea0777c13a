#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/c14
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -k "attention" 2>&1 | tail -4 | tee $O/tests.log
timeout 300 python tools/attn_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/attn_probe.log
timeout 600 python -m pytest tests/test_gpu_model.py -q -x -k "encoder" 2>&1 | tail -4 | tee -a $O/tests.log
