#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/c18
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_postprocess.py -q 2>&1 | tail -3 | tee $O/tests.log
timeout 200 python tools/pp_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/pp_probe.log
