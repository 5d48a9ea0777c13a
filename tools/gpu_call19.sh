#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/c19
mkdir -p $O
bash tools/final_measure.sh 2>&1 | tail -22 | tee $O/final.log
timeout 420 python -m pytest tests/test_gpu_model.py tests/test_gpu_modules.py tests/test_gpu_segment.py tests/test_gpu_parity_iou.py -q -x 2>&1 | tail -5 | tee $O/tests.log
