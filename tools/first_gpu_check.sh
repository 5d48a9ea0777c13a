#!/bin/bash
# First GPU call of a round: the -m gpu tests that were written without GPU access at the end of round 2 (prompt generators,
# prompt-based segmentation), then the whole -m gpu suite.  bash tools/first_gpu_check.sh   (through gpurun, ~10 min)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
MSAM_RUN_PENDING=1 timeout 1500 python -m pytest tests/test_gpu_zz_prompt_generator.py tests/test_gpu_zz_prompt_based_segmentation.py tests/test_gpu_zz_training_encoders.py -m gpu -q -rA \
    > gpurun_out/first_new_tests.log 2>&1
tail -15 gpurun_out/first_new_tests.log
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/first_full_suite.log 2>&1
tail -5 gpurun_out/first_full_suite.log
