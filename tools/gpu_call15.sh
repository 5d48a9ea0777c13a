#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/c15
mkdir -p $O
timeout 1000 python -m pytest tests -m gpu -q 2>&1 | tail -15 | tee $O/suite.log
bash tools/final_measure.sh 2>&1 | tail -30 | tee $O/final.log
