#!/bin/bash
# Round 3: the whole -m gpu suite, then the measurement bundle
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/c7
rm -f gpurun_out/parity_reports.json
timeout 1500 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/c7/suite.log 2>&1
tail -14 gpurun_out/c7/suite.log
bash tools/final_measure.sh
