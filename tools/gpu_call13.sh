#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/c13
mkdir -p $O
timeout 400 python tools/train_bench.py --model vit_b --op-profile 45 > $O/opprof.log 2>&1
grep -v amdgpu.ids $O/opprof.log | cut -c1-260 | head -120
