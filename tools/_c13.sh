cd $GRAFT_REPO_ROOT
(timeout 400 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "chained" 2>&1 | tail -15) > gpurun_out/c13_test.log 2>&1
timeout 200 python tools/chain_bench.py > gpurun_out/c13_chain.log 2>&1
tail -12 gpurun_out/c13_test.log; cat gpurun_out/c13_chain.log
