#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/c13
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_training.py tests/test_gpu_training_encoders.py -q -x 2>&1 | tail -4 | tee $O/tests.log
rm -f $O/train.log
for args in "--model vit_b --freeze image_encoder prompt_encoder" "--model vit_b" "--model vit_h"; do
  timeout 400 python tools/train_bench.py $args --steps 4 --warmup 2 2>&1 | tail -1 | tee -a $O/train.log
done
