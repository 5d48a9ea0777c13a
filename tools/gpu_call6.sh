#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/c6
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_postprocess.py "tests/test_gpu_model.py::test_token_side_fused_launches_give_the_same_decode" "tests/test_gpu_model.py::test_config1_vit_t_plumbing" "tests/test_gpu_model.py::test_tiled_amg_vs_oracle" "tests/test_gpu_model.py::test_config3_vit_l_tiled_volume_segment_slices" "tests/test_gpu_model.py::test_zarr_cache_gpu" -m gpu -q > $O/tests.log 2>&1; tail -6 $O/tests.log
timeout 400 python tools/host_profile.py config3 > $O/prof_config3.log 2>&1; grep -m1 "config3:" $O/prof_config3.log
timeout 400 python bench.py --workload config3 --steps 2 --warmup 1 --slices 4 > $O/config3.log 2> $O/config3.err; tail -c 700 $O/config3.log
