#!/bin/bash
# Round 3, GPU call 5: token-side launches fused (product + LayerNorm + operand copies per launch): decoder parity tests, A/B in the bench
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/c5
mkdir -p $O
timeout 900 python -m pytest "tests/test_gpu_kernels.py" -k "gemm" -m gpu -q > $O/tests_gemm.log 2>&1; tail -3 $O/tests_gemm.log
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_modules.py tests/test_gpu_parity_iou.py -m gpu -q -k "decoder or decode or amg or parity or per_instance or sensitivity or identical or predictor or batched or module" > $O/tests_dec.log 2>&1; tail -6 $O/tests_dec.log
for f in 1 0; do for l in 1 3; do
MSAM_TUNE="tok_fuse=$f" timeout 300 python bench.py --no-cpu-baseline --no-side --lanes $l --steps 4 > $O/bench_f${f}_l$l.log 2> $O/bench_f${f}_l$l.err
done; done
python - <<PY
import json
for f in (1, 0):
    for l in (1, 3):
        try:
            d = json.loads(open("$O/bench_f%d_l%d.log" % (f, l)).read().strip().splitlines()[-1])
            ks = {k["kernel"][:12]: (k["launches"], k["avg_launch_us"]) for k in [d["roofline"]] + d["roofline"]["other_kernels"]}
            print("tok_fuse", f, "lanes", l, d["value"], ks.get("gemm_kernel "), ks.get("up_fused_ker"))
        except Exception as e:
            print(f, l, "missing", e)
PY
