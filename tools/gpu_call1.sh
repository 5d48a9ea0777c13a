#!/bin/bash
# Round 3, GPU call 1: the whole -m gpu suite on the new code (split sites, unwrapped suites, RCCL smoke, sensitivity parity cases), a first
# bench line, timing experiments that decide the perf work (GEMM time split, GELU without its exponentials), the config-3 line, RCCL
# through bench.py, fine-tuning steps/s.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/c1
mkdir -p $O
rm -f gpurun_out/parity_reports.json
timeout 1200 python -m pytest tests -m gpu -q --durations=12 > $O/suite.log 2>&1
tail -25 $O/suite.log
timeout 400 python bench.py --no-cpu-baseline > $O/bench.log 2> $O/bench.err; tail -c 1500 $O/bench.log
for g in 2 3; do MSAM_TUNE="up_gelu16=$g" timeout 300 python bench.py --no-cpu-baseline --no-side --lanes 1 > $O/bench_gelu$g.log 2> $O/bench_gelu$g.err; done
timeout 300 python bench.py --no-cpu-baseline --no-side --lanes 1 > $O/bench_lanes1.log 2> $O/bench_lanes1.err
timeout 300 python tools/gemm_bench.py > $O/gemm_bench.log 2>&1; tail -8 $O/gemm_bench.log
MSAM_FORCE_DIST=1 timeout 300 python bench.py --no-cpu-baseline --no-side --steps 2 > $O/bench_forcedist.log 2> $O/bench_forcedist.err
timeout 400 python bench.py --workload config3 --steps 1 --warmup 1 --slices 2 > $O/config3.log 2> $O/config3.err; tail -c 600 $O/config3.log
timeout 300 python tools/train_bench.py --model vit_b --freeze image_encoder prompt_encoder --steps 3 > $O/train_b_dec.log 2>&1; tail -2 $O/train_b_dec.log
timeout 400 python tools/train_bench.py --model vit_b --steps 2 > $O/train_b_full.log 2>&1; tail -2 $O/train_b_full.log
python - <<PY
import json
for f in ("bench", "bench_gelu2", "bench_gelu3", "bench_lanes1", "bench_forcedist"):
    try:
        d = json.loads(open("$O/" + f + ".log").read().strip().splitlines()[-1])
        print(f, d["value"], d["roofline"]["kernel"][:20], d["roofline"]["avg_launch_us"], [(k["kernel"][:14], k["avg_launch_us"]) for k in d["roofline"]["other_kernels"]], d.get("api_inclusive"))
    except Exception as e:
        print(f, "missing", e)
PY
