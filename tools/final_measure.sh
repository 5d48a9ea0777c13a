#!/bin/bash
# Round-end measurement bundle, run on the GPU box through gpurun:  bash tools/final_measure.sh
# Outputs under gpurun_out/ are turned into profiles/<tag>_* by tools/summarize_profiles.py.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
rm -rf $O/final_prof $O/final_fetch $O/final_write
python $R/bench.py > $O/final_bench.log 2> $O/final_bench.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/final_prof -- python $R/bench.py --no-cpu-baseline --lanes 1 > $O/final_prof.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/final_fetch -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/final_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/final_write -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/final_write.log 2>&1
python $R/bench.py --encoder-dtype fp8 --no-cpu-baseline > $O/final_fp8.log 2> $O/final_fp8.err
python $R/bench.py --encoder-dtype fp16 --no-cpu-baseline > $O/final_fp16.log 2> $O/final_fp16.err
python $R/bench.py --lanes 1 --no-cpu-baseline > $O/final_lanes1.log 2> $O/final_lanes1.err
MSAM_TUNE="dec_chain=0" python $R/bench.py --no-cpu-baseline > $O/final_staged.log 2> $O/final_staged.err
python $R/tools/hbm_probe.py > $O/final_hbm_probe.log 2>&1
cd $R && python -c "import __graft_entry__ as g; g.smoke()" > $O/final_smoke.log 2>&1
tail -1 $O/final_smoke.log
python - <<PY
import json
for f in ("final_bench", "final_fp8", "final_fp16", "final_lanes1", "final_staged"):
    d = json.loads(open("$O/" + f + ".log").read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], d.get("cpu_baseline"))
PY
cat $O/final_hbm_probe.log
find $O/final_prof $O/final_fetch $O/final_write -type f -size +6M -delete
ls $O/final_prof/*/ | head; du -sh $O
