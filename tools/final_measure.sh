#!/bin/bash
# Round-end measurement bundle, run on the GPU box through gpurun:  bash tools/final_measure.sh [quick]
# Outputs under gpurun_out/ are turned into profiles/<tag>_* by tools/summarize_profiles.py (+ tools/pmc_summary.py for the SQ passes).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
rm -rf $O/final_prof $O/final_fetch $O/final_write $O/final_sq1 $O/final_sq2 $O/final_strict_prof
python $R/tools/csrc_sha.py > $O/final_csrc_sha.txt
python $R/bench.py > $O/final_bench.log 2> $O/final_bench.err
cp $O/bench_extras.json $O/final_bench_extras.json 2>/dev/null
# rocprof passes: hot path only (--no-side), one decode lane (kernels of different tiles do not overlap)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/final_prof -- python $R/bench.py --no-cpu-baseline --no-side --lanes 1 > $O/final_prof.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/final_fetch -- python $R/bench.py --steps 1 --warmup 1 --tiles-per-step 16 --distinct-tiles 16 --no-cpu-baseline --no-side --lanes 1 > $O/final_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/final_write -- python $R/bench.py --steps 1 --warmup 1 --tiles-per-step 16 --distinct-tiles 16 --no-cpu-baseline --no-side --lanes 1 > $O/final_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $O/final_sq1 -- python $R/tools/pmc_tile.py > $O/final_sq1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU_TRANS_F32 --output-format csv -d $O/final_sq2 -- python $R/tools/pmc_tile.py > $O/final_sq2.log 2>&1
python $R/tools/pmc_summary.py $O/final_sq1 $O/final_sq2 > $O/final_sq_table.md 2>&1
if [ "$1" != "quick" ]; then
# (round 4: configs 3 / 4 / 5 are side fields of the default bench line itself - fp8_side, config3_side, train_side)
python $R/bench.py --lanes 1 --no-cpu-baseline --no-side > $O/final_lanes1.log 2> $O/final_lanes1.err
fi
# the strict precision mode: its own kernel table (two tiles through the per-tile API loop in both modes + the f32-input MFMA product on the path's shapes)
rm -rf $O/final_strict_prof
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/final_strict_prof -- python $R/tools/strict_probe.py --out $O/final_strict_probe.json > $O/final_strict_prof.log 2>&1
# the split16 precision mode (round 6): kernel table of the per-tile API loop + its probe line (masks vs the strict mode's, segment_slices rate)
rm -rf $O/final_split16_prof
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/final_split16_prof -- python $R/tools/split16_probe.py --modes split16 --no-products --slice-tiles 0 > $O/final_split16_prof.log 2>&1
python $R/tools/split16_probe.py --modes split16,strict --out $O/final_split16_probe.json > $O/final_split16_probe.log 2>&1
python $R/tools/hbm_probe.py > $O/final_hbm_probe.log 2>&1
cd $R && python -c "import __graft_entry__ as g; g.smoke()" > $O/final_smoke.log 2>&1
tail -1 $O/final_smoke.log
python - <<PY
import json
for f in ("final_bench", "final_lanes1"):
    try:
        d = json.loads(open("$O/" + f + ".log").read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d.get("cpu_baseline"), d.get("api_inclusive"))
    except Exception as e:
        print(f, "missing", e)
PY
cat $O/final_hbm_probe.log
find $O/final_prof $O/final_fetch $O/final_write $O/final_sq1 $O/final_sq2 $O/final_strict_prof $O/final_split16_prof -type f -size +6M -delete
ls $O/final_prof/*/ | head; du -sh $O
