#!/bin/bash
# First GPU call of the next round (prepared at the end of round 3, when the GPU budget was spent): what do up_fused_kernel's waves wait for?
#   gpurun --timeout 900 -- 'bash tools/gpu_call_r4_first.sh'
# 1. tools/uf_lab.py: the shipped kernel against its timing-only variants (barrier / LayerNorm exchange / GELUs / stage-1 MFMAs / output path
#    taken out one at a time, one workgroup per CU) and the two verified candidates (permlane sums: bit-identical; one-pass LayerNorm).
# 2. the same launch under the SQ counters that separate issue from waiting (own pass, no tracing beside it).
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python tools/uf_lab.py 2>&1 | tee gpurun_out/uf_lab.log | tail -16
cd /tmp
timeout 240 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY \
    -d $OLDPWD/gpurun_out/uf_lab_pmc -o pmc --output-format csv -- python $OLDPWD/tools/uf_lab.py --only base --launches 3 > $OLDPWD/gpurun_out/uf_lab_pmc.log 2>&1
cd $OLDPWD
python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/uf_lab_pmc/**/*counter_collection.csv", recursive=True):
    acc = {}
    for r in csv.DictReader(open(f)):
        if "up_fused" in r.get("Kernel_Name", ""):
            acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print(f"{k:28s} {sum(v) / len(v):.4g} per launch ({len(v)} launches)")
PY
