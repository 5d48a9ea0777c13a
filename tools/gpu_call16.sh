#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
mkdir -p $O/c16
python tools/csrc_sha.py > $O/final_csrc_sha.txt
python bench.py > $O/final_bench.log 2> $O/final_bench.err
tail -1 $O/final_bench.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('bench', d['value'], d['ms_per_step'], 'traffic', r['traffic'], r.get('traffic_source','')[:80], 'api', d['api_inclusive']['value'], 'iou', d.get('mask_iou_vs_ref',{}).get('frac_ge_0.999'))"
python bench.py --no-cpu-baseline --no-side > $O/c16/bench2.log 2>/dev/null; tail -1 $O/c16/bench2.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bench repeat', d['value'])"
timeout 600 python -m pytest tests/test_gpu_parity_iou.py -q 2>&1 | tail -3
