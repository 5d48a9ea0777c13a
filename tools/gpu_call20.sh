#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --no-cpu-baseline --no-side --steps 2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('bench', d['value'], 'traffic', r['traffic'], r['traffic_source'][:70])"
