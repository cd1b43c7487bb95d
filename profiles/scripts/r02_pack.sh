#!/bin/bash
cd /root/repo
for a in "" "--pack-tables" "--pack-tables --pack-min-vocab 500000" "--pack-tables --pack-min-vocab 90000"; do
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline $a 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('[$a] step_ms', round(d['ms_per_step'],4), 'fwd as-run', round(r['kernel_ms'],4), 'alone', round(r.get('kernel_ms_alone',0),4))"; done; done
