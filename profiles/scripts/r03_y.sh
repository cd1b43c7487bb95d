#!/bin/bash
cd /root/repo
for v in main fork main fork side; do
RECBOX_AMD_FM_REZERO_ON=$v timeout 600 python bench.py --no-cpu-baseline --no-extra-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('fm rezero=$v', round(d['ms_per_step'],4), round(d['roofline']['frac'],3))"
done
RECBOX_AMD_FM_REZERO_ON=fork timeout 900 python -m pytest tests/test_gpu_ranking.py tests/test_gpu_optim.py -x -q 2>&1 | tail -2
