#!/bin/bash
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_ranking.py -x -q -m gpu 2>&1 | tail -2
for v in default nopipe; do
  if [ $v = default ]; then unset RECBOX_HIP_LIB; else export RECBOX_HIP_LIB=/root/repo/recbox_amd/lib/variants/$v.so; fi
  for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('$v step_ms', round(d['ms_per_step'],4), 'fwd as-run', round(r['kernel_ms'],4), 'alone', round(r.get('kernel_ms_alone',0),4), 'warm', round(r.get('kernel_ms_warm',0),4))"; done; done
