#!/bin/bash
cd /root/repo
for v in 1 0 1 0; do
  RECBOX_AMD_DEFER_IDS=$v timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('DEFER_IDS=$v', r['ms_per_step'], r['roofline']['kernel_ms'], r['roofline'].get('kernel_ms_alone'))"
done
rocm-smi --showclocks 2>/dev/null | head -20
