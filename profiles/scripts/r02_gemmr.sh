#!/bin/bash
cd /root/repo
for v in 0 1 2; do RBX_GEMM_ROUNDS=$v timeout 600 python bench.py --config deepfm --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('GEMM_ROUNDS=$v step_ms', round(d['ms_per_step'],3), 'gemm_ms', round(d['roofline']['kernel_ms'],4), 'frac', round(d['roofline']['frac'],3))"; done
