#!/bin/bash
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_matching.py -x -q -k "linear or mlp or split_bf16 or deepfm" 2>&1 | tail -3
PYTHONPATH=/root/repo timeout 300 python profiles/gemm_shapes.py 2>&1 | grep -v amdgpu.ids
for i in 1 2; do
timeout 300 python bench.py --config deepfm --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('deepfm', round(d['ms_per_step'],4))"
done
