#!/bin/bash
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_matching.py -x -q -k "linear or mlp or split_bf16 or deepfm or youtube or dssm" 2>&1 | tail -3
for i in 1 2; do
for cfg in deepfm youtubednn; do
timeout 300 python bench.py --config $cfg --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$cfg', round(d['ms_per_step'],4))"
done
done
