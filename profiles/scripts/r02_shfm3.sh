#!/bin/bash
export TMPDIR=/tmp PYTHONPATH=/root/repo
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_ranking.py tests/test_gpu_sharded_world2.py tests/test_gpu_shard.py -x -q -m gpu -k "sharded or shard" 2>&1 | tail -4
for i in 1 2; do timeout 300 python bench.py --force-sharded --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'; done
