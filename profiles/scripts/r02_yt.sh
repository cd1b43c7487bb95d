#!/bin/bash
out=/root/repo/gpurun_out/r2yt
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_matching.py tests/test_gpu_sharded_world2.py tests/test_gpu_shard.py -x -q -m gpu > $out/tests.log 2>&1
tail -4 $out/tests.log
for i in 1 2; do timeout 600 python bench.py --config youtubednn --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('youtubednn step_ms', round(d['ms_per_step'],4), d['roofline'].get('frac'))"; done
timeout 600 python bench.py --config youtubednn --force-sharded --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('youtubednn sharded-1 step_ms', round(d['ms_per_step'],4), d['roofline'].get('frac'))"
