#!/bin/bash
out=/root/repo/gpurun_out/r2half
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_matching.py tests/test_gpu_ranking.py -x -q -m gpu > $out/tests.log 2>&1
tail -3 $out/tests.log
for c in deepfm youtubednn sasrec; do for h in 1 0; do RBX_GEMM_HALF_TAIL=$h timeout 600 python bench.py --config $c --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$c HALF_TAIL=$h step_ms', round(d['ms_per_step'],3), 'kernel_ms', round(d['roofline']['kernel_ms'],4), 'frac', round(d['roofline']['frac'],3))"; done; done
