#!/bin/bash
out=/root/repo/gpurun_out/r2sas2
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_matching.py tests/test_gpu_cabi_vs_c_oracle.py -x -q -m gpu > $out/tests.log 2>&1
tail -3 $out/tests.log
for f in 1; do timeout 600 python bench.py --config sasrec --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('step_ms', round(d['ms_per_step'],3))"; done
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $out/prof -o b -- python /root/repo/bench.py --config sasrec --no-cpu-baseline --steps 20 --warmup 5 > $out/prof.log 2>&1)
python profiles/topk.py $(find $out/prof -name "*.db" | head -1) 40 > $out/sasrec_kernel_stats.txt
rm -rf $out/prof
head -16 $out/sasrec_kernel_stats.txt | cut -c1-150
