#!/bin/bash
out=/root/repo/gpurun_out/r2dfm
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_matching.py tests/test_gpu_sharded_world2.py -x -q -m gpu > $out/tests.log 2>&1
tail -4 $out/tests.log
for f in 1 0; do RECBOX_AMD_FUSE_DEEPFM_INPUT=$f timeout 600 python bench.py --config deepfm --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('FUSE_DEEPFM_INPUT=$f step_ms', round(d['ms_per_step'],3), d['roofline'].get('kernel_ms'))"; done
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $out/prof -o b -- python /root/repo/bench.py --config deepfm --no-cpu-baseline --steps 20 --warmup 5 > $out/prof.log 2>&1)
python profiles/topk.py $(find $out/prof -name "*.db" | head -1) 30 > $out/deepfm_kernel_stats.txt
rm -rf $out/prof
head -22 $out/deepfm_kernel_stats.txt | cut -c1-150
