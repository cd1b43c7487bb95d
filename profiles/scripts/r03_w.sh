#!/bin/bash
cd /root/repo
for v in fwd bwd 0; do
RECBOX_AMD_BN_IN_GEMM=$v timeout 300 python bench.py --config deepfm --no-cpu-baseline --steps 20 --warmup 5 2>/tmp/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('deepfm bn_in_gemm=$v', round(d['ms_per_step'],4))" || tail -5 /tmp/err.txt
done
export TMPDIR=/tmp
(cd /tmp && RECBOX_AMD_BN_IN_GEMM=1 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof -o b -- python /root/repo/bench.py --no-cpu-baseline --config deepfm --steps 20 --warmup 5 > /dev/null 2>&1)
python profiles/topk.py $(find /tmp/prof -name "*.db" | head -1) 14 | cut -c1-120
