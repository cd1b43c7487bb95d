#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof -o b -- python /root/repo/bench.py --no-cpu-baseline --config sasrec --steps 20 --warmup 5 > /dev/null 2>&1)
python profiles/topk.py $(find /tmp/prof -name "*.db" | head -1) 36
