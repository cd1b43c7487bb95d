#!/bin/bash
out=/root/repo/gpurun_out/r2shfm
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
cd /root/repo
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $out/prof -o b -- python /root/repo/bench.py --force-sharded --no-cpu-baseline > $out/prof.log 2>&1)
python profiles/topk.py $(find $out/prof -name "*.db" | head -1) 50 > $out/kernel_stats.txt
python profiles/timeline.py $(find $out/prof -name "*.db" | head -1) route_count 60 > $out/timeline.txt 2>&1
rm -rf $out/prof
head -60 $out/timeline.txt | cut -c1-120
