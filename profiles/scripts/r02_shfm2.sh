#!/bin/bash
out=/root/repo/gpurun_out/r2shfm2
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp PYTHONPATH=/root/repo
cd /root/repo
timeout 300 python bench.py --force-sharded --no-cpu-baseline 2>/dev/null | cut -c1-400
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $out/prof -o b -- python /root/repo/bench.py --force-sharded --no-cpu-baseline > $out/prof.log 2>&1)
python profiles/topk.py $(find $out/prof -name "*.db" | head -1) 40 > $out/kernel_stats.txt
python profiles/timeline.py $(find $out/prof -name "*.db" | head -1) route_count 40 > $out/timeline.txt 2>&1
rm -rf $out/prof
head -40 $out/kernel_stats.txt | cut -c1-140
head -90 $out/timeline.txt | cut -c1-130
