#!/bin/bash
out=/root/repo/gpurun_out/r2z
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
cd /root/repo
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | cut -c1-330; done
timeout 300 python bench.py --no-cpu-baseline --contiguous-ids 2>/dev/null | cut -c1-330
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $out/prof -o b -- python /root/repo/bench.py --no-cpu-baseline --contiguous-ids > $out/prof.log 2>&1)
python profiles/topk.py $(find $out/prof -name "*.db" | head -1) 24 > $out/kernel_stats_contig.txt
python profiles/timeline.py $(find $out/prof -name "*.db" | head -1) rezero_rows 30 > $out/timeline_contig.txt 2>&1
rm -rf $out/prof
head -24 $out/kernel_stats_contig.txt | cut -c1-150
head -24 $out/timeline_contig.txt
