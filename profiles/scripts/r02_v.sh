#!/bin/bash
out=/root/repo/gpurun_out/r2v
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
cd /root/repo
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $out/prof -o b -- python /root/repo/bench.py --config youtubednn --no-cpu-baseline --steps 20 --warmup 5 > $out/prof_yt.log 2>&1)
python profiles/topk.py $(find $out/prof -name "*.db" | head -1) 26 > $out/youtubednn_kernel_stats.txt
rm -rf $out/prof
head -30 $out/youtubednn_kernel_stats.txt
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $out/prof -o b -- python /root/repo/bench.py --config youtubednn --force-sharded --no-cpu-baseline --steps 20 --warmup 5 > $out/prof_yts.log 2>&1)
python profiles/topk.py $(find $out/prof -name "*.db" | head -1) 26 > $out/youtubednn_sharded1_kernel_stats.txt
rm -rf $out/prof
head -34 $out/youtubednn_sharded1_kernel_stats.txt
grep '^{' $out/prof_yts.log | cut -c1-300
