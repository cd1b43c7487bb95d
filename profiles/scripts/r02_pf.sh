#!/bin/bash
out=/root/repo/gpurun_out/r2pf
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
cd /root/repo
timeout 1200 python -m pytest tests/test_gpu_ranking.py -x -q -m gpu > $out/tests.log 2>&1
tail -4 $out/tests.log
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline > $out/bench_fm$i.json 2>$out/bench_fm$i.err; cut -c1-900 $out/bench_fm$i.json; tail -2 $out/bench_fm$i.err | cut -c1-300; done
timeout 300 python bench.py --no-cpu-baseline --prefetch-sort 2>/dev/null | cut -c1-330
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $out/prof -o b -- python /root/repo/bench.py --no-cpu-baseline > $out/prof.log 2>&1)
python profiles/topk.py $(find $out/prof -name "*.db" | head -1) 30 > $out/kernel_stats.txt
python profiles/timeline.py $(find $out/prof -name "*.db" | head -1) fm_fused_fwd 30 > $out/timeline.txt 2>&1
rm -rf $out/prof
head -30 $out/timeline.txt
