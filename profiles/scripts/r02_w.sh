#!/bin/bash
out=/root/repo/gpurun_out/r2w
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
cd /root/repo
timeout 2400 python -m pytest tests -x -q -m gpu > $out/tests.log 2>&1
tail -4 $out/tests.log
timeout 300 python bench.py --no-cpu-baseline > $out/bench_fm.json 2>/dev/null; cut -c1-1600 $out/bench_fm.json; timeout 300 python bench.py --no-cpu-baseline > $out/bench_fm2.json 2>/dev/null; cut -c1-300 $out/bench_fm2.json
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $out/prof -o b -- python /root/repo/bench.py --no-cpu-baseline > $out/prof.log 2>&1)
python profiles/topk.py $(find $out/prof -name "*.db" | head -1) 58 > $out/kernel_stats.txt
python profiles/timeline.py $(find $out/prof -name "*.db" | head -1) rezero_rows 30 > $out/timeline.txt 2>&1
rm -rf $out/prof
head -30 $out/timeline.txt
