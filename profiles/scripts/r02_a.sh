#!/bin/bash
# round 2, first GPU call: changed tests, rotating-batch bench, kernel stats of the rotating run
out=/root/repo/gpurun_out/r2a
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_matching.py tests/test_gpu_ranking.py -x -q -m gpu > $out/tests.log 2>&1
tail -3 $out/tests.log
timeout 300 python bench.py > $out/bench.json 2> $out/bench.err
timeout 200 python bench.py --rotate 1 --no-cpu-baseline > $out/bench_onebatch.json 2>/dev/null
timeout 200 python bench.py --dist zipf --no-cpu-baseline > $out/bench_zipf.json 2>/dev/null
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $out/prof -o b -- python /root/repo/bench.py --no-cpu-baseline > $out/prof.log 2>&1)
python profiles/topk.py $(find $out/prof -name "*.db" | head -1) 58 > $out/kernel_stats.txt
rm -rf $out/prof
cat $out/bench.json $out/bench_onebatch.json $out/bench_zipf.json
head -30 $out/kernel_stats.txt
