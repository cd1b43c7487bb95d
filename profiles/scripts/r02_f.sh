#!/bin/bash
out=/root/repo/gpurun_out/r2j
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_ranking.py tests/test_gpu_cabi_vs_c_oracle.py tests/test_gpu_edge_cases.py tests/test_gpu_sharded_world2.py -x -q -m gpu > $out/tests.log 2>&1
tail -15 $out/tests.log
timeout 300 python bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.err
cat $out/bench.json
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $out/prof -o b -- python /root/repo/bench.py --no-cpu-baseline > $out/prof.log 2>&1)
python profiles/topk.py $(find $out/prof -name "*.db" | head -1) 58 > $out/kernel_stats.txt
rm -rf $out/prof
head -24 $out/kernel_stats.txt
