#!/bin/bash
out=/root/repo/gpurun_out/r2q
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
cd /root/repo
timeout 1200 python -m pytest tests/test_gpu_matching.py -x -q -m gpu -k "attention or sasrec or sdpa or target" > $out/tests.log 2>&1; tail -3 $out/tests.log
timeout 600 python bench.py --config sasrec --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_sasrec.json 2> $out/err.txt
python -c "import json; r=json.load(open('$out/bench_sasrec.json')); print('sasrec', r['ms_per_step'], r['roofline']['kernel_ms'], r['roofline']['frac'])"
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $out/prof -o b -- python /root/repo/bench.py --config sasrec --no-cpu-baseline --steps 20 --warmup 5 > $out/prof.log 2>&1)
python profiles/topk.py $(find $out/prof -name "*.db" | head -1) 26 > $out/sasrec_kernel_stats.txt
rm -rf $out/prof
head -24 $out/sasrec_kernel_stats.txt
