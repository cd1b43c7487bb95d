#!/bin/bash
out=/root/repo/gpurun_out/r2i
rm -rf $out; mkdir -p $out
cd /root/repo
timeout 1200 python -m pytest tests/test_gpu_matching.py -x -q -m gpu > $out/matching.log 2>&1
tail -6 $out/matching.log
timeout 600 python bench.py --config sasrec --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_sasrec.json 2> $out/bench_sasrec.err
cat $out/bench_sasrec.json; tail -2 $out/bench_sasrec.err | cut -c1-300
timeout 600 python bench.py --config youtubednn --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_youtubednn.json 2> $out/bench_youtubednn.err
cat $out/bench_youtubednn.json; tail -2 $out/bench_youtubednn.err | cut -c1-300
