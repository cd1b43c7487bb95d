#!/bin/bash
out=/root/repo/gpurun_out/r2all
mkdir -p $out
export TMPDIR=/tmp
cd /root/repo
for c in youtubednn deepfm sasrec; do timeout 900 python bench.py --config $c > $out/bench_$c.json 2>$out/bench_$c.err; cut -c1-300 $out/bench_$c.json; done
timeout 600 python bench.py --config youtubednn --force-sharded --no-cpu-baseline > $out/bench_youtubednn_sharded1.json 2>/dev/null; cut -c1-300 $out/bench_youtubednn_sharded1.json
