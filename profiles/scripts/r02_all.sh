#!/bin/bash
# full GPU suite + the four bench configs (N = 1) with the current tree
out=/root/repo/gpurun_out/r2all
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
cd /root/repo
timeout 2400 python -m pytest tests -x -q -m gpu > $out/tests.log 2>&1
tail -3 $out/tests.log
timeout 600 python bench.py > $out/bench_fm.json 2>$out/bench_fm.err; cut -c1-400 $out/bench_fm.json
for c in youtubednn deepfm sasrec; do timeout 900 python bench.py --config $c > $out/bench_$c.json 2>$out/bench_$c.err; cut -c1-2200 $out/bench_$c.json; done
timeout 600 python bench.py --config youtubednn --force-sharded --no-cpu-baseline > $out/bench_youtubednn_sharded1.json 2>/dev/null; cut -c1-400 $out/bench_youtubednn_sharded1.json
