#!/bin/bash
out=/root/repo/gpurun_out/r03
mkdir -p $out
export TMPDIR=/tmp
cd /root/repo
python -m pytest tests/test_gpu_shard.py -x -q -m gpu -k persistent 2>&1 | tail -3
ms() { python -c "
import json,sys
d=json.loads(open('$1').readline()); print('$2', round(d['ms_per_step'],4))"; }
for c in youtubednn deepfm; do
  timeout 600 python bench.py --config $c --force-sharded --no-cpu-baseline > $out/j_bench_${c}_sharded1.json 2>$out/j_bench_${c}_sharded1.err; ms $out/j_bench_${c}_sharded1.json "$c sharded world-of-one persistent"
done
prof() { # name, bench args
  rm -rf $out/prof
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $out/prof -o b -- python /root/repo/bench.py --no-cpu-baseline $2 > $out/prof_$1.log 2>&1)
  python profiles/topk.py $(find $out/prof -name "*.db" | head -1) 25 > $out/$1_kernel_stats.txt
  rm -rf $out/prof
}
prof j_youtubednn_sharded1 "--config youtubednn --force-sharded --steps 20 --warmup 5"
head -40 $out/j_youtubednn_sharded1_kernel_stats.txt | cut -c1-120
