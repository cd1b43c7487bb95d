#!/bin/bash
out=/root/repo/gpurun_out/r03
mkdir -p $out
cd /root/repo
ms() { python -c "
import json,sys
d=json.loads(open('$1').readline()); print('$2', round(d['ms_per_step'],4))"; }
for o in none sparse_adam dense_adam; do
  timeout 900 python bench.py --config youtubednn --no-cpu-baseline --optimizer $o --steps 20 --warmup 5 > $out/l_bench_youtubednn_$o.json 2>$out/l_bench_youtubednn_$o.err; ms $out/l_bench_youtubednn_$o.json "youtubednn optimizer=$o" || tail -5 $out/l_bench_youtubednn_$o.err
done
for o in sparse_adam dense_adam; do
  timeout 900 python bench.py --config deepfm --no-cpu-baseline --optimizer $o --steps 20 --warmup 5 > $out/l_bench_deepfm_$o.json 2>$out/l_bench_deepfm_$o.err; ms $out/l_bench_deepfm_$o.json "deepfm optimizer=$o" || tail -5 $out/l_bench_deepfm_$o.err
done
