#!/bin/bash
out=/root/repo/gpurun_out/r03
mkdir -p $out
export TMPDIR=/tmp
cd /root/repo
python -m pytest tests/test_gpu_shard.py tests/test_gpu_sharded_world2.py -x -q -m gpu 2>&1 | tail -6 > $out/i_tests.log; cat $out/i_tests.log
ms() { python -c "
import json,sys
d=json.loads(open('$1').readline()); print('$2', round(d['ms_per_step'],4))"; }
for c in youtubednn deepfm; do
  timeout 600 python bench.py --config $c --force-sharded --no-cpu-baseline > $out/i_bench_${c}_sharded1.json 2>$out/i_bench_${c}_sharded1.err; ms $out/i_bench_${c}_sharded1.json "$c sharded world-of-one persistent"
  timeout 600 python bench.py --config $c --force-sharded --no-cpu-baseline --fresh-grads > $out/i_bench_${c}_sharded1_fresh.json 2>/dev/null; ms $out/i_bench_${c}_sharded1_fresh.json "$c sharded world-of-one fresh"
done
timeout 600 python bench.py --force-sharded --no-cpu-baseline > $out/i_bench_fm_sharded1.json 2>$out/i_bench_fm_sharded1.err; ms $out/i_bench_fm_sharded1.json "fm sharded world-of-one"
timeout 900 python bench.py > $out/i_bench_full.json 2>$out/i_bench_full.err; python -c "
import json
d=json.loads(open('$out/i_bench_full.json').readline())
print('fm', round(d['ms_per_step'],4), d['roofline']['frac'])
for k,v in d.get('configs',{}).items():
    print(k, v.get('ms_per_step'), (v.get('roofline') or {}).get('frac'), v.get('skipped'), v.get('wall_s'))"
