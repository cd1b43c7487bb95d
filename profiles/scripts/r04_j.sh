#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/r04k
mkdir -p $out
ms() { python -c "import json,sys; d=json.loads([l for l in open('$out/bench_$1.json') if l.startswith('{')][-1]); r=d.get('roofline') or {}; print('$1', round(d['ms_per_step'],4), r.get('kernel_ms'), d['config'].get('exchange'))" 2>&1 | tail -1 | tee -a $out/summary.txt; }
for v in 0; do
  RECBOX_AMD_SHARDED_ONE_SIDE=$v timeout 300 python bench.py --config fm --force-sharded --steps 30 --warmup 5 --no-cpu-baseline > $out/bench_fm_sharded1_oneside$v.json 2>/dev/null; ms fm_sharded1_oneside$v
done
for cfg in youtubednn deepfm; do
  timeout 300 python bench.py --config $cfg --force-sharded --steps 30 --warmup 5 --no-cpu-baseline > $out/bench_${cfg}_sharded1.json 2> $out/bench_${cfg}_sharded1.err; ms ${cfg}_sharded1
done
timeout 600 python -m pytest tests/test_gpu_shard.py tests/test_gpu_sharded_world2.py -q -m gpu -x > $out/tests.log 2>&1
echo "tests exit $?" | tee -a $out/summary.txt; tail -3 $out/tests.log | tee -a $out/summary.txt
