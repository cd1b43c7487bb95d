#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/r04i
mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_sharded_world2.py tests/test_gpu_matching.py -q -m gpu -x -k "rccl or bench_script or second_oracle or youtubednn_golden or cfg4 or cfg5" > $out/tests.log 2>&1
echo "tests exit $?" | tee -a $out/summary.txt; tail -3 $out/tests.log | tee -a $out/summary.txt
ms() { python -c "import json,sys; d=json.loads([l for l in open('$out/bench_$1.json') if l.startswith('{')][-1]); r=d.get('roofline') or {}; print('$1', round(d['ms_per_step'],4), r.get('kernel_ms'), d['config'].get('exchange'))" 2>&1 | tail -1 | tee -a $out/summary.txt; }
for cfg in fm youtubednn deepfm; do
  timeout 300 python bench.py --config $cfg --force-sharded --steps 30 --warmup 5 --no-cpu-baseline > $out/bench_${cfg}_sharded1.json 2> $out/bench_${cfg}_sharded1.err; ms ${cfg}_sharded1
done
rm -rf $out/prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $out/prof -o b -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --config fm --force-sharded --steps 20 --warmup 5 > $out/prof.log 2>&1)
db=$(find $out/prof -name "*.db" | head -1)
python profiles/topk.py $db 25 > $out/fm_sharded1_kernel_stats.txt
python profiles/timeline.py $db route_count 14 > $out/fm_sharded1_replay_timeline.txt 2>&1
rm -rf $out/prof
