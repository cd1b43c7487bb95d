#!/bin/bash
# round 4, second GPU call: full GPU suite (parity ledger), world-of-one sharded lines + kernel traces / replay timelines
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/r04b
mkdir -p $out
rm -f gpurun_out/parity_errors.txt
timeout 1500 python -m pytest tests -x -q -m gpu > $out/gpu_tests.log 2>&1
echo "gpu tests exit $?" | tee -a $out/summary.txt; tail -4 $out/gpu_tests.log | tee -a $out/summary.txt
cp gpurun_out/parity_errors.txt $out/ 2>/dev/null
ms() { python -c "import json,sys; d=json.loads([l for l in open('$out/bench_$1.json') if l.startswith('{')][-1]); print('$1', round(d['ms_per_step'],4), d['config']['workload'][-160:])" 2>&1 | tail -1 | tee -a $out/summary.txt; }
for cfg in fm youtubednn deepfm; do
  timeout 300 python bench.py --config $cfg --force-sharded --steps 30 --warmup 5 --no-cpu-baseline > $out/bench_${cfg}_sharded1.json 2> $out/bench_${cfg}_sharded1.err; ms ${cfg}_sharded1
done
timeout 300 python bench.py --config fm --force-sharded --sharded-graph pieces --steps 30 --warmup 5 --no-cpu-baseline > $out/bench_fm_sharded1_pieces.json 2>/dev/null; ms fm_sharded1_pieces
prof() { # name, bench args, anchor kernel, occurrence
  rm -rf $out/prof
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $out/prof -o b -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline $2 > $out/prof_$1.log 2>&1)
  db=$(find $out/prof -name "*.db" | head -1)
  python profiles/topk.py $db 25 > $out/$1_kernel_stats.txt
  python profiles/timeline.py $db "$3" $4 > $out/$1_replay_timeline.txt 2>&1
  rm -rf $out/prof
}
prof fm_sharded1 "--config fm --force-sharded --steps 20 --warmup 5" route_count -5
prof youtubednn_sharded1 "--config youtubednn --force-sharded --steps 20 --warmup 5" "shard_count_kernel<true>" -5
prof deepfm_sharded1 "--config deepfm --force-sharded --steps 20 --warmup 5" "shard_count_kernel<true>" -5
timeout 300 python bench.py --steps 30 --warmup 5 --no-extra-configs --no-cpu-baseline > $out/bench_fm.json 2> $out/bench_fm.err; ms fm
