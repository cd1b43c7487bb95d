#!/bin/bash
# round 4, third GPU call: GPU suite with tier C + zero-copy reductions, FM step with tier C on / off (bench, kernel trace, replay timeline), sharded lines
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/r04c
mkdir -p $out
rm -f gpurun_out/parity_errors.txt
timeout 1500 python -m pytest tests -q -m gpu > $out/gpu_tests.log 2>&1
echo "gpu tests exit $?" | tee -a $out/summary.txt; tail -8 $out/gpu_tests.log | tee -a $out/summary.txt
cp gpurun_out/parity_errors.txt $out/ 2>/dev/null
ms() { python -c "import json,sys; d=json.loads([l for l in open('$out/bench_$1.json') if l.startswith('{')][-1]); print('$1', round(d['ms_per_step'],4), (d.get('roofline') or {}).get('kernel_ms'), d['config']['workload'][-120:])" 2>&1 | tail -1 | tee -a $out/summary.txt; }
timeout 300 python bench.py --steps 50 --warmup 10 --no-extra-configs --no-cpu-baseline > $out/bench_fm.json 2> $out/bench_fm.err; ms fm
RBX_FM_TIER_C=0 timeout 300 python bench.py --steps 50 --warmup 10 --no-extra-configs --no-cpu-baseline > $out/bench_fm_tierc_off.json 2>/dev/null; ms fm_tierc_off
timeout 300 python bench.py --steps 50 --warmup 10 --no-extra-configs --no-cpu-baseline --dist zipf > $out/bench_fm_zipf.json 2>/dev/null; ms fm_zipf
RBX_FM_TIER_C=0 timeout 300 python bench.py --steps 50 --warmup 10 --no-extra-configs --no-cpu-baseline --dist zipf > $out/bench_fm_zipf_tierc_off.json 2>/dev/null; ms fm_zipf_tierc_off
RECBOX_AMD_FM_BLOCKSORT_AT=fwd timeout 300 python bench.py --steps 50 --warmup 10 --no-extra-configs --no-cpu-baseline > $out/bench_fm_blocksort_fwd.json 2>/dev/null; ms fm_blocksort_fwd
RECBOX_AMD_FM_NUMERIC=first timeout 300 python bench.py --steps 50 --warmup 10 --no-extra-configs --no-cpu-baseline > $out/bench_fm_numeric_first.json 2>/dev/null; ms fm_numeric_first
RECBOX_AMD_FM_TWO_CHAINS=0 timeout 300 python bench.py --steps 50 --warmup 10 --no-extra-configs --no-cpu-baseline > $out/bench_fm_one_chain.json 2>/dev/null; ms fm_one_chain
prof() { # name, bench args, anchor kernel, occurrence
  rm -rf $out/prof
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $out/prof -o b -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra-configs $2 > $out/prof_$1.log 2>&1)
  db=$(find $out/prof -name "*.db" | head -1)
  python profiles/topk.py $db 25 > $out/$1_kernel_stats.txt
  python profiles/timeline.py $db "$3" $4 > $out/$1_replay_timeline.txt 2>&1
  rm -rf $out/prof
}
prof fm "--steps 20 --warmup 5" compact_ids 30
prof fm_sharded1 "--config fm --force-sharded --steps 20 --warmup 5" route_count 16
for cfg in fm youtubednn deepfm; do
  timeout 300 python bench.py --config $cfg --force-sharded --steps 30 --warmup 5 --no-cpu-baseline > $out/bench_${cfg}_sharded1.json 2> $out/bench_${cfg}_sharded1.err; ms ${cfg}_sharded1
done
prof youtubednn_sharded1 "--config youtubednn --force-sharded --steps 20 --warmup 5" "shard_count_kernel<true>" 16
prof deepfm_sharded1 "--config deepfm --force-sharded --steps 20 --warmup 5" "shard_count_kernel<true>" 16
