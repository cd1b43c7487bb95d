#!/bin/bash
# round 4, first GPU call: teardown experiment, sharded tests with the direct collectives on by default, world-of-one lines
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04a
O=gpurun_out/r04a
for v in nocapture keep release release_sync abort shutdown; do
  timeout 90 python -u profiles/ubench/rccl_teardown.py $v > $O/teardown_$v.log 2>&1
  echo "teardown $v: exit $?" | tee -a $O/teardown.txt
  tail -2 $O/teardown_$v.log >> $O/teardown.txt
done
timeout 900 python -m pytest tests/test_gpu_shard.py tests/test_gpu_sharded_world2.py -x -q -m gpu > $O/tests_shard.log 2>&1
echo "shard tests exit $?" | tee -a $O/summary.txt; tail -5 $O/tests_shard.log
timeout 600 python -m pytest tests/test_gpu_ranking.py -x -q -m gpu -k "sharded or Sharded" > $O/tests_ranking_sharded.log 2>&1
echo "ranking sharded tests exit $?" | tee -a $O/summary.txt; tail -3 $O/tests_ranking_sharded.log
for cfg in fm youtubednn deepfm; do
  for mode in auto pieces eager; do
    if [ $cfg != fm ] && [ $mode = pieces ]; then continue; fi
    timeout 300 python bench.py --config $cfg --force-sharded --sharded-graph $mode --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_${cfg}_sharded1_$mode.json 2> $O/bench_${cfg}_sharded1_$mode.err
    echo "$cfg $mode exit $?: $(python -c "import json,sys; d=json.loads([l for l in open('$O/bench_${cfg}_sharded1_$mode.json') if l.startswith('{')][-1]); print(d['ms_per_step'], d['config']['workload'][-200:])" 2>&1 | tail -1)" | tee -a $O/summary.txt
  done
done
timeout 300 python bench.py --steps 30 --warmup 5 --no-extra-configs --no-cpu-baseline > $O/bench_fm.json 2> $O/bench_fm.err
echo "fm exit $?" | tee -a $O/summary.txt
