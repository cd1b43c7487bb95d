#!/bin/bash
# tier-A FM step: kernel stats + one replay timeline
out=/root/repo/gpurun_out/r03
mkdir -p $out
export TMPDIR=/tmp
cd /root/repo
python -m pytest tests/test_gpu_ranking.py -x -q -m gpu 2>&1 | tail -5 > $out/b_tests.log; cat $out/b_tests.log
prof() { # name, env, bench args
  rm -rf $out/prof
  (cd /tmp && env $2 timeout 900 rocprofv3 --kernel-trace --stats -d $out/prof -o b -- python /root/repo/bench.py --no-cpu-baseline $3 > $out/prof_$1.log 2>&1)
  db=$(find $out/prof -name "*.db" | head -1)
  python profiles/topk.py $db 40 > $out/$1_kernel_stats.txt
  python profiles/timeline.py $db rezero_rows 30 > $out/$1_replay_timeline.txt 2>&1
  tail -1 $out/prof_$1.log | cut -c1-200
  rm -rf $out/prof
}
prof fm_tier "RBX_X=1" ""
prof fm_notier "RBX_FM_TIER_A=0" ""
cat $out/fm_tier_replay_timeline.txt
