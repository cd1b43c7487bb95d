#!/bin/bash
# round 6: timeline of replayed FM steps with 4 steps per captured graph (5 consecutive steps: three boundaries inside a graph, one between graphs)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06d
mkdir -p $O
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
export RECBOX_AB_BLOCKSORT_LATE=1
for s in 4; do
  rm -rf /tmp/prof
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof -o tl -- python $R/bench.py --no-cpu-baseline --no-extra-configs --steps 40 --warmup 8 --steps-per-graph $s > /dev/null 2>&1)
  DB=$(find /tmp/prof -name "tl_results.db" | head -1)
  python profiles/timeline.py $DB rezero_rows 30 2 > $O/fm_replay_timeline_spg$s.txt 2>&1
done
cat $O/fm_replay_timeline_spg4.txt
