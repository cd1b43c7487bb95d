#!/bin/bash
# round 6: replayed timeline of the FM step on Zipf-like ids
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06t
mkdir -p $O
export TMPDIR=/tmp
rm -rf /tmp/prof
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof -o tl -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra-configs --dist zipf --steps 20 --warmup 8 > /dev/null 2>&1)
DB=$(find /tmp/prof -name "tl_results.db" | head -1)
python profiles/timeline.py $DB rezero_rows 26 2 > $O/fm_zipf_replay_timeline.txt 2>&1
cat $O/fm_zipf_replay_timeline.txt
