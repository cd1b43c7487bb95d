#!/bin/bash
# round 6: replayed timeline of YoutubeDNN at 8 192 samples (cfg 3's per-GPU batch under 8-GPU strong scaling)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06y
mkdir -p $O
export TMPDIR=/tmp
rm -rf /tmp/prof
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof -o tl -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --config youtubednn --batch 8192 --steps 20 --warmup 5 > /dev/null 2>&1)
DB=$(find /tmp/prof -name "tl_results.db" | head -1)
python profiles/timeline.py $DB embed_seq 35 > $O/youtubednn_b8192_timeline.txt 2>&1
cat $O/youtubednn_b8192_timeline.txt
