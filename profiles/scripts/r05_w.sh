#!/bin/bash
# round 5: the sharded YoutubeDNN step in a world of one (every collective issued): kernel stats + one replayed timeline
cd "$GRAFT_REPO_ROOT" || exit 1
out=$GRAFT_REPO_ROOT/gpurun_out/r05w
mkdir -p $out
export TMPDIR=/tmp
rm -rf $out/prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $out/prof -o b -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --config youtubednn --force-sharded --steps 20 --warmup 5 > $out/prof.log 2>&1)
tail -2 $out/prof.log | cut -c1-300
db=$(find $out/prof -name "*.db" | head -1)
python profiles/topk.py $db 40 > $out/youtubednn_sharded1_kernel_stats.txt
python profiles/timeline.py $db route_count -6 > $out/youtubednn_sharded1_replay_timeline.txt 2>&1
rm -rf $out/prof
cat $out/youtubednn_sharded1_replay_timeline.txt
