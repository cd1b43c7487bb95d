#!/bin/bash
# round 5: the replayed timelines of r05_final.sh again (its anchors picked eager warm-up steps), the optimiser tests of the
# union-of-ids step, embed_seq_kernel's parameters at cfg 3's shape
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/r05tl
rm -rf $out; mkdir -p $out
prof() { # name, bench args, anchor kernel, occurrence
  rm -rf $out/prof
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $out/prof -o b -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra-configs $2 > $out/prof_$1.log 2>&1)
  db=$(find $out/prof -name "*.db" | head -1)
  python profiles/timeline.py $db "$3" $4 > $out/$1_replay_timeline.txt 2>&1
  head -2 $out/$1_replay_timeline.txt
  rm -rf $out/prof $out/prof_$1.log
}
prof fm "--steps 20 --warmup 5" compact_ids 30
prof fm_sharded1 "--config fm --force-sharded --steps 20 --warmup 5" route_count 14
prof youtubednn "--config youtubednn --steps 20 --warmup 5" embed_seq 35
prof youtubednn_sharded1 "--config youtubednn --force-sharded --steps 20 --warmup 5" "shard_count_kernel<true>" 35
prof deepfm "--config deepfm --steps 20 --warmup 5" "embed_fwd_kernel<16" 35
prof sasrec "--config sasrec --steps 20 --warmup 5" embed_seq 35
