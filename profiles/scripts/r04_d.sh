#!/bin/bash
# tier C kernel by parts (RBX_TC_ABL: 2 = scan alone, 1 = scan + sort): kernel averages under rocprofv3
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/r04d
mkdir -p $out
for abl in 0 1 2; do
  rm -rf $out/prof
  (cd /tmp && RBX_TC_ABL=$abl timeout 300 rocprofv3 --kernel-trace --stats -d $out/prof -o b -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra-configs --steps 20 --warmup 5 > $out/prof_$abl.log 2>&1)
  db=$(find $out/prof -name "*.db" | head -1)
  python profiles/topk.py $db | grep -E "tc_reduce|ta_reduce|kernel " | tee -a $out/summary.txt
  rm -rf $out/prof
done
