#!/bin/bash
# round 6: sharded FM step in a world of one: ONE hipGraph against hipGraph pieces with the collectives launched between them
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06m
mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2; do
for mode in whole pieces; do
  timeout 300 python bench.py --config fm --force-sharded --sharded-graph $mode --steps 60 --warmup 10 --no-cpu-baseline --no-extra-configs > $O/x.json 2> $O/x.err
  python - <<PY
import json
d = json.loads([l for l in open("$O/x.json") if l.startswith("{")][-1])
print("$mode rep $rep  ms_per_step %.4f" % d["ms_per_step"])
PY
done
done
rm -rf /tmp/prof
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof -o tl -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra-configs --config fm --force-sharded --sharded-graph pieces --steps 20 --warmup 5 > /dev/null 2>&1)
DB=$(find /tmp/prof -name "tl_results.db" | head -1)
python profiles/timeline.py $DB route_count 14 > $O/fm_sharded1_pieces_timeline.txt 2>&1
cat $O/fm_sharded1_pieces_timeline.txt
