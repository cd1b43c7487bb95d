#!/bin/bash
# round 6: sharded FM step (world of one through RCCL): the gradients' exchange on a stream of its own (RECBOX_AB_WIRE=1) against behind the owners' id sort (0)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06l
mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2 3; do
for arm in ${ARMS:-1 0}; do
  export RECBOX_AB_WIRE=$arm
  timeout 300 python bench.py --config fm --force-sharded --steps 60 --warmup 10 --no-cpu-baseline --no-extra-configs > $O/x.json 2> $O/x.err
  python - <<PY
import json
d = json.loads([l for l in open("$O/x.json") if l.startswith("{")][-1])
print("wire $arm rep $rep  ms_per_step %.4f" % d["ms_per_step"])
PY
done
done
export RECBOX_AB_WIRE=${TL_ARM:-1}
rm -rf /tmp/prof
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof -o tl -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra-configs --config fm --force-sharded --steps 20 --warmup 5 > /dev/null 2>&1)
DB=$(find /tmp/prof -name "tl_results.db" | head -1)
python profiles/timeline.py $DB route_count 14 > $O/fm_sharded1_timeline.txt 2>&1
cat $O/fm_sharded1_timeline.txt
python -m pytest tests/test_gpu_ranking.py -x -q -m gpu -k "sharded" 2>&1 | tail -2
