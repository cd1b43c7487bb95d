#!/bin/bash
# round 6: the id sort of batch i + 1 beside step i (bench.py --prefetch-sort: a loader one batch ahead) on this round's tree
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06s
mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2; do
for arm in "--steps-per-graph 1" "--prefetch-sort" "--steps-per-graph 4"; do
for dist in uniform zipf; do
  timeout 300 python bench.py --config fm --dist $dist --steps 200 --warmup 16 $arm --no-cpu-baseline --no-extra-configs > $O/x.json 2> $O/x.err
  python - <<PY
import json
try:
    d = json.loads([l for l in open("$O/x.json") if l.startswith("{")][-1])
    print("$arm $dist rep $rep  ms_per_step %.4f" % d["ms_per_step"])
except Exception as e:
    print("$arm $dist rep $rep failed", e)
PY
done
done
done
