#!/bin/bash
# round 6: sorted pairs per lane group of the large tables' segmented reduce (24 by the plan's rule at 786 k pairs)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06r
mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2; do
for v in 24 16 32 40; do
for dist in uniform zipf; do
  export RBX_AB_CHUNK=$v
  timeout 300 python bench.py --config fm --dist $dist --steps 200 --warmup 16 --no-cpu-baseline --no-extra-configs > $O/x.json 2> $O/x.err
  python - <<PY
import json
try:
    d = json.loads([l for l in open("$O/x.json") if l.startswith("{")][-1])
    print("chunk $v $dist rep $rep  ms_per_step %.4f" % d["ms_per_step"])
except Exception as e:
    print("chunk $v $dist rep $rep failed", e)
PY
done
done
done
