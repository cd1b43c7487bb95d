#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06p
mkdir -p $O
export TMPDIR=/tmp
for s in 1 4; do
for steps in 8 20 200; do
  timeout 300 python bench.py --config fm --steps $steps --warmup 8 --steps-per-graph $s --no-cpu-baseline --no-extra-configs > $O/x.json 2> $O/x.err
  python - <<PY
import json
d = json.loads([l for l in open("$O/x.json") if l.startswith("{")][-1])
print("spg $s steps $steps  ms_per_step %.4f  host enqueue %.4f ms per step" % (d["ms_per_step"], d["host_enqueue_ms_per_step"]))
PY
done
done
