#!/bin/bash
# round 6: how far ahead of the GPU is the host when it replays the FM graphs?  (host_enqueue_ms_per_step against ms_per_step)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06o
mkdir -p $O
export TMPDIR=/tmp
nproc; lscpu | grep -E "Model name|MHz" | head -3
for rep in 1 2; do
for s in 1 4 8; do
  timeout 300 python bench.py --config fm --steps 200 --warmup 16 --steps-per-graph $s --no-cpu-baseline --no-extra-configs > $O/x.json 2> $O/x.err
  python - <<PY
import json
d = json.loads([l for l in open("$O/x.json") if l.startswith("{")][-1])
print("spg $s rep $rep  ms_per_step %.4f  host enqueue %.4f ms per step" % (d["ms_per_step"], d["host_enqueue_ms_per_step"]))
PY
done
done
