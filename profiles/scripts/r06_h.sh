#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06h
mkdir -p $O
export TMPDIR=/tmp
export RECBOX_AB_BLOCKSORT_LATE=1
for rep in 1 2 3; do
for first in 0 1; do
for s in 1 4; do
for dist in uniform zipf; do
  export RBX_AB_NUM_FIRST=$first
  timeout 300 python bench.py --config fm --dist $dist --steps 200 --warmup 16 --steps-per-graph $s --no-cpu-baseline --no-extra-configs > $O/x.json 2> $O/x.err
  python - <<PY
import json
d = json.loads(open("$O/x.json").read().strip().splitlines()[-1])
print("num_first $first spg $s $dist rep $rep  ms_per_step %.4f  fwd %.1f us" % (d["ms_per_step"], d["roofline"]["kernel_ms"] * 1e3))
PY
done
done
done
done
