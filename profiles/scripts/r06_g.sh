#!/bin/bash
# round 6: per-block sorts of the small tables behind the loss (inside the backward, RECBOX_AB_BLOCKSORT_LATE=1) against behind the forward kernel (0)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06g
mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2 3; do
for arm in 1 0; do
for s in 1 4; do
for dist in uniform zipf; do
  export RECBOX_AB_BLOCKSORT_LATE=$arm
  timeout 300 python bench.py --config fm --dist $dist --steps 200 --warmup 16 --steps-per-graph $s --no-cpu-baseline --no-extra-configs > $O/x.json 2> $O/x.err
  python - <<PY
import json
d = json.loads(open("$O/x.json").read().strip().splitlines()[-1])
print("blocksort_late $arm spg $s $dist rep $rep  ms_per_step %.4f  fwd %.1f us" % (d["ms_per_step"], d["roofline"]["kernel_ms"] * 1e3))
PY
done
done
done
done
