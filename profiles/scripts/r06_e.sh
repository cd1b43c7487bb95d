#!/bin/bash
# round 6 (scratch experiment): does the step shorten when the main chain's tail (numeric reductions, 27 us) is gone -- i.e. is the join at the step boundary worth ~10 us?
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06e
mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2; do
for arm in 0 1; do
for s in 1 4; do
  if [ $arm = 1 ]; then export RBX_DEBUG_SKIP_NUMERIC=1; else unset RBX_DEBUG_SKIP_NUMERIC; fi
  timeout 300 python bench.py --config fm --steps 200 --warmup 16 --steps-per-graph $s --no-cpu-baseline --no-extra-configs > $O/x.json 2> $O/x.err
  python - <<PY
import json
d = json.loads(open("$O/x.json").read().strip().splitlines()[-1])
print("skip_numeric $arm spg $s rep $rep  ms_per_step %.4f  fwd %.1f us" % (d["ms_per_step"], d["roofline"]["kernel_ms"] * 1e3))
PY
done
done
done
