#!/bin/bash
# round 6: 1 against 4 steps per captured graph once more (the evidence run's box showed no difference), short and long runs
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06k
mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2 3; do
for steps in 20 200; do
for s in 1 4; do
  timeout 300 python bench.py --config fm --steps $steps --warmup 8 --steps-per-graph $s --no-cpu-baseline --no-extra-configs > $O/x.json 2> $O/x.err
  python - <<PY
import json
d = json.loads([l for l in open("$O/x.json") if l.startswith("{")][-1])
print("spg $s steps $steps rep $rep  ms_per_step %.4f  fwd %.1f us" % (d["ms_per_step"], d["roofline"]["kernel_ms"] * 1e3))
PY
done
done
done
rocm-smi --showclocks 2>/dev/null | head -20
