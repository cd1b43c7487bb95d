#!/bin/bash
# round 6: workgroups of the long fix-up launch of the tiered FM plan (32 since round 3: uniform ids leave it empty; Zipf-like ids leave it ~300 chains)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06u
mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2; do
for v in 32 256 512; do
for dist in uniform zipf; do
  export RBX_AB_LONG_CAP=$v
  timeout 300 python bench.py --config fm --dist $dist --steps 200 --warmup 16 --no-cpu-baseline --no-extra-configs > $O/x.json 2> $O/x.err
  python - <<PY
import json
try:
    d = json.loads([l for l in open("$O/x.json") if l.startswith("{")][-1])
    print("long_cap $v $dist rep $rep  ms_per_step %.4f" % d["ms_per_step"])
except Exception as e:
    print("long_cap $v $dist rep $rep failed", e)
PY
done
done
done
