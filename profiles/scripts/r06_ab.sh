#!/bin/bash
# round 6: the generic lookup's id-sort fork captured BEHIND the first tower GEMM (RECBOX_AB_DEFER_SORT=1) instead of in front of it
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06ab
mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2 3; do
for arm in 1 0; do
for a in "--config youtubednn --batch 8192" "--config youtubednn" "--config deepfm --batch 8192" "--config deepfm"; do
  export RECBOX_AB_DEFER_SORT=$arm
  timeout 300 python bench.py $a --steps 40 --warmup 8 --no-cpu-baseline > $O/x.json 2> $O/x.err
  python - "$a" <<PY
import json, sys
try:
    d = json.loads([l for l in open("$O/x.json") if l.startswith("{")][-1])
    print("defer $arm", sys.argv[1], "rep $rep  ms_per_step %.4f" % d["ms_per_step"])
except Exception as e:
    print("defer $arm", sys.argv[1], "rep $rep failed", e)
PY
done
done
done
