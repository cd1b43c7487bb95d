#!/bin/bash
# round 6: the row limit of tier A (tables written in full from per-block LDS sorts) once more, now that the large tables' chain bounds the step
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06q
mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2; do
for v in 4096 6000 13000 16384 2200; do
for dist in uniform zipf; do
  export RBX_AB_TA_VOCAB=$v
  timeout 300 python bench.py --config fm --dist $dist --steps 200 --warmup 16 --no-cpu-baseline --no-extra-configs > $O/x.json 2> $O/x.err
  python - <<PY
import json
try:
    d = json.loads([l for l in open("$O/x.json") if l.startswith("{")][-1])
    print("ta_vocab $v $dist rep $rep  ms_per_step %.4f" % d["ms_per_step"])
except Exception as e:
    print("ta_vocab $v $dist rep $rep failed", e)
PY
done
done
done
