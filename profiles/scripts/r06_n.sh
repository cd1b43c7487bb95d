#!/bin/bash
# round 6: the streaming parts of the towers' side work (db column sums, DeepFM's first-order head) ahead of dx (RECBOX_AB_DB_FIRST=1) against behind the dW GEMM (0)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06n
mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests/test_gpu_matching.py -x -q -m gpu -k "deepfm or DeepFM" 2>&1 | tail -2
for rep in 1 2 3; do
for arm in 1 0; do
for cfg in deepfm; do
  export RECBOX_AB_DB_FIRST=$arm
  timeout 300 python bench.py --config $cfg --steps 40 --warmup 8 --no-cpu-baseline > $O/x.json 2> $O/x.err
  python - <<PY
import json
d = json.loads([l for l in open("$O/x.json") if l.startswith("{")][-1])
print("db_first $arm $cfg rep $rep  ms_per_step %.4f" % d["ms_per_step"])
PY
done
done
done
