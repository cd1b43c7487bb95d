#!/bin/bash
# round 6: A/B of two library builds at small batches (see profiles/r06/small_batches.txt)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06z
mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2 3; do
for lib in new old; do
for a in "--config youtubednn --batch 8192" "--config deepfm --batch 8192" "--config youtubednn --batch 16384"; do
  export RECBOX_HIP_LIB=$GRAFT_REPO_ROOT/profiles/ubench/ab/$lib.so
  timeout 300 python bench.py $a --steps 40 --warmup 8 --no-cpu-baseline > $O/x.json 2> $O/x.err
  python - "$a" <<PY
import json, sys
try:
    d = json.loads([l for l in open("$O/x.json") if l.startswith("{")][-1])
    print("$lib", sys.argv[1], "rep $rep  ms_per_step %.4f" % d["ms_per_step"])
except Exception as e:
    print("$lib", sys.argv[1], "rep $rep failed", e)
PY
done
done
done
RECBOX_HIP_LIB=$GRAFT_REPO_ROOT/profiles/ubench/ab/new.so python -m pytest tests/test_gpu_matching.py -x -q -m gpu 2>&1 | tail -2
