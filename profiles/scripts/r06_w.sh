#!/bin/bash
# round 6: rbx_linear_bwd's general path with db (column sums) BEFORE the dW GEMM ("new") against behind it ("old"); two builds of the library
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06w
mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2 3; do
for lib in new old; do
for cfg in youtubednn deepfm; do
  export RECBOX_HIP_LIB=$GRAFT_REPO_ROOT/profiles/ubench/ab/$lib.so
  timeout 300 python bench.py --config $cfg --steps 40 --warmup 8 --no-cpu-baseline > $O/x.json 2> $O/x.err
  python - <<PY
import json
try:
    d = json.loads([l for l in open("$O/x.json") if l.startswith("{")][-1])
    print("$lib $cfg rep $rep  ms_per_step %.4f" % d["ms_per_step"])
except Exception as e:
    print("$lib $cfg rep $rep failed", e)
PY
done
done
done
