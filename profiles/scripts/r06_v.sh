#!/bin/bash
# round 6: after the long fix-up change: full GPU suite, every bench line once
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06v
mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests -q -m gpu 2>&1 | tail -2
for args in "--config fm" "--config fm --dist zipf" "--config fm --steps-per-graph 1 --dist zipf" "--config youtubednn" "--config deepfm" "--config sasrec" "--config fm --force-sharded" "--config youtubednn --dist zipf" "--config deepfm --dist zipf"; do
  timeout 400 python bench.py $args --steps 40 --warmup 8 --no-cpu-baseline --no-extra-configs > $O/x.json 2> $O/x.err
  python - "$args" <<PY
import json, sys
try:
    d = json.loads([l for l in open("$O/x.json") if l.startswith("{")][-1])
    print(sys.argv[1], " ms_per_step %.4f" % d["ms_per_step"])
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
done
