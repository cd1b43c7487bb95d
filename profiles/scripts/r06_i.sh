#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06i
mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests/test_gpu_ranking.py tests/test_gpu_sharded_world2.py -x -q -m gpu 2>&1 | tail -2
for rep in 1 2 3; do
  timeout 300 python bench.py --config fm --force-sharded --steps 60 --warmup 10 --no-cpu-baseline > $O/fm_sh.json 2> $O/fm_sh.err
  python - <<PY
import json
d = json.loads(open("$O/fm_sh.json").read().strip().splitlines()[-1])
print("fm_sharded1 rep $rep ms_per_step %.4f" % d["ms_per_step"])
PY
done
