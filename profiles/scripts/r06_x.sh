#!/bin/bash
# round 6: the id compaction at the head of the side stream's chain (the forward kernel then follows the re-zero directly) against on the current stream in front of the forward
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06x
mkdir -p $O
export TMPDIR=/tmp
export RECBOX_AB_COMPACT_SIDE=1
python -m pytest tests/test_gpu_ranking.py -x -q -m gpu 2>&1 | tail -2
for rep in 1 2 3; do
for arm in 1 0; do
for dist in uniform zipf; do
  export RECBOX_AB_COMPACT_SIDE=$arm
  timeout 300 python bench.py --config fm --dist $dist --steps 200 --warmup 16 --no-cpu-baseline --no-extra-configs > $O/x.json 2> $O/x.err
  python - <<PY
import json
try:
    d = json.loads([l for l in open("$O/x.json") if l.startswith("{")][-1])
    print("compact_side $arm $dist rep $rep  ms_per_step %.4f  fwd %.1f" % (d["ms_per_step"], d["roofline"]["kernel_ms"]*1e3))
except Exception as e:
    print("compact_side $arm $dist rep $rep failed", e)
PY
done
done
done
