#!/bin/bash
# round 6: the numeric features' reductions as extra workgroups of tier A's launch (RBX_AB_NUM_IN_TA=1) against the launch of their own (0)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06f
mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests/test_gpu_ranking.py tests/test_gpu_cabi_vs_c_oracle.py -x -q -m gpu 2>&1 | tail -5
for rep in 1 2 3; do
for arm in 1 0; do
for s in 1 4; do
  export RBX_AB_NUM_IN_TA=$arm
  timeout 300 python bench.py --config fm --steps 200 --warmup 16 --steps-per-graph $s --no-cpu-baseline --no-extra-configs > $O/x.json 2> $O/x.err
  python - <<PY
import json
d = json.loads(open("$O/x.json").read().strip().splitlines()[-1])
print("num_in_ta $arm spg $s rep $rep  ms_per_step %.4f  fwd %.1f us" % (d["ms_per_step"], d["roofline"]["kernel_ms"] * 1e3))
PY
done
done
done
