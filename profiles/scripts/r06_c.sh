#!/bin/bash
# round 6: several consecutive steps (one per resident batch) captured in ONE hipGraph -- what the ~20 us between two replays costs.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06c
mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2; do
for s in 1 2 4 8; do
  timeout 300 python bench.py --config fm --steps 200 --warmup 16 --steps-per-graph $s --no-cpu-baseline --no-extra-configs > $O/fm_spg${s}_$rep.json 2> $O/fm_spg${s}_$rep.err
  python - <<PY
import json
d = json.loads(open("$O/fm_spg${s}_$rep.json").read().strip().splitlines()[-1])
print("spg $s rep $rep  ms_per_step %.4f  fwd %.1f us" % (d["ms_per_step"], d["roofline"]["kernel_ms"] * 1e3))
PY
done
done
for s in 1 4; do
  timeout 300 python bench.py --config fm --dist zipf --steps 200 --warmup 16 --steps-per-graph $s --no-cpu-baseline --no-extra-configs > $O/fm_zipf_spg${s}.json 2> $O/fm_zipf_spg${s}.err
  python - <<PY
import json
d = json.loads(open("$O/fm_zipf_spg${s}.json").read().strip().splitlines()[-1])
print("zipf spg $s  ms_per_step %.4f  fwd %.1f us" % (d["ms_per_step"], d["roofline"]["kernel_ms"] * 1e3))
PY
done
