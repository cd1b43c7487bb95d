#!/bin/bash
# round 5: the generic lookup's re-zero left for the backward pass (config.defer_rezero) on / off: tests, then youtubednn / deepfm
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05dz
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_optim.py tests/test_gpu_matching.py tests/test_gpu_ranking.py tests/test_gpu_edge_cases.py -q -m gpu -x 2>&1 | tail -4 | tee $O/tests.txt
for rep in 1 2; do
for cfg in youtubednn deepfm; do
  for dz in 1 0; do
    n=${cfg}_defer${dz}_$rep
    RECBOX_AMD_DEFER_REZERO=$dz timeout 300 python bench.py --config $cfg --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_$n.json 2> $O/bench_$n.err
    python - <<PY | tee -a $O/ab.txt
import json
try:
    d=json.loads([l for l in open('$O/bench_$n.json') if l.startswith('{')][-1])
    print('%-28s ms_per_step %.4f' % ('$n', d['ms_per_step']))
except Exception as e:
    print('$n', 'failed', e); print(open('$O/bench_$n.err').read()[-1500:])
PY
  done
done
done
