#!/bin/bash
# round 5: the sharded FM step with the owners' half of the backward on the side stream beside the local backward (A/B, world of one
# through RCCL) + the sharded tests
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05p
mkdir -p $O
export TMPDIR=/tmp
RECBOX_AMD_SHARDED_OWNER_BESIDE=3 timeout 1200 python -m pytest tests/test_gpu_ranking.py -x -q -m gpu -k "sharded_fm" 2>&1 | tail -4 | tee $O/tests.txt
for rep in 1 2 3; do
  for v in 0 1 2 3; do
    n=fm_sharded1_beside${v}_$rep
    RECBOX_AMD_SHARDED_OWNER_BESIDE=$v timeout 300 python bench.py --config fm --force-sharded --steps 100 --warmup 5 --no-cpu-baseline > $O/bench_$n.json 2> $O/bench_$n.err
    python - <<PY | tee -a $O/ab.txt
import json
try:
    d=json.loads([l for l in open('$O/bench_$n.json') if l.startswith('{')][-1]); r=d['roofline']
    print('%-28s ms_per_step %.4f  fwd %.1f us' % ('$n', d['ms_per_step'], r['kernel_ms']*1e3))
except Exception as e:
    print('$n', 'failed', e); print(open('$O/bench_$n.err').read()[-1500:])
PY
  done
done
