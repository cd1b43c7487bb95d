#!/bin/bash
# round 5: weight gradients beside the next backward node also under DenseGradSync (its all-reduce joins the side stream first):
# the sharded tests, then the sharded forms in a world of one, on / off
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05db4
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_sharded_world2.py tests/test_gpu_shard.py tests/test_gpu_matching.py -q -m gpu -x 2>&1 | tail -3 | tee $O/tests.txt
for rep in 1 2 3; do
for cfg in deepfm youtubednn; do
  for v in 1 0; do
    n=${cfg}_sharded1_beside${v}_$rep
    RECBOX_AMD_DW_BESIDE=$v timeout 300 python bench.py --config $cfg --force-sharded --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_$n.json 2> $O/bench_$n.err
    python - <<PY | tee -a $O/ab.txt
import json
try:
    d=json.loads([l for l in open('$O/bench_$n.json') if l.startswith('{')][-1])
    print('%-36s ms_per_step %.4f' % ('$n', d['ms_per_step']))
except Exception as e:
    print('$n', 'failed', e); print(open('$O/bench_$n.err').read()[-1500:])
PY
  done
done
done
