#!/bin/bash
# round 5: GPU_MAX_HW_QUEUES above the default 4 for the sharded FM step in a world of one (five streams in its graph)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05hq
mkdir -p $O
for rep in 1 2; do
for q in d 5 6 8; do
  if [ $q = d ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  timeout 300 python bench.py --force-sharded --no-extra-configs --steps 50 --warmup 10 --no-cpu-baseline > $O/b.json 2> $O/b.err
  python -c "
import json
try:
    d=json.loads([l for l in open('$O/b.json') if l.startswith('{')][-1])
    print('fm_sharded1_queues_${q}_$rep  ms_per_step %.4f' % d['ms_per_step'])
except Exception as e:
    print('fm_sharded1_queues_${q}_$rep failed', open('$O/b.err').read()[-300:])" | tee -a $O/ab.txt
done
done
