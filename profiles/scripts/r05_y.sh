#!/bin/bash
# round 5: GPU_MAX_HW_QUEUES (default 4) = 2 / 3 for every bench line (r05_x.sh: 2 gains 2-3 % on the FM step, 8 loses 2.5x)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05y
mkdir -p $O
export TMPDIR=/tmp
run() { # name, queues, bench args
  n=$1; q=$2; shift 2
  if [ $q = d ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  timeout 400 python bench.py --no-extra-configs --no-cpu-baseline "$@" > $O/bench_${n}_q$q.json 2> $O/bench_${n}_q$q.err
  python - <<PY | tee -a $O/ab.txt
import json
try:
    d=json.loads([l for l in open('$O/bench_${n}_q$q.json') if l.startswith('{')][-1])
    print('%-28s queues %s  ms_per_step %.4f' % ('$n', '$q', d['ms_per_step']))
except Exception as e:
    print('$n q$q', 'failed', e); print(open('$O/bench_${n}_q$q.err').read()[-800:])
PY
}
for q in d 2 3 d 2 3; do
  run fm $q --steps 100 --warmup 10
  run fm_zipf $q --steps 100 --warmup 10 --dist zipf
  run fm_sharded1 $q --force-sharded --steps 50 --warmup 10
done
for q in d 2 3; do
  run youtubednn $q --config youtubednn --steps 30 --warmup 5
  run youtubednn_sharded1 $q --config youtubednn --force-sharded --steps 30 --warmup 5
  run deepfm $q --config deepfm --steps 30 --warmup 5
  run sasrec $q --config sasrec --steps 20 --warmup 5
done
