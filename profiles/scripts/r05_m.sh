#!/bin/bash
# round 5: chained sort on / off, several processes each (a 1.8x outlier on Zipf ids in r05_l.sh)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05m
mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2 3 4 5; do
for d in zipf uniform; do
  for ch in 1 0; do
    n=fm_${d}_ch${ch}_$rep
    RBX_SORT_CHAINED=$ch timeout 200 python bench.py --steps 100 --warmup 10 --no-extra-configs --no-cpu-baseline --dist $d > $O/bench_$n.json 2> $O/bench_$n.err
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
done
