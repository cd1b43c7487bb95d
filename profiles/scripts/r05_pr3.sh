#!/bin/bash
# round 5: the FM step's side chain (the large tables' sort -> reduce: the critical path) on a HIGH priority stream (lab switch)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05pr3
mkdir -p $O
for rep in 1 2 3; do
for d in uniform zipf; do
  for hi in 1 0; do
    RECBOX_AMD_FM_SORT_HIGH=$hi timeout 300 python bench.py --no-extra-configs --steps 100 --warmup 10 --dist $d --no-cpu-baseline > $O/b.json 2> $O/b.err
    python -c "
import json
try:
    d=json.loads([l for l in open('$O/b.json') if l.startswith('{')][-1]); r=d['roofline']
    print('fm_${d}_sort_high${hi}_$rep  ms_per_step %.4f  fwd %.1f us' % (d['ms_per_step'], r['kernel_ms']*1e3))
except Exception as e:
    print('fm_${d}_sort_high${hi}_$rep failed', open('$O/b.err').read()[-300:])" | tee -a $O/ab.txt
  done
done
done
