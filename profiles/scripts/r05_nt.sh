#!/bin/bash
# round 5: the quad FM forward with non-temporal loads (1 rows, 2 first-order weights, 3 both), bench.py, two processes each
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05nt
mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2; do
for v in base qnt1 qnt2 qnt3; do
  if [ $v != base ]; then export RECBOX_HIP_LIB=$GRAFT_REPO_ROOT/recbox_amd/lib/librecbox_hip_$v.so; else unset RECBOX_HIP_LIB; fi
  for d in uniform zipf; do
    n=fm_${d}_${v}_$rep
    timeout 200 python bench.py --steps 100 --warmup 10 --no-extra-configs --no-cpu-baseline --dist $d > $O/bench_$n.json 2> $O/bench_$n.err
    python - <<PY | tee -a $O/ab.txt
import json
try:
    d=json.loads([l for l in open('$O/bench_$n.json') if l.startswith('{')][-1]); r=d['roofline']
    print('%-28s ms_per_step %.4f  fwd %.1f us' % ('$n', d['ms_per_step'], r['kernel_ms']*1e3))
except Exception as e:
    print('$n', 'failed', e); print(open('$O/bench_$n.err').read()[-800:])
PY
  done
done
done
