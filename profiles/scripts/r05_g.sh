#!/bin/bash
# round 5: FM with the sort of batch i+1 beside step i (bench.py --prefetch-sort), uniform and Zipf, + timeline
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05g
mkdir -p $O
export TMPDIR=/tmp
for d in uniform zipf; do
  for pf in "" "--prefetch-sort"; do
    n=fm_${d}${pf:+_prefetch}
    timeout 200 python bench.py --steps 50 --warmup 10 --no-extra-configs --no-cpu-baseline --dist $d $pf > $O/bench_$n.json 2> $O/bench_$n.err
    python - <<PY
import json
try:
    d=json.loads([l for l in open('$O/bench_$n.json') if l.startswith('{')][-1]); r=d['roofline']
    print('%-28s ms_per_step %.4f  fwd %.1f us' % ('$n', d['ms_per_step'], r['kernel_ms']*1e3))
except Exception as e:
    print('$n', 'failed', e); print(open('$O/bench_$n.err').read()[-1500:])
PY
  done
done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -f csv -d $GRAFT_REPO_ROOT/$O/prof -o fm -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --no-extra-configs --no-cpu-baseline --prefetch-sort > /dev/null 2>&1)
python profiles/timeline.py $(find $O/prof -name "*kernel_trace.csv" | head -1) fm_quad_fwd 20 > $O/fm_prefetch_replay_timeline.txt 2>&1; cat $O/fm_prefetch_replay_timeline.txt
find $O/prof -name "*.csv" -size +4000k -delete
