#!/bin/bash
# round 5: capture order of the backward's two chains (which one stays on the queue of the loss kernel in the replayed graph)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05n
mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2 3; do
for d in uniform zipf; do
  for v in side_first main_first; do
    n=fm_${d}_${v}_$rep
    RECBOX_AMD_FM_BWD_ORDER=$v timeout 200 python bench.py --steps 100 --warmup 10 --no-extra-configs --no-cpu-baseline --dist $d > $O/bench_$n.json 2> $O/bench_$n.err
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
(cd /tmp && RECBOX_AMD_FM_BWD_ORDER=main_first timeout 300 rocprofv3 --kernel-trace -f csv -d $GRAFT_REPO_ROOT/$O/prof -o fm -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --no-extra-configs --no-cpu-baseline > /dev/null 2>&1)
python profiles/timeline.py $(find $O/prof -name "*kernel_trace.csv" | head -1) compact_ids 30 > $O/fm_replay_timeline_main_first.txt 2>&1; cat $O/fm_replay_timeline_main_first.txt
find $O/prof -name "*.csv" -size +4000k -delete
