#!/bin/bash
# round 5: id compaction beside the re-zero (same batch size as the step before) against re-zero -> compaction in a row
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05o
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ranking.py -x -q -m gpu 2>&1 | tail -3 | tee $O/tests.txt
for rep in 1 2 3; do
for d in uniform zipf; do
  for v in 1 0; do
    n=fm_${d}_beside${v}_$rep
    RECBOX_AMD_FM_COMPACT_BESIDE=$v timeout 200 python bench.py --steps 100 --warmup 10 --no-extra-configs --no-cpu-baseline --dist $d > $O/bench_$n.json 2> $O/bench_$n.err
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
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -f csv -d $GRAFT_REPO_ROOT/$O/prof -o fm -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --no-extra-configs --no-cpu-baseline > /dev/null 2>&1)
python profiles/timeline.py $(find $O/prof -name "*kernel_trace.csv" | head -1) compact_ids 30 > $O/fm_replay_timeline_compact_beside.txt 2>&1; cat $O/fm_replay_timeline_compact_beside.txt
find $O/prof -name "*.csv" -size +4000k -delete
