#!/bin/bash
# round 5: where the re-zero goes when the sort keeps the previous step's pairs apart: front (before the forward kernel), mid (behind
# it), late (behind the per-block sorts), against the serial order (apart 0)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05j
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ranking.py -x -q -m gpu -k "reuse or graph or bench_configuration or sharded" 2>&1 | tail -3 | tee $O/tests.txt
for rep in 1 2; do
for d in uniform zipf; do
  for v in 0 front mid late; do
    n=fm_${d}_${v}_$rep
    ap=1; [ $v = 0 ] && ap=0
    RECBOX_AMD_FM_REZERO_AT=$v RECBOX_AMD_FM_SORT_APART=$ap timeout 200 python bench.py --steps 50 --warmup 10 --no-extra-configs --no-cpu-baseline --dist $d > $O/bench_$n.json 2> $O/bench_$n.err
    python - <<PY | tee -a $O/ab.txt
import json
try:
    d=json.loads([l for l in open('$O/bench_$n.json') if l.startswith('{')][-1]); r=d['roofline']
    print('%-10s %-8s rep $rep  ms_per_step %.4f  fwd %.1f us' % ('$d', '$v', d['ms_per_step'], r['kernel_ms']*1e3))
except Exception as e:
    print('$n', 'failed', e); print(open('$O/bench_$n.err').read()[-1500:])
PY
  done
done
done
for v in mid late; do
(cd /tmp && RECBOX_AMD_FM_REZERO_AT=$v timeout 300 rocprofv3 --kernel-trace -f csv -d $GRAFT_REPO_ROOT/$O/prof$v -o fm -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --no-extra-configs --no-cpu-baseline > /dev/null 2>&1)
python profiles/timeline.py $(find $O/prof$v -name "*kernel_trace.csv" | head -1) compact_ids 30 > $O/fm_apart_${v}_replay_timeline.txt 2>&1; cat $O/fm_apart_${v}_replay_timeline.txt
find $O/prof$v -name "*.csv" -size +4000k -delete
done
