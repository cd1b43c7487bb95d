#!/bin/bash
# round 5: quick loop for the chained sort -- sort-heavy tests, FM bench lines (uniform, Zipf), replay timeline
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05l
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ranking.py tests/test_gpu_cabi_vs_c_oracle.py tests/test_gpu_edge_cases.py -x -q -m gpu 2>&1 | tail -3 | tee $O/tests.txt
for rep in 1 2 3; do
for d in uniform zipf; do
    n=fm_${d}_$rep
    timeout 200 python bench.py --steps 100 --warmup 10 --no-extra-configs --no-cpu-baseline --dist $d > $O/bench_$n.json 2> $O/bench_$n.err
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
timeout 600 python bench.py --config deepfm --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_deepfm.json 2>/dev/null
python -c "import json; d=json.loads([l for l in open('$O/bench_deepfm.json') if l.startswith('{')][-1]); print('deepfm', round(d['ms_per_step'],4))" | tee -a $O/ab.txt
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -f csv -d $GRAFT_REPO_ROOT/$O/prof -o fm -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --no-extra-configs --no-cpu-baseline > /dev/null 2>&1)
python profiles/timeline.py $(find $O/prof -name "*kernel_trace.csv" | head -1) compact_ids 30 > $O/fm_replay_timeline.txt 2>&1; cat $O/fm_replay_timeline.txt
find $O/prof -name "*.csv" -size +4000k -delete
