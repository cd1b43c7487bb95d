#!/bin/bash
# round 5: the fused FM step with the large tables' sort "apart" (re-zero beside compaction + radix passes; only the last scatter
# waits for it) against the serial order -- tests of the fused op first, then A/B on one box (uniform, Zipf), then the replay timeline
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05i
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ranking.py tests/test_gpu_optim.py tests/test_gpu_cabi_vs_c_oracle.py tests/test_gpu_edge_cases.py -x -q -m gpu 2>&1 | tail -5 | tee $O/tests.txt
for rep in 1 2; do
for d in uniform zipf; do
  for ap in 1 0; do
    n=fm_${d}_apart${ap}_$rep
    RECBOX_AMD_FM_SORT_APART=$ap timeout 200 python bench.py --steps 50 --warmup 10 --no-extra-configs --no-cpu-baseline --dist $d > $O/bench_$n.json 2> $O/bench_$n.err
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
for ap in 1 0; do
(cd /tmp && RECBOX_AMD_FM_SORT_APART=$ap timeout 300 rocprofv3 --kernel-trace -f csv -d $GRAFT_REPO_ROOT/$O/prof$ap -o fm -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --no-extra-configs --no-cpu-baseline > /dev/null 2>&1)
python profiles/timeline.py $(find $O/prof$ap -name "*kernel_trace.csv" | head -1) compact_ids 30 > $O/fm_apart${ap}_replay_timeline.txt 2>&1; cat $O/fm_apart${ap}_replay_timeline.txt
find $O/prof$ap -name "*.csv" -size +4000k -delete
done
