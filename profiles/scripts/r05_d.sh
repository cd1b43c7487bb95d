#!/bin/bash
# round 5: the quad forward in the product -- its tests, the FM tests, the bench with the kernel on and off
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05d
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ranking.py -x -q -m gpu -k "quad or fm or FM or bench_configuration" > $O/tests.log 2>&1
echo "tests exit $?"; tail -5 $O/tests.log
for q in 1 0; do
  RBX_FM_QUAD=$q timeout 300 python bench.py --steps 30 --warmup 5 --no-extra-configs --no-cpu-baseline > $O/bench_fm_quad$q.json 2> $O/bench_fm_quad$q.err
  echo "quad=$q exit $?"; python - <<PY
import json
d=json.loads([l for l in open('$O/bench_fm_quad$q.json') if l.startswith('{')][-1])
r=d['roofline']
print('ms_per_step', d['ms_per_step'], 'kernel', r.get('kernel'), 'kernel_ms', r.get('kernel_ms'), 'frac', r.get('frac'), 'alone', r.get('kernel_ms_alone'))
PY
done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/$O/prof -o fm -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --no-extra-configs --no-cpu-baseline > /dev/null 2>&1)
python profiles/topk.py $(find $O/prof -name "*kernel_stats.csv" | head -1) 25 2>&1 | head -40 > $O/fm_kernel_stats.txt; cat $O/fm_kernel_stats.txt
find $O/prof -name "*.csv" -size +1000k -delete
