#!/bin/bash
# round 5: FM step with re-zero + compaction leading the SIDE stream's chain and the forward kernel first on the current stream
# (RECBOX_AMD_FM_FRONT_SIDE=1) against the default placement; FM tests under it, bench A/B, one timeline
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05fs
mkdir -p $O
export TMPDIR=/tmp
RECBOX_AMD_FM_FRONT_SIDE=1 timeout 900 python -m pytest tests/test_gpu_ranking.py -q -m gpu -x 2>&1 | tail -3 | tee $O/tests.txt
for rep in 1 2 3; do
for d in uniform zipf; do
  for fs in 1 0; do
    n=fm_${d}_side${fs}_$rep
    RECBOX_AMD_FM_FRONT_SIDE=$fs timeout 200 python bench.py --steps 100 --warmup 10 --no-extra-configs --no-cpu-baseline --dist $d > $O/bench_$n.json 2> $O/bench_$n.err
    python - <<PY | tee -a $O/ab.txt
import json
try:
    d=json.loads([l for l in open('$O/bench_$n.json') if l.startswith('{')][-1]); r=d['roofline']
    print('%-28s ms_per_step %.4f  fwd %.1f us  frac %.3f' % ('$n', d['ms_per_step'], r['kernel_ms']*1e3, r['frac']))
except Exception as e:
    print('$n', 'failed', e); print(open('$O/bench_$n.err').read()[-1500:])
PY
  done
done
done
out=$GRAFT_REPO_ROOT/$O
rm -rf $out/prof
(cd /tmp && RECBOX_AMD_FM_FRONT_SIDE=1 timeout 600 rocprofv3 --kernel-trace --stats -d $out/prof -o b -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra-configs --steps 20 --warmup 5 > $out/prof_fm.log 2>&1)
db=$(find $out/prof -name "*.db" | head -1)
python profiles/timeline.py $db rezero_rows 30 > $out/fm_replay_timeline.txt 2>&1
rm -rf $out/prof
cat $out/fm_replay_timeline.txt
