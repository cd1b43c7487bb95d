#!/bin/bash
# round 5: rbx_fm_front (re-zero + id compaction in one launch) on / off, three processes each; its test and the FM tests;
# tower layer 1 by row pitch; one replayed timeline with the front launch
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05s
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ranking.py -q -m gpu -x 2>&1 | tail -6 | tee $O/tests.txt
for rep in 1 2 3; do
for d in uniform zipf; do
  for fr in 1 0; do
    n=fm_${d}_front${fr}_$rep
    RECBOX_AMD_FM_FRONT=$fr timeout 200 python bench.py --steps 100 --warmup 10 --no-extra-configs --no-cpu-baseline --dist $d > $O/bench_$n.json 2> $O/bench_$n.err
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
timeout 300 python profiles/ubench/gemm_pitch.py > $O/gemm_pitch.txt 2>&1; cat $O/gemm_pitch.txt
out=$GRAFT_REPO_ROOT/$O
rm -rf $out/prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $out/prof -o b -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra-configs --steps 20 --warmup 5 > $out/prof_fm.log 2>&1)
db=$(find $out/prof -name "*.db" | head -1)
python profiles/topk.py $db 30 > $out/fm_kernel_stats.txt
python profiles/timeline.py $db fm_front 30 > $out/fm_replay_timeline.txt 2>&1
rm -rf $out/prof
cat $out/fm_replay_timeline.txt
