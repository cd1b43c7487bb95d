#!/bin/bash
# round 5: DeepFM's input stage with dW1 / db1 / the first-order head's gradients on a side stream beside the embedding lookup's
# backward (config.dw_beside_lookup) on / off: tests, bench A/B, one timeline
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05db
mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_matching.py tests/test_gpu_ranking.py -q -m gpu -x -k "deepfm or DeepFM or tower or graph" 2>&1 | tail -3 | tee $O/tests.txt
for rep in 1 2 3; do
  for v in 1 0; do
    n=deepfm_beside${v}_$rep
    RECBOX_AMD_DW_BESIDE=$v timeout 300 python bench.py --config deepfm --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_$n.json 2> $O/bench_$n.err
    python - <<PY | tee -a $O/ab.txt
import json
try:
    d=json.loads([l for l in open('$O/bench_$n.json') if l.startswith('{')][-1])
    print('%-28s ms_per_step %.4f' % ('$n', d['ms_per_step']))
except Exception as e:
    print('$n', 'failed', e); print(open('$O/bench_$n.err').read()[-1500:])
PY
  done
done
out=$GRAFT_REPO_ROOT/$O
rm -rf $out/prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $out/prof -o b -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --config deepfm --steps 20 --warmup 5 > $out/prof.log 2>&1)
db=$(find $out/prof -name "*.db" | head -1)
python profiles/timeline.py $db "embed_fwd_kernel<16" 15 > $out/deepfm_replay_timeline.txt 2>&1
rm -rf $out/prof
tail -22 $out/deepfm_replay_timeline.txt | cut -c1-110
