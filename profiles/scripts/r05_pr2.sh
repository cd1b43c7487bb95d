#!/bin/bash
# round 5: the captured step's own stream at HIGH HIP priority (-1; side streams stay at 0 = low): every bench line, lab switch
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05pr2
mkdir -p $O
for rep in 1 2 3; do
for cfg in fm youtubednn deepfm sasrec; do
  for pr in none -1; do
    if [ $pr = none ]; then unset RECBOX_AMD_CAPTURE_PRIORITY; else export RECBOX_AMD_CAPTURE_PRIORITY=$pr; fi
    extra="--config $cfg --steps 30 --warmup 5"; [ $cfg = fm ] && extra="--no-extra-configs --steps 100 --warmup 10"
    timeout 300 python bench.py $extra --no-cpu-baseline > $O/b.json 2> $O/b.err
    python -c "
import json
try:
    d=json.loads([l for l in open('$O/b.json') if l.startswith('{')][-1])
    print('${cfg}_capture_priority_${pr}_$rep  ms_per_step %.4f' % d['ms_per_step'])
except Exception as e:
    print('${cfg}_capture_priority_${pr}_$rep failed', open('$O/b.err').read()[-300:])" | tee -a $O/ab.txt
  done
done
done
