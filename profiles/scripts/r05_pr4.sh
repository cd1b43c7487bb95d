#!/bin/bash
# round 5: EVERY stream of the captured step at high HIP priority (lab switch)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05pr4
mkdir -p $O
for rep in 1 2 3; do
for cfg in fm youtubednn deepfm; do
  for hi in 1 0; do
    extra="--config $cfg --steps 30 --warmup 5"; [ $cfg = fm ] && extra="--no-extra-configs --steps 100 --warmup 10"
    RECBOX_AMD_ALL_HIGH=$hi timeout 300 python bench.py $extra --no-cpu-baseline > $O/b.json 2> $O/b.err
    python -c "
import json
try:
    d=json.loads([l for l in open('$O/b.json') if l.startswith('{')][-1])
    print('${cfg}_all_high${hi}_$rep  ms_per_step %.4f' % d['ms_per_step'])
except Exception as e:
    print('${cfg}_all_high${hi}_$rep failed', open('$O/b.err').read()[-300:])" | tee -a $O/ab.txt
  done
done
done
