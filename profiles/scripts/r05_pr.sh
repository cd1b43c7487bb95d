#!/bin/bash
# round 5: the generic lookup's re-zero + id sort on a side stream of another HIP priority (lab switch), YoutubeDNN / DeepFM
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05pr
mkdir -p $O
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range())" | tee $O/ab.txt
for rep in 1 2; do
for cfg in youtubednn deepfm; do
  for pr in none 0 1 2 -1 -2; do
    if [ $pr = none ]; then unset RECBOX_AMD_SIDE_PRIORITY; else export RECBOX_AMD_SIDE_PRIORITY=$pr; fi
    timeout 300 python bench.py --config $cfg --steps 30 --warmup 5 --no-cpu-baseline > $O/b.json 2> $O/b.err
    python -c "
import json
try:
    d=json.loads([l for l in open('$O/b.json') if l.startswith('{')][-1])
    print('${cfg}_priority_${pr}_$rep  ms_per_step %.4f' % d['ms_per_step'])
except Exception as e:
    print('${cfg}_priority_${pr}_$rep failed', open('$O/b.err').read()[-300:])" | tee -a $O/ab.txt
  done
done
done
