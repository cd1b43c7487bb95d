#!/bin/bash
# round 5: the re-zero of the generic lookup's persistent gradients throttled to N workgroups (it has a millisecond of slack beside
# the towers and slows their column reductions 3-8x at full width): YoutubeDNN / DeepFM bench, two rounds
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05rz
mkdir -p $O
for rep in 1 2; do
for cfg in youtubednn deepfm; do
for w in 0 1024 512 256 128 64; do
  RBX_REZERO_WGS=$w timeout 300 python bench.py --config $cfg --steps 30 --warmup 5 --no-cpu-baseline > $O/b.json 2> $O/b.err
  python -c "
import json
d=json.loads([l for l in open('$O/b.json') if l.startswith('{')][-1])
print('${cfg}_rezero_wgs${w}_$rep  ms_per_step %.4f' % d['ms_per_step'])" | tee -a $O/ab.txt
done
done
done
