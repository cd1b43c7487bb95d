#!/bin/bash
# round 5: DeepFM: weight gradients beside -- every tower layer (1) / the input stage only (2) / none (0), four processes each
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05db3
mkdir -p $O
for rep in 1 2 3 4; do
  for v in 1 2 0; do
    n=deepfm_beside${v}_$rep
    RECBOX_AMD_DW_BESIDE=$v timeout 300 python bench.py --config deepfm --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_$n.json 2> $O/bench_$n.err
    python -c "
import json
d=json.loads([l for l in open('$O/bench_$n.json') if l.startswith('{')][-1])
print('%-32s ms_per_step %.4f' % ('$n', d['ms_per_step']))" | tee -a $O/ab.txt
  done
done
