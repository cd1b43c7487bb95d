#!/bin/bash
# round 5: ops.linear's weight gradients beside only below a FLOP count of the dW GEMM (1e10: DeepFM's 400 x 400 layers stay in line,
# YoutubeDNN's layers go beside) against every layer (1e30)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05fl
mkdir -p $O
for rep in 1 2 3; do
for cfg in deepfm youtubednn; do
  for fl in 1e30 1e10; do
    RECBOX_AMD_DW_BESIDE_FLOPS=$fl timeout 300 python bench.py --config $cfg --steps 30 --warmup 5 --no-cpu-baseline > $O/b.json 2> $O/b.err
    python -c "
import json
d=json.loads([l for l in open('$O/b.json') if l.startswith('{')][-1])
print('${cfg}_flops${fl}_$rep  ms_per_step %.4f' % d['ms_per_step'])" | tee -a $O/ab.txt
  done
done
done
