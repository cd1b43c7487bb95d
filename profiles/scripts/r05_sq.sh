#!/bin/bash
# round 5: SASRec block: the in-projection weight gradients (rbx_seqblock_inproj_dw) on the side stream (lab switch) on / off
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05sq
mkdir -p $O
RECBOX_AMD_SEQ_DW_BESIDE=1 timeout 900 python -m pytest tests/test_gpu_seqblock.py tests/test_gpu_matching.py -q -m gpu -x -k "seq or sasrec or SASRec" 2>&1 | tail -2
for rep in 1 2 3; do
  for v in 1 0; do
    RECBOX_AMD_SEQ_DW_BESIDE=$v timeout 300 python bench.py --config sasrec --steps 20 --warmup 5 --no-cpu-baseline > $O/b.json 2> $O/b.err
    python -c "
import json
d=json.loads([l for l in open('$O/b.json') if l.startswith('{')][-1])
print('sasrec_seq_dw_beside${v}_$rep  ms_per_step %.4f' % d['ms_per_step'])" | tee -a $O/ab.txt
  done
done
