#!/bin/bash
# round 5: DeepFM's FM / first-order pass on the side stream beside the tower's first GEMM (part of config.dw_beside_lookup):
# current tree (ff1) against the tree of the evidence run (the pass behind the GEMM: RECBOX_AMD_DW_BESIDE=1 there too), via git stash? no --
# against RECBOX_AMD_DW_BESIDE=0 for reference; DeepFM tests
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05ff
mkdir -p $O
true
for rep in 1 2 3 4; do
  for v in 1 3 0; do
    n=deepfm_beside${v}_$rep
    RECBOX_AMD_DW_BESIDE=$v timeout 300 python bench.py --config deepfm --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_$n.json 2> $O/bench_$n.err
    python -c "
import json
d=json.loads([l for l in open('$O/bench_$n.json') if l.startswith('{')][-1]); r=d['roofline']
print('%-32s ms_per_step %.4f  layer-1 GEMM %.1f us' % ('$n', d['ms_per_step'], r['kernel_ms']*1e3))" | tee -a $O/ab.txt
  done
done
