#!/bin/bash
out=/root/repo/gpurun_out/r03
mkdir -p $out
cd /root/repo
ms() { python -c "
import json,sys
d=json.loads(open('$1').readline()); print('$2', round(d['ms_per_step'],4), 'fwd as run', round(d['roofline']['kernel_ms']*1e3,1))"; }
for rep in 1 2; do
for rz in side main; do for bs in bwd fwd; do
  RECBOX_AMD_FM_REZERO_ON=$rz RECBOX_AMD_FM_BLOCKSORT_AT=$bs python bench.py --no-cpu-baseline > $out/p_bench_${rz}_$bs.json 2>/dev/null; ms $out/p_bench_${rz}_$bs.json "rezero_on=$rz blocksort_at=$bs"
done; done; done
