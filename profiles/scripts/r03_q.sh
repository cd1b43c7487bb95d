#!/bin/bash
out=/root/repo/gpurun_out/r03
mkdir -p $out
export TMPDIR=/tmp
cd /root/repo
python -m pytest tests/test_gpu_ranking.py tests/test_gpu_optim.py -x -q -m gpu 2>&1 | tail -3
ms() { python -c "
import json,sys
d=json.loads(open('$1').readline()); print('$2', round(d['ms_per_step'],4), 'fwd as run', round(d['roofline']['kernel_ms']*1e3,1))"; }
for rep in 1 2; do
for bs in after_fwd bwd; do for nm in last first; do
  RECBOX_AMD_FM_BLOCKSORT_AT=$bs RECBOX_AMD_FM_NUMERIC=$nm python bench.py --no-cpu-baseline > $out/q_bench_${bs}_$nm.json 2>/dev/null; ms $out/q_bench_${bs}_$nm.json "blocksort_at=$bs numeric=$nm"
done; done; done
rm -rf $out/prof
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $out/prof -o b -- python /root/repo/bench.py --no-cpu-baseline > $out/prof_q.log 2>&1)
python profiles/timeline.py $(find $out/prof -name "*.db" | head -1) compact_ids 30
rm -rf $out/prof
