#!/bin/bash
# per-variant FM step time + reduce kernel time
out=/root/repo/gpurun_out/r2var
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
cd /root/repo
for v in default "$@"; do
  if [ $v = default ]; then unset RECBOX_HIP_LIB; else export RECBOX_HIP_LIB=/root/repo/recbox_amd/lib/variants/$v.so; fi
  for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$v step_ms', round(d['ms_per_step'],4), 'fwd_ms', round(d['roofline']['kernel_ms'],4))"; done
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $out/prof -o b -- python /root/repo/bench.py --no-cpu-baseline > $out/prof_$v.log 2>&1)
  python profiles/topk.py $(find $out/prof -name "*.db" | head -1) 24 > $out/kernel_stats_$v.txt
  rm -rf $out/prof
  grep "segment_" $out/kernel_stats_$v.txt | cut -c1-130
done
