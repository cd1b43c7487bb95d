#!/bin/bash
out=/root/repo/gpurun_out/r2ln
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
cd /root/repo
for v in default ln2 ln4; do
  if [ $v = default ]; then unset RECBOX_HIP_LIB; else export RECBOX_HIP_LIB=/root/repo/recbox_amd/lib/variants/$v.so; fi
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $out/prof -o b -- python /root/repo/bench.py --config sasrec --no-cpu-baseline --steps 10 --warmup 3 > $out/prof_$v.log 2>&1)
  python profiles/topk.py $(find $out/prof -name "*.db" | head -1) 40 > $out/stats_$v.txt
  rm -rf $out/prof
  echo $v; grep "ln_fwd\|\"ms_per_step\"" $out/stats_$v.txt $out/prof_$v.log | cut -c1-140 | head -3; grep -o '"ms_per_step": [0-9.]*' $out/prof_$v.log
done
