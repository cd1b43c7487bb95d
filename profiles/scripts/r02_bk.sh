#!/bin/bash
out=/root/repo/gpurun_out/r2bk
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
cd /root/repo
for v in default bk32; do
  if [ $v = default ]; then unset RECBOX_HIP_LIB; else export RECBOX_HIP_LIB=/root/repo/recbox_amd/lib/variants/$v.so; fi
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $out/prof -o b -- python /root/repo/bench.py --config sasrec --no-cpu-baseline --steps 10 --warmup 3 > $out/prof_$v.log 2>&1)
  python profiles/topk.py $(find $out/prof -name "*.db" | head -1) 40 > $out/stats_$v.txt
  rm -rf $out/prof
  echo $v $(grep -o '"ms_per_step": [0-9.]*' $out/prof_$v.log); grep "gemm_f32\|tall_dw" $out/stats_$v.txt | cut -c1-140
  timeout 600 python bench.py --config deepfm --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$v deepfm step_ms', round(d['ms_per_step'],3), 'gemm', round(d['roofline']['kernel_ms'],4))"
done
