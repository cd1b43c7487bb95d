#!/bin/bash
# FM step: re-zero from a copy of the sorted ids on the main stream (RECBOX_AMD_REZERO_COPY), reduce + fix-ups on the sort's stream
out=/root/repo/gpurun_out/r2rz
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp PYTHONPATH=/root/repo
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_ranking.py tests/test_gpu_cabi_vs_c_oracle.py -x -q -m gpu 2>&1 | tail -3
for v in "1 1" "1 0" "0 0" "1 1" "0 0"; do
  set -- $v
  echo "REZERO_COPY=$1 REDUCE_ON_SORT_STREAM=$2"
  RECBOX_AMD_REZERO_COPY=$1 RECBOX_AMD_REDUCE_ON_SORT_STREAM=$2 timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | cut -c1-330 | grep -o '"ms_per_step": [0-9.]*'
done
for v in "1 1"; do
  set -- $v
  (cd /tmp && RECBOX_AMD_REZERO_COPY=$1 RECBOX_AMD_REDUCE_ON_SORT_STREAM=$2 timeout 400 rocprofv3 --kernel-trace --stats -d $out/prof$1$2 -o b -- python /root/repo/bench.py --no-cpu-baseline > $out/prof$1$2.log 2>&1)
  python profiles/timeline.py $(find $out/prof$1$2 -name "*.db" | head -1) fm_fused_fwd 30 > $out/timeline_$1$2.txt 2>&1
  python profiles/topk.py $(find $out/prof$1$2 -name "*.db" | head -1) 24 > $out/kernel_stats_$1$2.txt
  rm -rf $out/prof$1$2
  head -30 $out/timeline_$1$2.txt | cut -c1-130
done
