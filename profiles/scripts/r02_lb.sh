#!/bin/bash
# radix sort: look-back scatter passes (RBX_SORT_LOOKBACK) against the histogram + scan kernels in front of every pass
out=/root/repo/gpurun_out/r2lb
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp PYTHONPATH=/root/repo
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_ranking.py tests/test_gpu_cabi_vs_c_oracle.py tests/test_gpu_edge_cases.py -x -q -m gpu 2>&1 | tail -4
for v in 1 0 1 0; do echo "RBX_SORT_LOOKBACK=$v"; RBX_SORT_LOOKBACK=$v timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'; done
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $out/prof -o b -- python /root/repo/bench.py --no-cpu-baseline > $out/prof.log 2>&1)
python profiles/timeline.py $(find $out/prof -name "*.db" | head -1) rezero_rows 30 > $out/timeline.txt 2>&1
python profiles/topk.py $(find $out/prof -name "*.db" | head -1) 16 > $out/kernel_stats.txt
rm -rf $out/prof
head -26 $out/timeline.txt | cut -c1-130
