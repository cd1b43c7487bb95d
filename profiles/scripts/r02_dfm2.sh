#!/bin/bash
out=/root/repo/gpurun_out/r2dfm2
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp PYTHONPATH=/root/repo
cd /root/repo
timeout 300 python bench.py --config deepfm --no-cpu-baseline 2>/dev/null | tee $out/bench_deepfm.json | cut -c1-330
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $out/prof -o b -- python /root/repo/bench.py --config deepfm --no-cpu-baseline > $out/prof.log 2>&1)
python profiles/topk.py $(find $out/prof -name "*.db" | head -1) 40 > $out/deepfm_kernel_stats.txt
rm -rf $out/prof
head -44 $out/deepfm_kernel_stats.txt | cut -c1-150
