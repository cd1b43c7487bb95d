#!/bin/bash
export TMPDIR=/tmp PYTHONPATH=/root/repo
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_matching.py tests/test_gpu_cabi_vs_c_oracle.py -x -q -m gpu -k "cos or youtube or dssm or l2" 2>&1 | tail -3
for i in 1 2; do timeout 400 python bench.py --config youtubednn --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'; done
out=/root/repo/gpurun_out/r2cos; rm -rf $out; mkdir -p $out
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $out/prof -o b -- python /root/repo/bench.py --config youtubednn --no-cpu-baseline --steps 20 --warmup 5 > $out/prof.log 2>&1)
python profiles/topk.py $(find $out/prof -name "*.db" | head -1) 30 > $out/kernel_stats.txt; rm -rf $out/prof
head -30 $out/kernel_stats.txt | cut -c1-130
