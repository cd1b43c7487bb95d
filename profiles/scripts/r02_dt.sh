#!/bin/bash
# FM forward with the ids dtype as a template parameter (one decode path per launch) against the per-feature dtype switch
out=/root/repo/gpurun_out/r2dt
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_ranking.py tests/test_gpu_cabi_vs_c_oracle.py -x -q -m gpu 2>&1 | tail -4
for v in 1 0 1 0; do echo "RBX_FM_FAST_DTYPE=$v"; RBX_FM_FAST_DTYPE=$v timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | cut -c1-330; done
for v in 1 0; do
(cd /tmp && RBX_FM_FAST_DTYPE=$v timeout 400 rocprofv3 --kernel-trace --stats -d $out/prof$v -o b -- python /root/repo/bench.py --no-cpu-baseline > $out/prof$v.log 2>&1)
python profiles/topk.py $(find $out/prof$v -name "*.db" | head -1) 12 > $out/kernel_stats_dt$v.txt
rm -rf $out/prof$v
head -14 $out/kernel_stats_dt$v.txt | cut -c1-150
done
