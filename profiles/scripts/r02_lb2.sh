#!/bin/bash
out=/root/repo/gpurun_out/r2lb2
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp PYTHONPATH=/root/repo
cd /root/repo
for v in 1 0; do RBX_SORT_LOOKBACK=$v timeout 200 python profiles/sort_ubench.py 2>&1 | tail -1; done
(cd /tmp && RBX_SORT_LOOKBACK=1 timeout 300 rocprofv3 --kernel-trace --stats -d $out/prof -o b -- python /root/repo/profiles/sort_ubench.py > $out/prof.log 2>&1)
python profiles/topk.py $(find $out/prof -name "*.db" | head -1) 8 | cut -c1-130
rm -rf $out/prof
