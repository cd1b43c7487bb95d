#!/bin/bash
out=/root/repo/gpurun_out/r3u
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
cd /root/repo
for v in "RBX_DW64_WGS=512" "RBX_DW64_WGS=256" "RBX_DW64_WGS=512 RBX_DW64_ABL=1"; do
(cd /tmp && env $v timeout 600 rocprofv3 --kernel-trace --stats -d $out/prof -o b -- python /root/repo/profiles/ubench/tall_gemm.py > $out/prof.log 2>&1)
echo "== $v"; python profiles/topk.py $(find $out/prof -name "*.db" | head -1) 10 | grep -E "tall_dw|k64n64"
rm -rf $out/prof
done
