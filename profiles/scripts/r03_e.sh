#!/bin/bash
out=/root/repo/gpurun_out/r03
mkdir -p $out
export TMPDIR=/tmp
cd /root/repo
for vmax in 16384 2200; do
echo "== RBX_FM_TIER_A_VMAX=$vmax"
RBX_FM_TIER_A_VMAX=$vmax python profiles/ubench/fm_bwd_parts.py 20 2>&1 | grep -v Warn
rm -rf $out/prof
(cd /tmp && RBX_FM_TIER_A_VMAX=$vmax rocprofv3 --kernel-trace --stats -d $out/prof -o b -- python /root/repo/profiles/ubench/fm_bwd_parts.py 20 > $out/prof_parts.log 2>&1)
python profiles/topk.py $(find $out/prof -name "*.db" | head -1) > $out/e_parts_${vmax}_kernel_stats.txt
grep -v "at::\|rocclr" $out/e_parts_${vmax}_kernel_stats.txt | cut -c1-110
done
rm -rf $out/prof
