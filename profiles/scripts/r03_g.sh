#!/bin/bash
out=/root/repo/gpurun_out/r03
mkdir -p $out
export TMPDIR=/tmp
cd /root/repo
python -m pytest tests/test_gpu_ranking.py tests/test_gpu_cabi_vs_c_oracle.py tests/test_gpu_edge_cases.py -x -q -m gpu 2>&1 | tail -4 > $out/g_tests.log; cat $out/g_tests.log
ms() { python -c "
import json,sys
d=json.loads(open('$1').readline()); print('$2', round(d['ms_per_step'],4), 'fwd as run', round(d['roofline']['kernel_ms']*1e3,1))"; }
for vmax in 4096 16384; do for lds in 1 0; do
  RBX_FM_TIER_A_VMAX=$vmax RBX_TA_LDS=$lds python bench.py --no-cpu-baseline > $out/g_bench_${vmax}_$lds.json 2>/dev/null; ms $out/g_bench_${vmax}_$lds.json "vmax=$vmax lds=$lds"
done; done
for vmax in 4096 16384; do
echo "== RBX_FM_TIER_A_VMAX=$vmax"
RBX_FM_TIER_A_VMAX=$vmax python profiles/ubench/fm_bwd_parts.py 20 2>&1 | grep -v "Warn\|amdgpu.ids"
rm -rf $out/prof
(cd /tmp && RBX_FM_TIER_A_VMAX=$vmax rocprofv3 --kernel-trace --stats -d $out/prof -o b -- python /root/repo/profiles/ubench/fm_bwd_parts.py 20 > $out/prof_parts.log 2>&1)
python profiles/topk.py $(find $out/prof -name "*.db" | head -1) > $out/g_parts_${vmax}_kernel_stats.txt
grep "ta_\|segment_" $out/g_parts_${vmax}_kernel_stats.txt | cut -c1-110
done
rm -rf $out/prof
