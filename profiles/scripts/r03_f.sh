#!/bin/bash
out=/root/repo/gpurun_out/r03
mkdir -p $out
export TMPDIR=/tmp
cd /root/repo
ms() { python -c "
import json,sys
d=json.loads(open('$1').readline()); print('$2', round(d['ms_per_step'],4), 'fwd as run', round(d['roofline']['kernel_ms']*1e3,1))"; }
for vmax in 2200 4096 16384; do for on in side main; do
  RBX_FM_TIER_A_VMAX=$vmax RECBOX_AMD_FM_TIER_A_ON=$on python bench.py --no-cpu-baseline > $out/f_bench_${vmax}_$on.json 2>/dev/null; ms $out/f_bench_${vmax}_$on.json "vmax=$vmax tierA_on=$on"
done; done
prof() { # name, env, bench args
  rm -rf $out/prof
  (cd /tmp && env $2 timeout 900 rocprofv3 --kernel-trace --stats -d $out/prof -o b -- python /root/repo/bench.py --no-cpu-baseline $3 > $out/prof_$1.log 2>&1)
  db=$(find $out/prof -name "*.db" | head -1)
  python profiles/topk.py $db 40 > $out/$1_kernel_stats.txt
  python profiles/timeline.py $db compact_ids 30 > $out/$1_replay_timeline.txt 2>&1
  rm -rf $out/prof
}
prof f_fm_4096_side "RBX_FM_TIER_A_VMAX=4096 RECBOX_AMD_FM_TIER_A_ON=side" ""
cat $out/f_fm_4096_side_replay_timeline.txt
