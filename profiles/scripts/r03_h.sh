#!/bin/bash
out=/root/repo/gpurun_out/r03
mkdir -p $out
export TMPDIR=/tmp
cd /root/repo
python -m pytest tests/test_gpu_ranking.py tests/test_gpu_cabi_vs_c_oracle.py tests/test_gpu_edge_cases.py -x -q -m gpu 2>&1 | tail -4 > $out/h_tests.log; cat $out/h_tests.log
ms() { python -c "
import json,sys
d=json.loads(open('$1').readline()); print('$2', round(d['ms_per_step'],4), 'fwd as run', round(d['roofline']['kernel_ms']*1e3,1))"; }
for rep in 1 2; do for mode in split side; do for vmax in 4096 16384; do
  RECBOX_AMD_FM_IDS_WORK=$mode RBX_FM_TIER_A_VMAX=$vmax python bench.py --no-cpu-baseline > $out/h_bench_${mode}_$vmax.json 2>/dev/null; ms $out/h_bench_${mode}_$vmax.json "ids_work=$mode vmax=$vmax"
done; done; done
prof() { # name, env, bench args
  rm -rf $out/prof
  (cd /tmp && env $2 timeout 900 rocprofv3 --kernel-trace --stats -d $out/prof -o b -- python /root/repo/bench.py --no-cpu-baseline $3 > $out/prof_$1.log 2>&1)
  db=$(find $out/prof -name "*.db" | head -1)
  python profiles/topk.py $db 40 > $out/$1_kernel_stats.txt
  python profiles/timeline.py $db compact_ids 30 > $out/$1_replay_timeline.txt 2>&1
  rm -rf $out/prof
}
prof h_fm_split "RECBOX_AMD_FM_IDS_WORK=split" ""
cat $out/h_fm_split_replay_timeline.txt
