#!/bin/bash
# tier threshold sweep x one / two chains, adaptive reduce chunk; full ranking tests first
out=/root/repo/gpurun_out/r03
mkdir -p $out
export TMPDIR=/tmp
cd /root/repo
python -m pytest tests/test_gpu_ranking.py tests/test_gpu_cabi_vs_c_oracle.py tests/test_gpu_edge_cases.py tests/test_gpu_matching.py -x -q -m gpu 2>&1 | tail -4 > $out/d_tests.log; cat $out/d_tests.log
ms() { python -c "
import json,sys
d=json.loads(open('$1').readline()); print('$2', round(d['ms_per_step'],4), 'fwd as run', round(d['roofline']['kernel_ms']*1e3,1))"; }
for vmax in 0 128 2200 4096 16384; do for two in 1 0; do
  RBX_FM_TIER_A_VMAX=$vmax RECBOX_AMD_FM_TWO_CHAINS=$two python bench.py --no-cpu-baseline > $out/d_bench_${vmax}_$two.json 2>/dev/null; ms $out/d_bench_${vmax}_$two.json "vmax=$vmax two_chains=$two"
done; done
for vmax in 2200 16384; do
  RECBOX_HIP_LIB=/root/repo/recbox_amd/lib/variants/rb10.so RBX_FM_TIER_A_VMAX=$vmax python bench.py --no-cpu-baseline > $out/d_bench_rb10_$vmax.json 2>/dev/null; ms $out/d_bench_rb10_$vmax.json "rb10 vmax=$vmax"
done
for c in 16 24 32 40; do
  RBX_REDUCE_CHUNK=$c python bench.py --no-cpu-baseline > $out/d_bench_chunk$c.json 2>/dev/null; ms $out/d_bench_chunk$c.json "chunk=$c"
done
prof() { # name, env, bench args
  rm -rf $out/prof
  (cd /tmp && env $2 timeout 900 rocprofv3 --kernel-trace --stats -d $out/prof -o b -- python /root/repo/bench.py --no-cpu-baseline $3 > $out/prof_$1.log 2>&1)
  db=$(find $out/prof -name "*.db" | head -1)
  python profiles/topk.py $db 40 > $out/$1_kernel_stats.txt
  python profiles/timeline.py $db compact_ids 30 > $out/$1_replay_timeline.txt 2>&1
  rm -rf $out/prof
}
prof d_fm_2200 "RBX_FM_TIER_A_VMAX=2200" ""
cat $out/d_fm_2200_replay_timeline.txt
prof d_fm_16384 "RBX_FM_TIER_A_VMAX=16384" ""
cat $out/d_fm_16384_replay_timeline.txt
