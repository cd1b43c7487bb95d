#!/bin/bash
# tier-A FM step after the kernel fixes + two chains: tests, bench, timeline; tier-B reduce ablations / radix / chunk variants
out=/root/repo/gpurun_out/r03
mkdir -p $out
export TMPDIR=/tmp
cd /root/repo
python -m pytest tests/test_gpu_ranking.py tests/test_gpu_cabi_vs_c_oracle.py tests/test_gpu_edge_cases.py -x -q -m gpu 2>&1 | tail -4 > $out/c_tests.log; cat $out/c_tests.log
ms() { python -c "
import json,sys
d=json.loads(open('$1').readline()); print('$2', round(d['ms_per_step'],4), 'fwd as run', round(d['roofline']['kernel_ms']*1e3,1))"; }
python bench.py --no-cpu-baseline > $out/c_bench.json 2>$out/c_bench.err; ms $out/c_bench.json two_chains
RECBOX_AMD_FM_TWO_CHAINS=0 python bench.py --no-cpu-baseline > $out/c_bench_one.json 2>/dev/null; ms $out/c_bench_one.json one_chain
for v in rb10 chunk16 chunk24 abl1 abl2 abl4 abl8; do
  RECBOX_HIP_LIB=/root/repo/recbox_amd/lib/variants/$v.so python bench.py --no-cpu-baseline > $out/c_bench_$v.json 2>/dev/null; ms $out/c_bench_$v.json $v
done
prof() { # name, env, bench args
  rm -rf $out/prof
  (cd /tmp && env $2 timeout 900 rocprofv3 --kernel-trace --stats -d $out/prof -o b -- python /root/repo/bench.py --no-cpu-baseline $3 > $out/prof_$1.log 2>&1)
  db=$(find $out/prof -name "*.db" | head -1)
  python profiles/topk.py $db 40 > $out/$1_kernel_stats.txt
  python profiles/timeline.py $db rezero_rows 30 > $out/$1_replay_timeline.txt 2>&1
  rm -rf $out/prof
}
prof c_fm "RBX_X=1" ""
cat $out/c_fm_replay_timeline.txt
for v in abl1 abl2 abl4 abl8 chunk16; do
prof c_$v "RECBOX_HIP_LIB=/root/repo/recbox_amd/lib/variants/$v.so" ""
grep -E "segment_reduce|ta_reduce|ta_final" $out/c_${v}_replay_timeline.txt
done
