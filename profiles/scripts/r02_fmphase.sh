#!/bin/bash
# the FM bench line + its rocprofv3 kernel stats, with the forward kernel averaged per phase of the bench (profiles/kernel_slice.py)
out=/root/repo/gpurun_out/r2final
mkdir -p $out
export TMPDIR=/tmp
cd /root/repo
timeout 600 python bench.py > $out/bench_fm.json 2>$out/bench_fm.err; cut -c1-300 $out/bench_fm.json
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $out/prof -o b -- python /root/repo/bench.py --no-cpu-baseline > $out/prof_fm.log 2>&1)
db=$(find $out/prof -name "*.db" | head -1)
python profiles/topk.py $db 48 > $out/fm_kernel_stats.txt
python profiles/timeline.py $db rezero_rows 30 > $out/fm_replay_timeline.txt 2>&1
python profiles/kernel_slice.py $db fm_fused_fwd 24 60 > $out/fm_fwd_kernel_by_phase.txt 2>&1
rm -rf $out/prof
cat $out/fm_fwd_kernel_by_phase.txt
python -c "
import json
d=json.loads(open('$out/bench_fm.json').readline()); r=d['roofline']
print('bench: kernel_ms', r['kernel_ms'], 'alone', r.get('kernel_ms_alone'), 'warm', r.get('kernel_ms_warm'), 'ms_per_step', d['ms_per_step'])"
