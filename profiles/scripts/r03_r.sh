#!/bin/bash
# FM step: replay timeline with the current defaults
out=/root/repo/gpurun_out/r3r
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
cd /root/repo
timeout 600 python bench.py --no-cpu-baseline --no-extra-configs > $out/bench_fm.json 2>/dev/null
python -c "
import json
d=json.loads(open('$out/bench_fm.json').readline()); print('fm', round(d['ms_per_step'],4), d['roofline']['frac'])"
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $out/prof -o b -- python /root/repo/bench.py --no-cpu-baseline --no-extra-configs > $out/prof_fm.log 2>&1)
db=$(find $out/prof -name "*.db" | head -1)
python profiles/timeline.py $db compact_ids 30 > $out/fm_replay_timeline.txt 2>&1
python profiles/timeline.py $db compact_ids 40 > $out/fm_replay_timeline2.txt 2>&1
rm -rf $out/prof
cat $out/fm_replay_timeline.txt | cut -c1-120
