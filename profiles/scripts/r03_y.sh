#!/bin/bash
cd /root/repo
for i in 1 2; do
timeout 600 python bench.py --no-cpu-baseline --no-extra-configs 2>/tmp/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('fm', round(d['ms_per_step'],4), round(d['roofline']['frac'],3))"
grep -c AccumulateGrad /tmp/err.txt
done
timeout 600 python bench.py --config youtubednn --no-cpu-baseline 2>/tmp/err2.txt | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('yt', round(d['ms_per_step'],4))"
grep -c "AccumulateGrad" /tmp/err2.txt
