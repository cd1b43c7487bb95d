#!/bin/bash
cd /root/repo
for c in youtubednn deepfm sasrec; do for m in "" "--rotate-by-copy"; do timeout 900 python bench.py --config $c --no-cpu-baseline $m 2>/tmp/err_$c.txt | python -c "
import json,sys
l=sys.stdin.readline()
try:
    d=json.loads(l); print('$c $m step_ms', round(d['ms_per_step'],4), 'frac', round(d['roofline']['frac'],3))
except Exception as e:
    print('$c $m FAILED', l[:200]); print(open('/tmp/err_$c.txt').read()[-800:])"; done; done
