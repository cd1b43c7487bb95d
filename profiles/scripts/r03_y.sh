#!/bin/bash
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_edge_cases.py tests/test_gpu_matching.py tests/test_gpu_ranking.py -x -q 2>&1 | tail -4
timeout 600 python bench.py --config sasrec --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('sasrec', round(d['ms_per_step'],4))"
timeout 600 python bench.py --no-cpu-baseline --no-extra-configs --dist zipf 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('fm zipf', round(d['ms_per_step'],4))"
RBX_FM_TIER_A=0 timeout 600 python bench.py --no-cpu-baseline --no-extra-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('fm one tier', round(d['ms_per_step'],4))"
timeout 600 python bench.py --no-cpu-baseline --no-extra-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('fm', round(d['ms_per_step'],4))"
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof -o b -- python /root/repo/bench.py --no-cpu-baseline --config sasrec --steps 20 --warmup 5 > /dev/null 2>&1)
python profiles/topk.py $(find /tmp/prof -name "*.db" | head -1) 40 | grep -i "fixup\|segment_reduce"
