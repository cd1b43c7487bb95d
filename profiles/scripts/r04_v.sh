#!/bin/bash
# the segment reduce with every load of a batch issued before the first use (RBX_REDUCE_BATCHED): tests and the four configs
out=/root/repo/gpurun_out/r04v
mkdir -p $out
cd /root/repo
timeout 1500 python -m pytest tests -q -m gpu -x > $out/tests.log 2>&1; tail -2 $out/tests.log
B="--steps 50 --warmup 10 --no-extra-configs --no-cpu-baseline"
for i in 1 2; do timeout 300 python bench.py $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fm', round(d['ms_per_step'],4), d['roofline']['frac'])"; done
timeout 300 python bench.py $B --dist zipf 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fm zipf', round(d['ms_per_step'],4))"
for cfg in youtubednn deepfm sasrec; do timeout 400 python bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg', round(d['ms_per_step'],4))"; done
timeout 300 python bench.py --force-sharded --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fm sharded1', round(d['ms_per_step'],4))"
