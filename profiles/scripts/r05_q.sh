#!/bin/bash
# round 5: full GPU suite on the current code + sharded world-of-one lines
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05q
mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -6 | tee $O/tests.txt
for cfg in fm youtubednn deepfm; do
  timeout 300 python bench.py --config $cfg --force-sharded --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_${cfg}_sharded1.json 2> $O/bench_${cfg}_sharded1.err
  python -c "import json; d=json.loads([l for l in open('$O/bench_${cfg}_sharded1.json') if l.startswith('{')][-1]); print('${cfg}_sharded1', round(d['ms_per_step'],4))" | tee -a $O/ab.txt
done
