#!/bin/bash
# round 5: weight gradients of the tower layers on a side stream beside whatever consumes dx (config.dw_beside_lookup, now also
# in ops._Linear) on / off: the tower / model tests, then every model bench line, three processes each
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05db2
mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -3 | tee $O/tests.txt
for rep in 1 2 3; do
for cfg in deepfm youtubednn sasrec; do
  for v in 1 0; do
    n=${cfg}_beside${v}_$rep
    RECBOX_AMD_DW_BESIDE=$v timeout 300 python bench.py --config $cfg --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_$n.json 2> $O/bench_$n.err
    python - <<PY | tee -a $O/ab.txt
import json
try:
    d=json.loads([l for l in open('$O/bench_$n.json') if l.startswith('{')][-1])
    print('%-32s ms_per_step %.4f' % ('$n', d['ms_per_step']))
except Exception as e:
    print('$n', 'failed', e); print(open('$O/bench_$n.err').read()[-1500:])
PY
  done
done
done
for cfg in deepfm youtubednn; do
  for v in 1 0; do
    n=${cfg}_sharded1_beside${v}
    RECBOX_AMD_DW_BESIDE=$v timeout 300 python bench.py --config $cfg --force-sharded --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_$n.json 2> $O/bench_$n.err
    python -c "
import json
d=json.loads([l for l in open('$O/bench_$n.json') if l.startswith('{')][-1])
print('%-32s ms_per_step %.4f' % ('$n', d['ms_per_step']))" | tee -a $O/ab.txt
  done
done
