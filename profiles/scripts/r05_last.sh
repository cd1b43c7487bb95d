#!/bin/bash
# round 5: the full GPU suite, smoke and the driver's bench line on the round's last tree
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/r05last
rm -rf $out; mkdir -p $out
rm -f gpurun_out/parity_errors.txt
timeout 2400 python -m pytest tests -q -m gpu > $out/gpu_tests.log 2>&1
echo "gpu tests exit $?" | tee -a $out/summary.txt; tail -3 $out/gpu_tests.log | tee -a $out/summary.txt
cp gpurun_out/parity_errors.txt $out/parity_errors.txt 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | tee -a $out/summary.txt
timeout 900 python bench.py > $out/bench_fm.json 2> $out/bench_fm.err
python -c "import json; d=json.loads([l for l in open('$out/bench_fm.json') if l.startswith('{')][-1]); r=d['roofline']; print('fm', round(d['ms_per_step'],4), r.get('kernel_ms'), r.get('frac'), 'cpu', d['cpu_baseline']['value'])" | tee -a $out/summary.txt
