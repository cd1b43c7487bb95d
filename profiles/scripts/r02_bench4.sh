#!/bin/bash
# the four single-GPU bench lines (+ optionally the whole GPU test suite first: r02_bench4.sh tests)
export TMPDIR=/tmp PYTHONPATH=/root/repo
cd /root/repo
out=/root/repo/gpurun_out/bench4; mkdir -p $out
if [ "$1" = "tests" ]; then timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3; fi
for c in fm youtubednn deepfm sasrec; do
  timeout 400 python bench.py --config $c --no-cpu-baseline 2>/dev/null | tee $out/bench_$c.json | cut -c1-420
done
