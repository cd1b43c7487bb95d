#!/bin/bash
out=/root/repo/gpurun_out/r2s
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_ranking.py -x -q -m gpu -k "fm or graph or reuse or determin or packed" > $out/tests.log 2>&1; tail -2 $out/tests.log
for v in 0 1; do
  for rep in 1 2; do
    RECBOX_AMD_NUMERIC_BESIDE=$v timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('NUMERIC_BESIDE=$v', r['ms_per_step'], r['roofline']['kernel_ms'], r['roofline'].get('kernel_ms_alone'))"
  done
done
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $out/prof -o b -- python /root/repo/bench.py --no-cpu-baseline > $out/prof.log 2>&1)
python profiles/topk.py $(find $out/prof -name "*.db" | head -1) 58 > $out/kernel_stats.txt
python profiles/timeline.py $(find $out/prof -name "*.db" | head -1) rezero_rows 30 > $out/timeline.txt 2>&1
rm -rf $out/prof
cat $out/timeline.txt | head -34
