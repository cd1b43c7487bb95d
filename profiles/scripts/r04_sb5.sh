#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
python -m pytest tests/test_gpu_seqblock.py -q 2>&1 | tail -5 > $out/sb_tests.log
python -m pytest tests/test_gpu_matching.py -x -q -k "sasrec" 2>&1 | tail -4 >> $out/sb_tests.log
B="--config sasrec --steps 20 --warmup 5 --no-cpu-baseline"
timeout 300 python bench.py $B > $out/sb_bench_on.json 2> $out/sb_bench_on.err
RECBOX_AMD_SEQBLOCK_BWD=0 timeout 300 python bench.py $B > $out/sb_bench_fwdonly.json 2> /dev/null
for f in sb_bench_on sb_bench_fwdonly; do echo $f $(python -c "import json,sys; d=json.load(open('$out/$f.json')); print(d['ms_per_step'])"); done
grep -E "passed|failed" $out/sb_tests.log
