#!/bin/bash
# whole GPU suite + SASRec step after the sequence-block chains and the view in SASRec.forward
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > $out/sb_gpu_tests.log
B="--config sasrec --steps 20 --warmup 5 --no-cpu-baseline"
timeout 300 python bench.py $B > $out/sb_bench_on.json 2> $out/sb_bench_on.err
grep -h ms_per_step $out/sb_bench_on.json | cut -c1-300
tail -4 $out/sb_gpu_tests.log
