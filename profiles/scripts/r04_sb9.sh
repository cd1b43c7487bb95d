#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
python -m pytest tests/test_gpu_seqblock.py -q 2>&1 | tail -5 > $out/sb_tests.log
B="--config sasrec --steps 20 --warmup 5 --no-cpu-baseline"
for i in 1 2; do
RECBOX_AMD_SEQBLOCK_FFN3=1 timeout 300 python bench.py $B > $out/sb_ffn3_on_$i.json 2> $out/sb_bench_on.err
timeout 300 python bench.py $B > $out/sb_ffn3_off_$i.json 2> /dev/null
done
rm -rf $out/prof
(cd /tmp && export TMPDIR=/tmp && RECBOX_AMD_SEQBLOCK_FFN3=1 timeout 600 rocprofv3 --kernel-trace --stats -d $out/prof -o b -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra-configs --config sasrec --steps 10 --warmup 3 > $out/prof_sb.log 2>&1)
db=$(find $out/prof -name "*.db" | head -1)
python profiles/topk.py $db 40 > $out/sb_sasrec_kernel_stats.txt
rm -rf $out/prof
for f in sb_ffn3_on_1 sb_ffn3_off_1 sb_ffn3_on_2 sb_ffn3_off_2; do echo $f $(python -c "import json,sys; d=json.load(open('$out/$f.json')); print(d['ms_per_step'])"); done
grep -E "passed|failed" $out/sb_tests.log
grep -h "sb_\|tall_dw" $out/sb_sasrec_kernel_stats.txt | cut -c1-110
