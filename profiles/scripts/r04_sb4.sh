#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
python -m pytest tests/test_gpu_seqblock.py -q 2>&1 | tail -5 > $out/sb_tests.log
RECBOX_AMD_FM_NUMERIC_ON=side python -m pytest tests/test_gpu_ranking.py -q -x -k "fm" 2>&1 | tail -4 >> $out/sb_tests.log
B="--config sasrec --steps 20 --warmup 5 --no-cpu-baseline"
timeout 300 python bench.py $B > $out/sb_bench_on.json 2> $out/sb_bench_on.err
F="--steps 50 --warmup 10 --no-extra-configs --no-cpu-baseline"
for i in 1 2; do
timeout 300 python bench.py $F > $out/fm_main_$i.json 2>/dev/null
RECBOX_AMD_FM_NUMERIC_ON=side timeout 300 python bench.py $F > $out/fm_side_$i.json 2>/dev/null
done
RECBOX_AMD_FM_NUMERIC_ON=side RBX_FM_TIER_C=1 RECBOX_AMD_FM_BLOCKSORT_AT=side timeout 300 python bench.py $F > $out/fm_side_tierc.json 2>/dev/null
rm -rf $out/prof
(cd /tmp && export TMPDIR=/tmp && RECBOX_AMD_FM_NUMERIC_ON=side timeout 600 rocprofv3 --kernel-trace --stats -d $out/prof -o b -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra-configs --steps 20 --warmup 5 > $out/prof_sb.log 2>&1)
db=$(find $out/prof -name "*.db" | head -1)
python profiles/timeline.py $db compact_ids 30 > $out/fm_side_replay_timeline.txt 2>&1
rm -rf $out/prof
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $out/prof -o b -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra-configs --config sasrec --steps 10 --warmup 3 > $out/prof_sb.log 2>&1)
db=$(find $out/prof -name "*.db" | head -1)
python profiles/topk.py $db 40 > $out/sb_sasrec_kernel_stats.txt
rm -rf $out/prof
for f in sb_bench_on fm_main_1 fm_side_1 fm_main_2 fm_side_2 fm_side_tierc; do echo $f $(python -c "import json,sys; d=json.load(open('$out/$f.json')); print(d['ms_per_step'])"); done
grep -E "passed|failed" $out/sb_tests.log
grep -h "sb_" $out/sb_sasrec_kernel_stats.txt | cut -c1-110
