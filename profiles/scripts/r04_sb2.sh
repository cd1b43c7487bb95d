#!/bin/bash
# sequence-block chains: parity + SASRec step A/B (order of the FFN backward's products) on one box + kernel stats
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
python -m pytest tests/test_gpu_seqblock.py -q 2>&1 | tail -40 > $out/sb_tests.log
RBX_SB_BWD_ORDER=0 python -m pytest tests/test_gpu_seqblock.py -q -k "ffn_backward" 2>&1 | tail -5 >> $out/sb_tests.log
python -m pytest tests/test_gpu_matching.py -x -q -k "sasrec" 2>&1 | tail -8 >> $out/sb_tests.log
B="--config sasrec --steps 20 --warmup 5 --no-cpu-baseline"
RBX_SB_BWD_ORDER=0 timeout 300 python bench.py $B > $out/sb_bench_order0.json 2> $out/sb_bench_order0.err
timeout 300 python bench.py $B > $out/sb_bench_on.json 2> $out/sb_bench_on.err
for o in 0 1; do
rm -rf $out/prof
(cd /tmp && export TMPDIR=/tmp && RBX_SB_BWD_ORDER=$o timeout 600 rocprofv3 --kernel-trace --stats -d $out/prof -o b -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra-configs --config sasrec --steps 10 --warmup 3 > $out/prof_sb.log 2>&1)
db=$(find $out/prof -name "*.db" | head -1)
python profiles/topk.py $db 40 > $out/sb_sasrec_kernel_stats_order$o.txt
done
python profiles/timeline.py $db embed_seq 8 > $out/sb_sasrec_replay_timeline.txt 2>&1
rm -rf $out/prof
grep -h ms_per_step $out/sb_bench_*.json | cut -c1-300
grep -E "passed|failed" $out/sb_tests.log
grep -h "sb_" $out/sb_sasrec_kernel_stats_order*.txt | cut -c1-110
