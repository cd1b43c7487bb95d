#!/bin/bash
out=/root/repo/gpurun_out/r2l
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
cd /root/repo
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $out/prof -o b -- python /root/repo/bench.py --no-cpu-baseline --contiguous-ids > $out/prof.log 2>&1)
python profiles/topk.py $(find $out/prof -name "*.db" | head -1) 58 > $out/kernel_stats_contiguous_ids.txt
rm -rf $out/prof
grep -E "fm_fused_fwd|segment_reduce|build_keys" $out/kernel_stats_contiguous_ids.txt
grep '^{' $out/prof.log | cut -c1-400
timeout 300 python bench.py --no-cpu-baseline --force-sharded 2>$out/sharded.err | grep '^{' > $out/bench_fm_sharded1.json; cut -c1-700 $out/bench_fm_sharded1.json; tail -2 $out/sharded.err | cut -c1-300
timeout 300 python bench.py --config sasrec --no-cpu-baseline --force-sharded --steps 10 --warmup 3 2>$out/sasrec_dp.err | grep '^{' > $out/bench_sasrec_dp1.json; cut -c1-500 $out/bench_sasrec_dp1.json; tail -2 $out/sasrec_dp.err | cut -c1-300
