#!/bin/bash
# Regenerates the round-1 measurement files under gpurun_out/refresh/ (run on the GPU box from the repo root);
# the summaries are then copied into profiles/ by hand.
out=/root/repo/gpurun_out/refresh
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
cd /root/repo
timeout 400 python bench.py > $out/bench_fused.json 2> $out/bench_fused.err
timeout 200 python bench.py --no-cpu-baseline --dist zipf > $out/bench_fused_zipf.json 2>/dev/null
timeout 200 python bench.py --no-cpu-baseline --eager > $out/bench_fused_eager.json 2>/dev/null
timeout 200 python bench.py --no-cpu-baseline --path layers > $out/bench_layers.json 2>/dev/null
timeout 200 python bench.py --no-cpu-baseline --force-sharded 2>/dev/null | grep '^{' > $out/bench_sharded_w1.json
timeout 200 python bench.py --no-cpu-baseline --force-sharded --eager 2>/dev/null | grep '^{' > $out/bench_sharded_w1_eager.json
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $out/prof_fused -o b -- python /root/repo/bench.py --no-cpu-baseline > $out/prof_fused.log 2>&1)
python profiles/topk.py $(find $out/prof_fused -name "*.db" | head -1) 58 > $out/fused_kernel_stats.txt
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $out/prof_sharded -o b -- python /root/repo/bench.py --no-cpu-baseline --force-sharded > $out/prof_sharded.log 2>&1)
python profiles/topk.py $(find $out/prof_sharded -name "*.db" | head -1) 64 > $out/sharded_kernel_stats.txt
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $out/prof_layers -o b -- python /root/repo/bench.py --no-cpu-baseline --path layers > $out/prof_layers.log 2>&1)
python profiles/topk.py $(find $out/prof_layers -name "*.db" | head -1) 73 > $out/layers_kernel_stats.txt
# one hipGraph replay as a timeline (which kernels overlap, where the chain waits)
(cd /tmp && timeout 400 rocprofv3 --kernel-trace -d $out/prof_tl -o b -- python /root/repo/bench.py --no-cpu-baseline > /dev/null 2>&1)
python profiles/timeline.py $(find $out/prof_tl -name "*.db" | head -1) rezero_rows 30 > $out/fused_replay_timeline.txt
# HBM traffic of the dominant kernels: one counter per pass, kernel trace only (MI355X_MICROARCH.md, HBM section)
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $c -d $out/pmc_$c -o b -- python /root/repo/bench.py --no-cpu-baseline --eager --steps 5 --warmup 3 > /dev/null 2>&1)
  python profiles/pmc.py $(find $out/pmc_$c -name "*.db" | head -1) $c > $out/pmc_$c.txt
done
timeout 500 python profiles/ubench/kernels_bench.py > $out/kernels_bench.txt 2>/dev/null
(cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats -d $out/prof_kb -o b -- python /root/repo/profiles/ubench/kernels_bench.py > $out/prof_kb.log 2>&1)
python profiles/topk.py $(find $out/prof_kb -name "*.db" | head -1) > $out/kernels_bench_kernel_stats.txt
timeout 600 python profiles/ubench/models_bench.py > $out/models_bench.txt 2>/dev/null
timeout 300 python profiles/ubench/rocblas_compare.py > $out/rocblas_compare.txt 2>/dev/null
rm -rf $out/prof_fused $out/prof_layers $out/prof_sharded $out/prof_kb $out/prof_tl $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE
ls -la $out
