#!/bin/bash
out=/root/repo/gpurun_out/r2o
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
cd /root/repo
timeout 1200 python -m pytest tests/test_gpu_matching.py -x -q -m gpu > $out/tests.log 2>&1; tail -3 $out/tests.log
for w in 0 1 2; do
  RBX_GEMM_WIDE=$w timeout 300 python bench.py --config deepfm --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('deepfm WIDE=$w', round(r['ms_per_step'],3), round(r['roofline']['kernel_ms'],4), round(r['roofline']['frac'],3))"
  RBX_GEMM_WIDE=$w timeout 300 python bench.py --config youtubednn --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('youtubednn WIDE=$w', round(r['ms_per_step'],3))"
done
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $out/prof -o b -- python /root/repo/bench.py --config deepfm --no-cpu-baseline --steps 20 --warmup 5 > $out/prof.log 2>&1)
python profiles/topk.py $(find $out/prof -name "*.db" | head -1) 26 > $out/deepfm_kernel_stats.txt
rm -rf $out/prof
head -22 $out/deepfm_kernel_stats.txt
timeout 300 python bench.py --no-cpu-baseline > $out/bench_fm.json 2>/dev/null; cut -c1-1800 $out/bench_fm.json
