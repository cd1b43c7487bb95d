cd /root/repo; mkdir -p gpurun_out
python -m pytest tests/test_gpu_matching.py -x -q -m gpu 2>&1 | tail -4 > gpurun_out/bn_test.txt
for m in "" 0; do
  echo "== BN_IN_GEMM=$m"; env ${m:+RECBOX_AMD_BN_IN_GEMM=$m} python bench.py --config deepfm --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null| python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"
done > gpurun_out/bn_ab.txt
