cd /root/repo; mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu -k "attn or attention or sasrec or sdpa or mha or dropout" 2>&1 | tail -3 > gpurun_out/attn_test.txt
for m in 1 0; do
  echo "== RBX_ATTN_SPLIT=$m"; RBX_ATTN_SPLIT=$m python bench.py --config sasrec --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null| python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['roofline'].get('kernel_ms'))"
done > gpurun_out/attn_ab.txt 2>&1
