cd /root/repo; mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu -k "sasrec or row_scale or rowscale or position" 2>&1 | tail -3 > gpurun_out/pos_test.txt
for m in 1 0; do
  echo "== RECBOX_AMD_SEQ_POSITIONS=$m"; RECBOX_AMD_SEQ_POSITIONS=$m python bench.py --config sasrec --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null| python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"
done > gpurun_out/pos_ab.txt 2>&1
