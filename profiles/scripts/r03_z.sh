cd /root/repo; mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | tail -3 > gpurun_out/sr_test.txt
for c in fm youtubednn deepfm sasrec; do
  if [ $c = fm ]; then a="--no-extra-configs"; else a="--config $c"; fi
  python bench.py $a --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$c', d['ms_per_step'])"
done > gpurun_out/sr_bench.txt 2>&1
