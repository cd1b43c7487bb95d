cd /root/repo; mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests -x -q -m gpu -k "attn or attention or sasrec or sdpa or mha or dropout" 2>&1 | tail -3 > gpurun_out/attn_test.txt
for v in default nohoist; do
  if [ $v = default ]; then unset RECBOX_HIP_LIB; else export RECBOX_HIP_LIB=/root/repo/recbox_amd/lib/variants/$v.so; fi
  (cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/p$v -o b -- python /root/repo/bench.py --config sasrec --steps 10 --warmup 3 --no-cpu-baseline > /tmp/log_$v 2>&1)
  db=$(find /tmp/p$v -name "*.db" | head -1)
  python profiles/topk.py $db 13 2>&1 | grep -i "attn" > gpurun_out/attn_$v.txt
  python bench.py --config sasrec --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null| python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$v', d['ms_per_step'])" >> gpurun_out/attn_$v.txt
done
