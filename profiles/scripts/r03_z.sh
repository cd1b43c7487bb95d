cd /root/repo; mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests -x -q -m gpu -k "attn or attention or sasrec or sdpa or mha or dropout" 2>&1 | tail -3 > gpurun_out/attn_test.txt
for b in 0 0_noqfirst 0 0_noqfirst; do ./profiles/ubench/attn_parts_$b; done > gpurun_out/attn_q.txt 2>&1
(cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/p1 -o b -- python /root/repo/bench.py --config sasrec --steps 10 --warmup 3 --no-cpu-baseline > /tmp/log_1 2>&1)
db=$(find /tmp/p1 -name "*.db" | head -1)
python profiles/topk.py $db 13 2>&1 | grep -i "attn" >> gpurun_out/attn_q.txt
python bench.py --config sasrec --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null| python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])" >> gpurun_out/attn_q.txt
