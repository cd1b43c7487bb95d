cd /root/repo; mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests -x -q -m gpu -k "layer_norm or layernorm or sasrec or ln_ or norm" 2>&1 | tail -3 > gpurun_out/ln_test.txt
(cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/p1 -o b -- python /root/repo/bench.py --config sasrec --steps 10 --warmup 3 --no-cpu-baseline > /tmp/log_1 2>&1)
db=$(find /tmp/p1 -name "*.db" | head -1)
python profiles/topk.py $db 13 2>&1 | grep -i "ln_\|kernel " > gpurun_out/ln_prof.txt
python bench.py --config sasrec --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null| python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])" >> gpurun_out/ln_prof.txt
