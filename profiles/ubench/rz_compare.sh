# scratch: ranking tests + default / fresh-grads bench lines
mkdir -p gpurun_out/rz
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -1
for f in "" "--fresh-grads" ""; do python bench.py --no-cpu-baseline $f 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$f', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])"; done
