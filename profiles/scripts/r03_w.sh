#!/bin/bash
out=/root/repo/gpurun_out/r3w
rm -rf $out; mkdir -p $out
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_matching.py -x -q -k "linear or mlp" 2>&1 | tail -3
for v in 1; do
echo "== RECBOX_AMD_GEMM_BX6=$v"
RECBOX_AMD_GEMM_BX6=$v PYTHONPATH=/root/repo timeout 300 python profiles/gemm_shapes.py 2>&1 | grep -v amdgpu.ids | tee $out/gemm_shapes_bx6_$v.txt
for cfg in deepfm youtubednn; do
RECBOX_AMD_GEMM_BX6=$v timeout 300 python bench.py --config $cfg --no-cpu-baseline --steps 20 --warmup 5 2>$out/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$cfg bx6=$v', round(d['ms_per_step'],4))" || tail -5 $out/err.txt
done
done
