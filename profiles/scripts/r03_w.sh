#!/bin/bash
out=/root/repo/gpurun_out/r3w
rm -rf $out; mkdir -p $out
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_matching.py -x -q -k "linear or mlp or split_bf16" 2>&1 | tail -3
for v in 1 2; do
echo "== RBX_GEMM_BX6=$v"
RBX_GEMM_BX6=$v PYTHONPATH=/root/repo timeout 300 python profiles/gemm_shapes.py 2>&1 | grep -v amdgpu.ids | tee $out/gemm_shapes_bx6_$v.txt
RBX_GEMM_BX6=$v timeout 300 python bench.py --config deepfm --no-cpu-baseline --steps 20 --warmup 5 2>$out/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('deepfm bx6=$v', round(d['ms_per_step'],4))" || tail -5 $out/err.txt
done
