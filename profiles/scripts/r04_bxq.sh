#!/bin/bash
# split-operand GEMM with k tiles of 32 (gemm_bxq_kernel, RBX_GEMM_BXQ=1) against the 16-k form: parity, shapes, DeepFM step
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
RBX_GEMM_BXQ=1 python -m pytest tests/test_gpu_matching.py -q -k "split_bf16 or linear or mlp or deepfm or youtubednn" 2>&1 | tail -4 > $out/bxq_tests.log
python profiles/gemm_shapes.py > $out/gemm_shapes_bxp.txt 2>&1
RBX_GEMM_BXQ=1 python profiles/gemm_shapes.py > $out/gemm_shapes_bxq.txt 2>&1
B="--config deepfm --steps 20 --warmup 5 --no-cpu-baseline"
for i in 1 2; do
timeout 300 python bench.py $B > $out/dfm_bxp_$i.json 2> /dev/null
RBX_GEMM_BXQ=1 timeout 300 python bench.py $B > $out/dfm_bxq_$i.json 2> /dev/null
done
for f in dfm_bxp_1 dfm_bxq_1 dfm_bxp_2 dfm_bxq_2; do echo $f $(python -c "import json,sys; d=json.load(open('$out/$f.json')); print(d['ms_per_step'])"); done
grep -E "passed|failed" $out/bxq_tests.log
paste $out/gemm_shapes_bxp.txt $out/gemm_shapes_bxq.txt | cut -c1-250 | head -20
