#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out
export PYTHONPATH=$GRAFT_REPO_ROOT
python profiles/gemm_shapes.py > $out/gemm_shapes_bxp.txt 2>&1
RBX_GEMM_BXQ=1 python profiles/gemm_shapes.py > $out/gemm_shapes_bxq.txt 2>&1
echo "--- 16-k tiles (gemm_bxp_kernel)"; cat $out/gemm_shapes_bxp.txt | cut -c1-160
echo "--- 32-k tiles (gemm_bxq_kernel)"; cat $out/gemm_shapes_bxq.txt | cut -c1-160
